#!/bin/bash
# r03_af: the fused arg-max kernel with unconditional atomics: micro-benchmark, config 1 with the switches on / off, whole GPU suite
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 120 python tools/bench_vocab_argmax.py 2>&1 | grep rows | cut -c1-140 > gpurun_out/r03_af_micro.txt || exit 1
cat gpurun_out/r03_af_micro.txt
: > gpurun_out/r03_af_cfg1.txt
for sw in "" "TST_NO_ARGMAX_KEYS=1" "TST_NO_ARGMAX_KEYS=1 TST_NO_HOST_COUNT=1" "" "TST_NO_ARGMAX_KEYS=1" "TST_NO_ARGMAX_KEYS=1 TST_NO_HOST_COUNT=1"; do
  echo "switches=[$sw]" >> gpurun_out/r03_af_cfg1.txt
  env $sw timeout -k 10 200 python tools/bench_configs.py cfg1 --sentences 256 2>&1 | grep '"case"' | cut -c80-190 >> gpurun_out/r03_af_cfg1.txt || exit 1
done
cat gpurun_out/r03_af_cfg1.txt
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r03_af_tests.log 2>&1 || { tail -40 gpurun_out/r03_af_tests.log; exit 1; }
tail -2 gpurun_out/r03_af_tests.log
