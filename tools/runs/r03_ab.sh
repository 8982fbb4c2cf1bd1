#!/bin/bash
# r03_ab: greedy small batches with final LayerNorm + vocabulary projection + arg-max as one kernel and the survivor count stored
# into host memory by the tail kernel: parity tests, config 1 (128 / 256 sentences) with each switch on and off
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_argmax_keys.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r03_ab_tests.log 2>&1 || { tail -40 gpurun_out/r03_ab_tests.log; exit 1; }
tail -2 gpurun_out/r03_ab_tests.log
: > gpurun_out/r03_ab_lines.txt
for n in 256 128; do
  for sw in "" "TST_NO_ARGMAX_KEYS=1" "TST_NO_HOST_COUNT=1" "TST_NO_ARGMAX_KEYS=1 TST_NO_HOST_COUNT=1" ""; do
    echo "sentences=$n switches=[$sw]" >> gpurun_out/r03_ab_lines.txt
    env $sw timeout -k 10 200 python tools/bench_configs.py cfg1 --sentences $n 2>&1 | grep '"case"' | cut -c1-330 >> gpurun_out/r03_ab_lines.txt || exit 1
  done
done
cat gpurun_out/r03_ab_lines.txt
