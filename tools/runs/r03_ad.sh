#!/bin/bash
# r03_ad: the fused LayerNorm + projection + arg-max kernel against its three separate launches, and a build of it without the
# atomics (abl_out/libtst_va1.so, -DVA_ABL=1); then its parity tests and config 1 with it on and off
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r03_ad.txt
timeout -k 10 120 python tools/bench_vocab_argmax.py >> gpurun_out/r03_ad.txt 2>&1 || { tail gpurun_out/r03_ad.txt; exit 1; }
TST_HIP_LIB=abl_out/libtst_va1.so timeout -k 10 120 python tools/bench_vocab_argmax.py >> gpurun_out/r03_ad.txt 2>&1 || { tail gpurun_out/r03_ad.txt; exit 1; }
grep rows gpurun_out/r03_ad.txt
timeout -k 10 600 python -u -m pytest tests/test_gpu_argmax_keys.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r03_ad_tests.log 2>&1 || { tail -40 gpurun_out/r03_ad_tests.log; exit 1; }
tail -2 gpurun_out/r03_ad_tests.log
for sw in "" "TST_NO_ARGMAX_KEYS=1" "" "TST_NO_ARGMAX_KEYS=1"; do
  echo "switches=[$sw]"
  env $sw timeout -k 10 200 python tools/bench_configs.py cfg1 --sentences 256 2>&1 | grep '"case"' | cut -c80-190 || exit 1
done
