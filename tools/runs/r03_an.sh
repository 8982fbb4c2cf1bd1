#!/bin/bash
# r03_an: config 4 (bf16) with the 128 x 128 / three-workgroups-per-CU tile for the small fp32 + residual products, against the
# library before it (abl_out/libtst_base.so); the routing switches in child processes
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r03_an.txt
for lib in "" "TST_HIP_LIB=abl_out/libtst_base.so" "" "TST_HIP_LIB=abl_out/libtst_base.so"; do
  echo "[$lib]" >> gpurun_out/r03_an.txt
  env $lib timeout -k 10 300 python tools/bench_configs.py cfg4 --dtype bf16 2>&1 | grep '"case"' | cut -c1-230 >> gpurun_out/r03_an.txt || exit 1
done
cat gpurun_out/r03_an.txt
timeout -k 10 900 python -u -m pytest tests/test_gpu_switches.py tests/test_gpu_lm_bf16.py -x -q --timeout 300 --timeout-method thread 2>&1 | tail -3
