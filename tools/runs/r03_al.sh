#!/bin/bash
# r03_al: fp32 + residual products of the bf16 model (o-projection K = 512, ff2 K = 2048, N = 512): ring of 3 / 4 / 5 stages on
# the 256 x 256 tile, and the other tile shapes, at 8 k - 65 k rows
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r03_al.txt
for rows in 8192 12288 18432 24576 32768 65536; do
  for env in "TST_BF16_TILE=2 TST_BF16_RES_NST=3" "TST_BF16_TILE=2 TST_BF16_RES_NST=4" "TST_BF16_TILE=2 TST_BF16_RES_NST=5" "TST_BF16_TILE=1" "TST_BF16_TILE=0" "TST_BF16_TILE=3"; do
    echo "rows=$rows [$env]" >> gpurun_out/r03_al.txt
    env $env timeout -k 10 100 python tools/bench_gemm_bf16.py $rows proj+res,ff2+res 2>&1 | grep -E "proj|ff2" | cut -c1-110 >> gpurun_out/r03_al.txt || exit 1
  done
done
cat gpurun_out/r03_al.txt
