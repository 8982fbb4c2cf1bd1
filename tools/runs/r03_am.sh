#!/bin/bash
# r03_am: the bf16 layer products on 128 x 128 tiles with a ring of 3 stages at THREE workgroups per CU (TST_BF16_TILE=4) against the
# shapes in use (default rule), 4 k - 65 k rows
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r03_am.txt
for rows in 4096 8192 12288 18432 24576 32768 49152 65536; do
  for env in "" "TST_BF16_TILE=4"; do
    echo "rows=$rows [$env]" >> gpurun_out/r03_am.txt
    env $env timeout -k 10 100 python tools/bench_gemm_bf16.py $rows "qkv,proj+res,ff1 relu,ff2+res" 2>&1 | grep -E "qkv|proj|ff1|ff2" | cut -c1-100 >> gpurun_out/r03_am.txt || exit 1
  done
done
cat gpurun_out/r03_am.txt
