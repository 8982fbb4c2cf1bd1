#!/bin/bash
set -o pipefail
for r in 65536 20480; do
timeout -k 10 120 python -u tools/bench_gemm_bf16.py $r vocab,qkv,"ff1 relu" || exit 1
TST_BF16_NO_PERSIST=1 timeout -k 10 120 python -u tools/bench_gemm_bf16.py $r vocab,qkv,"ff1 relu" || exit 1
done
timeout -k 10 200 python -u tools/bench_configs.py cfg4 || exit 1
TST_BF16_NO_PERSIST=1 timeout -k 10 200 python -u tools/bench_configs.py cfg4 || exit 1
