#!/bin/bash
set -o pipefail
for r in 8192 16384 24576 32768 65536; do
for tile in 1 2 p; do
echo "rows $r tile $tile"
if [ $tile = p ]; then
timeout -k 10 120 python -u tools/bench_gemm_bf16.py $r "vocab,qkv,ff1 relu" || exit 1
else
TST_BF16_TILE=$tile TST_BF16_NO_PERSIST=1 timeout -k 10 120 python -u tools/bench_gemm_bf16.py $r "vocab,proj+res,ff2+res,qkv,ff1 relu" || exit 1
fi
done; done
