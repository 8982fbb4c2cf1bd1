#!/bin/bash
# layer products at K = 256 with the weights held in registers (TST_GEMM_WREG=1) against the tile kernels: times and checksums
for rows in 15676 8192 30000 4200 20480; do
  echo "== rows $rows: tile kernels"; REPS=20 timeout -k 10 120 python tools/bench_gemm_layers.py $rows || exit 1
  echo "== rows $rows: weights in registers"; TST_GEMM_WREG=1 REPS=20 timeout -k 10 120 python tools/bench_gemm_layers.py $rows || exit 1
done
