#!/bin/bash
# r03_m: bf16 residual products (N = 512) on whole-row tiles of 128 x 512 (what a LayerNorm epilogue would need) against the tiles in use
set -o pipefail
for r in 8192 18432 32768 65536; do
  for tile in d 3; do
    echo "== rows $r tile $tile"
    if [ $tile = d ]; then timeout -k 10 120 python -u tools/bench_gemm_bf16.py $r "proj+res,ff2+res" || exit 1
    else TST_BF16_TILE=$tile timeout -k 10 120 python -u tools/bench_gemm_bf16.py $r "proj+res,ff2+res" || exit 1; fi
  done
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_m_bf16_rowtile.txt
