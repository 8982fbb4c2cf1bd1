#!/bin/bash
set -o pipefail
timeout -k 10 400 python -u -m pytest tests/test_gpu_kernels.py -x -q --timeout 200 --timeout-method thread -k "gemm" 2>&1 | tail -2 || exit 1
for sw in 0 1 0 1; do
  echo "TST_GEMM_NO_K32=$sw"
  TST_GEMM_NO_K32=$sw timeout -k 10 200 python -u tools/bench_gemm_layers.py 15676 8192 4096 | grep -v "qkv\|ff1" || exit 1
done
