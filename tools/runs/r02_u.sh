#!/bin/bash
set -o pipefail
timeout -k 10 400 python -u -m pytest tests/test_gpu_kernels.py -x -q --timeout 200 --timeout-method thread -k "gemm" 2>&1 | tail -2 || exit 1
for lib in "" tools/probe/libtst_base.so; do
  echo "lib=$lib"
  TST_HIP_LIB=$lib timeout -k 10 200 python -u tools/bench_gemm_layers.py 15676 8192 30000 4096 | grep "sum\|o-proj\|ff1" || exit 1
done
