#!/bin/bash
set -o pipefail
for lib in "" tools/probe/libtst_base.so "" tools/probe/libtst_base.so; do
  echo "lib=$lib"
  TST_HIP_LIB=$lib timeout -k 10 200 python -u tools/bench_gemm_layers.py 15676 8192 | grep "sum\|qkv\|ff2" | cut -c1-70 || exit 1
  TST_HIP_LIB=$lib timeout -k 10 200 python -u tools/bench_gru_cell.py | cut -c1-160 | tail -3 || exit 1
done
