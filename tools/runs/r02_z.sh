#!/bin/bash
set -o pipefail
for lib in "" tools/probe/libtst_base.so "" tools/probe/libtst_base.so; do
  echo "lib=$lib"
  TST_HIP_LIB=$lib timeout -k 10 200 python -u tools/bench_gemm_dma.py 32768 15676 | sed 's/.*| dma v3/dma v3/' || exit 1
  TST_HIP_LIB=$lib timeout -k 10 200 python -u tools/bench_gemm_bf16.py 65536 "vocab,qkv,ff1 relu,ff2+res" | cut -c1-75 || exit 1
done
