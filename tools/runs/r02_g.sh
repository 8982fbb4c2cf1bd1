#!/bin/bash
set -o pipefail
O=gpurun_out/r02g; mkdir -p $O
timeout -k 10 200 python -u tools/bench_gemm_dma.py 32768 15676 8192 > $O/gemm_dma.log 2>&1 || { tail -20 $O/gemm_dma.log; exit 1; }
cat $O/gemm_dma.log
