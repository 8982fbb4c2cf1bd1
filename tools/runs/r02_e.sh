#!/bin/bash
set -o pipefail
O=gpurun_out/r02e; mkdir -p $O
timeout -k 10 200 python -u tools/bench_gemm_dma.py > $O/gemm_dma.log 2>&1 || { tail -20 $O/gemm_dma.log; exit 1; }
cat $O/gemm_dma.log
