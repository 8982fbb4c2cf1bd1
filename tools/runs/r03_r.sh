#!/bin/bash
# r03_r: fp32 vocabulary projection on 256 x 256 tiles with 8 waves (variant 5) against the 128 x 256 tile: parity, micro-benchmark
set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py -x -q -k "gemm_dma" --timeout 120 --timeout-method thread 2>&1 | tail -3 || exit 1
timeout -k 10 200 python tools/bench_gemm_dma.py 32768 15676 8192 4096 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_r_microbench.txt || exit 1
