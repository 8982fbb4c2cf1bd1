#!/bin/bash
# r03_d: GRU cell with a k-major copy of the state + LDS-DMA output projection (config 3, V = 32000): parity, then throughput
set -o pipefail
mkdir -p gpurun_out/r03d
timeout -k 10 600 python -u -m pytest tests/test_gpu_kernels.py -x -q --timeout 300 --timeout-method thread -k "gru_cell_fused or gemm_dma" 2>&1 | tail -4 || exit 1
timeout -k 10 600 python -u -m pytest tests/test_gpu_config_sizes.py tests/test_gpu_decode.py -x -q --timeout 400 --timeout-method thread -k "config3 or gru" 2>&1 | tail -4 || exit 1
timeout -k 10 400 python -u tools/bench_configs.py cfg3 > gpurun_out/r03d/cfg3.jsonl 2>&1 || exit 1
cut -c1-330 gpurun_out/r03d/cfg3.jsonl
