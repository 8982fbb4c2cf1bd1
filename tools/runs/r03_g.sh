#!/bin/bash
# r03_g: row-tile chain kernel: parity tests, then chain against the three separate launches
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_rowchain.py -x -q --timeout 120 --timeout-method thread > gpurun_out/r03_g_tests.log 2>&1 || { tail -40 gpurun_out/r03_g_tests.log; exit 1; }
tail -3 gpurun_out/r03_g_tests.log
timeout -k 10 200 python tools/bench_chain.py > gpurun_out/r03_g_bench.log 2>&1 || { tail -30 gpurun_out/r03_g_bench.log; exit 1; }
cat gpurun_out/r03_g_bench.log
