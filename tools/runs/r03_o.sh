#!/bin/bash
# r03_o: concurrent-batches test, then the whole GPU suite and the default bench line
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 200 python -u -m pytest tests/test_gpu_edge_cases.py -x -q -k concurrent --timeout 120 --timeout-method thread 2>&1 | tail -15 || exit 1
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/r03_o_tests.log 2>&1 || { tail -30 gpurun_out/r03_o_tests.log; exit 1; }
tail -3 gpurun_out/r03_o_tests.log
timeout -k 10 300 python bench.py > gpurun_out/r03_o_bench.log 2>&1 || { tail -30 gpurun_out/r03_o_bench.log; exit 1; }
tail -1 gpurun_out/r03_o_bench.log | cut -c1-400
