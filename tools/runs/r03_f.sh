#!/bin/bash
# r03_f: the whole GPU suite and the default bench line on a freshly built tree
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/r03_f_tests.log 2>&1 || { tail -30 gpurun_out/r03_f_tests.log; exit 1; }
tail -3 gpurun_out/r03_f_tests.log
timeout -k 10 300 python bench.py > gpurun_out/r03_f_bench.log 2>&1 || { tail -30 gpurun_out/r03_f_bench.log; exit 1; }
tail -1 gpurun_out/r03_f_bench.log
