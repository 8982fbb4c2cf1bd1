#!/bin/bash
set -o pipefail
O=gpurun_out/r02c; mkdir -p $O
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method thread --durations=12 > $O/gpu_tests.log 2>&1 || { tail -40 $O/gpu_tests.log; exit 1; }
tail -25 $O/gpu_tests.log
