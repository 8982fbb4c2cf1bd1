#!/bin/bash
set -o pipefail
O=gpurun_out/r02i; mkdir -p $O
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method thread > $O/gpu_tests.log 2>&1 || { tail -40 $O/gpu_tests.log; exit 1; }
tail -4 $O/gpu_tests.log
