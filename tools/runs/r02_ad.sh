#!/bin/bash
set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py -x -q --timeout 200 --timeout-method thread -k "topk or logsoftmax" 2>&1 | tail -2 || exit 1
for lib in "" tools/probe/libtst_base.so "" tools/probe/libtst_base.so; do
  echo "lib=$lib"
  TST_HIP_LIB=$lib timeout -k 10 100 python -u tools/probe/bench_select.py | cut -c1-60 || exit 1
  TST_HIP_LIB=$lib REPS=3 timeout -k 10 200 python -u tools/probe/e2e_cfg2.py | tail -1 || exit 1
done
