#!/bin/bash
set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py -x -q --timeout 200 --timeout-method thread -k "gemm" 2>&1 | tail -2 || exit 1
for lib in "" tools/probe/libtst_base.so "" tools/probe/libtst_base.so; do
  echo "lib=$lib"
  TST_HIP_LIB=$lib timeout -k 10 300 python -u tools/bench_configs.py cfg3 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{') and 'case' in l:
        d=json.loads(l); print(d['case'][:52], round(d['ms_per_batch'],2), round(d['tokens_per_s']))" || exit 1
done
