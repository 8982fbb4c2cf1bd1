#!/bin/bash
set -o pipefail
timeout -k 10 400 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_decode.py tests/test_gpu_lm_bf16.py tests/test_gpu_config_sizes.py -x -q --timeout 200 --timeout-method thread 2>&1 | tail -2 || exit 1
for lib in "" tools/probe/libtst_base.so "" tools/probe/libtst_base.so; do
  echo "lib=$lib"
  TST_HIP_LIB=$lib timeout -k 10 200 python -u tools/probe/e2e_cfg2.py | tail -2 || exit 1
  TST_HIP_LIB=$lib timeout -k 10 200 python -u tools/bench_configs.py cfg4 --dtype bf16 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('cfg4 ms', d['ms_per_batch'])" || exit 1
done
