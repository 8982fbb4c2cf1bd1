#!/bin/bash
set -o pipefail
timeout -k 10 400 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_decode.py -x -q --timeout 200 --timeout-method thread 2>&1 | tail -2 || exit 1
for lib in "" tools/probe/libtst_base.so "" tools/probe/libtst_base.so; do
  TST_HIP_LIB=$lib timeout -k 10 200 python -u tools/bench_configs.py cfg1 --sentences 256 | python -c "import sys,json; ls=[json.loads(l) for l in sys.stdin if l.startswith('{')]; print('lib=$lib cfg1 ms', [(d['sentences'], round(d['ms_per_batch'],2)) for d in ls])" || exit 1
done
