#!/bin/bash
set -o pipefail
O=gpurun_out/r02f; mkdir -p $O
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method thread > $O/gpu_tests.log 2>&1 || { tail -40 $O/gpu_tests.log; exit 1; }
tail -4 $O/gpu_tests.log
timeout -k 10 400 python -u bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_cfg2.json 2> $O/bench_cfg2.err || { tail -30 $O/bench_cfg2.err; exit 1; }
python - <<'PY'
import json
r = json.loads(open('gpurun_out/r02f/bench_cfg2.json').read().strip().splitlines()[-1])
print('value', r['value'], 'ms', r['ms_per_step'], 'roofline', r['roofline']['achieved'], r['roofline']['avg_launch_ms'], 'select', r['roofline_select']['achieved'], 'checksum', r['output_checksum'])
PY
