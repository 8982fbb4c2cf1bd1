#!/bin/bash
set -o pipefail
timeout -k 10 900 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread 2>&1 | tail -4 || exit 1
timeout -k 10 300 python -u bench.py --no-cpu-baseline > gpurun_out/r02p_bench.json || exit 1
python - <<'PY'
import json; d=json.load(open('gpurun_out/r02p_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['output_checksum'])
PY
timeout -k 10 200 python -u tools/bench_configs.py cfg4 | head -1
