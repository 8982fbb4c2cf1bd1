#!/bin/bash
# the vocabulary projection with 64-row tiles (two MFMA chains per wave; default from 4096 rows) against 32-row tiles (TST_WREG_ROW_BLOCKS=1):
# parity tests at full size first, then config 2 alternating
timeout -k 10 600 python -u -m pytest tests/test_gpu_config_sizes.py tests/test_gpu_switches.py -q -x --timeout 300 --timeout-method thread -p no:cacheprovider 2>&1 | tail -3 || exit 1
for v in 1 2 1 2 1 2; do
  TST_WREG_ROW_BLOCKS=$v timeout -k 10 200 python bench.py --no-cpu-baseline --steps 8 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('row blocks $v', round(d['ms_per_step'],2), round(d['value']), d['output_checksum'], 'projection %.1f TF %.3f ms per launch' % (d['roofline']['achieved'], d['roofline']['avg_launch_ms']))" || exit 1
done
