#!/bin/bash
# LayerNorm inside the staging of the weights-in-registers kernel for the large K = 256 products behind a LayerNorm (q|k|v, cross-q, ff1):
# parity under every switch first, then config 2 with TST_LN_WREG = 0 (separate LayerNorm launch + tile kernels), 1 (outputs of >= 512 columns), 2 (all)
timeout -k 10 600 python -u -m pytest tests/test_gpu_switches.py tests/test_gpu_config_sizes.py -q -x --timeout 300 --timeout-method thread -p no:cacheprovider 2>&1 | tail -3 || exit 1
for v in 0 1 2 0 1 2; do
  TST_LN_WREG=$v timeout -k 10 200 python bench.py --no-cpu-baseline --no-roofline --steps 8 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('TST_LN_WREG=$v', round(d['ms_per_step'],2), round(d['value']), d['output_checksum'])" || exit 1
done
