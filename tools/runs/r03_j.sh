#!/bin/bash
# r03_j: config 2 end to end with the row-tile chain kernels on (TST_TFM_CHAIN=1) against the default path, same box:
# parity test at full size under the switch, then the two bench lines and the kernel stats of the chain run
set -o pipefail
mkdir -p gpurun_out
TST_TFM_CHAIN=1 timeout -k 10 300 python -u -m pytest tests/test_gpu_config_sizes.py -x -q -k "config2_beam4 or config2_batch_order" --timeout 200 --timeout-method thread > gpurun_out/r03_j_tests.log 2>&1 || { tail -30 gpurun_out/r03_j_tests.log; exit 1; }
tail -2 gpurun_out/r03_j_tests.log
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/r03_j_off.json 2>gpurun_out/r03_j_off.err || { tail gpurun_out/r03_j_off.err; exit 1; }
TST_TFM_CHAIN=1 timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/r03_j_on.json 2>gpurun_out/r03_j_on.err || { tail gpurun_out/r03_j_on.err; exit 1; }
python - <<'PY'
import json
for f in ('off', 'on'):
    d = json.loads(open('gpurun_out/r03_j_%s.json' % f).read().strip().splitlines()[-1])
    print('chain', f, 'ms_per_step %.2f' % d['ms_per_step'], 'tokens/s %.0f' % d['value'], 'checksum', d['output_checksum'])
PY
TST_TFM_CHAIN=1 bash tools/prof_stats.sh r03_chain_on bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline || exit 1
head -16 gpurun_out/r03_chain_on_stats.txt
