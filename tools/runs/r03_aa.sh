#!/bin/bash
# r03_aa: bench.py with every batch decoded as 1 / 2 / 3 concurrent sub-batches (--sub-batches), alternating on one box
set -o pipefail
for p in 1 2 1 2 3; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline --sub-batches $p > gpurun_out/r03_aa_$p.json 2>gpurun_out/r03_aa.err || { tail gpurun_out/r03_aa.err; exit 1; }
  python - <<PY
import json
d = json.loads(open('gpurun_out/r03_aa_$p.json').read().strip().splitlines()[-1])
print('sub_batches $p ms_per_step %.2f tokens/s %.0f row_steps %.0f checksum %d' % (d['ms_per_step'], d['value'], d['config']['row_steps_per_batch'], d['output_checksum']))
PY
done | tee gpurun_out/r03_aa_sub_batches.txt
