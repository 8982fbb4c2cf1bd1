#!/bin/bash
set -o pipefail
O=gpurun_out/r02ab; mkdir -p $O
for n in 1 2 4; do
  extra=""; [ $n -gt 1 ] && extra="--share-gpu --backend gloo"
  timeout -k 10 500 python -u bench.py --gpus $n $extra --config 5 --sentences 262144 --no-cpu-baseline --no-roofline > $O/n$n.json 2> $O/n$n.err || { tail -20 $O/n$n.err; exit 1; }
  python - <<PY
import json
d=json.loads([l for l in open('$O/n$n.json') if l.startswith('{')][-1]); print('ranks', d['n_gpus'], 'checksum', d['output_checksum'], 'tokens/s %.0f' % d['value'], 'ms %.0f' % d['ms_per_step'], d['config']['ranks_seen_by_backend'])
PY
done
