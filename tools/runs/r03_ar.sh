#!/bin/bash
# r03_ar: config 5 on the final tree: one million sentences on one GPU; 262144 sentences by 1, 2 and 4 ranks sharing the GPU (gloo):
# equal output checksums, per-rank times
set -o pipefail
O=gpurun_out/r03_ar; mkdir -p $O
timeout -k 10 500 python -u bench.py --config 5 --sentences 1000000 --no-cpu-baseline --no-roofline > $O/c5_1M.json 2> $O/c5_1M.err || { tail -20 $O/c5_1M.err; exit 1; }
for n in 1 2 4; do
  if [ $n = 1 ]; then extra=""; else extra="--gpus $n --share-gpu --backend gloo"; fi
  timeout -k 10 500 python -u bench.py $extra --config 5 --sentences 262144 --no-cpu-baseline --no-roofline > $O/c5_262k_n$n.json 2> $O/c5_262k_n$n.err || { tail -20 $O/c5_262k_n$n.err; exit 1; }
done
python - <<'PY'
import json
for f in ('c5_1M', 'c5_262k_n1', 'c5_262k_n2', 'c5_262k_n4'):
    d = json.loads([l for l in open('gpurun_out/r03_ar/%s.json' % f) if l.startswith('{')][-1])
    print(f, 'ranks', d['n_gpus'], 'checksum', d['output_checksum'], 'tokens/s %.0f' % d['value'], 'ms %.0f' % d['ms_per_step'], 'rank_ms', d.get('rank_ms'), 'gather_ms', d.get('gather_ms'))
PY
