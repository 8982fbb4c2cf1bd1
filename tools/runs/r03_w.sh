#!/bin/bash
# r03_w: the N > 1 code path on the final tree: config 5 at 65536 sentences with 1 rank and with 2 ranks sharing the GPU (gloo),
# and config 2 with 2 ranks sharing the GPU: output checksums, rank_ms / gather_ms fields
set -o pipefail
O=gpurun_out/r03_w; mkdir -p $O
timeout -k 10 300 python -u bench.py --config 5 --sentences 65536 --no-cpu-baseline --no-roofline > $O/c5_n1.json 2> $O/c5_n1.err || { tail -20 $O/c5_n1.err; exit 1; }
timeout -k 10 400 python -u bench.py --gpus 2 --share-gpu --backend gloo --config 5 --sentences 65536 --no-cpu-baseline --no-roofline > $O/c5_n2.json 2> $O/c5_n2.err || { tail -20 $O/c5_n2.err; exit 1; }
timeout -k 10 400 python -u bench.py --gpus 2 --share-gpu --backend gloo --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/c2_n2.json 2> $O/c2_n2.err || { tail -20 $O/c2_n2.err; exit 1; }
python - <<'PY'
import json
for f in ('c5_n1', 'c5_n2', 'c2_n2'):
    d = json.loads([l for l in open('gpurun_out/r03_w/%s.json' % f) if l.startswith('{')][-1])
    print(f, 'ranks', d['n_gpus'], 'checksum', d['output_checksum'], 'tokens/s %.0f' % d['value'], 'ms %.0f' % d['ms_per_step'], 'rank_ms', d.get('rank_ms'), 'gather_ms', d.get('gather_ms'))
PY
