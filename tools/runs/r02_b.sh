#!/bin/bash
# full GPU suite (no-alloc test last), peaks probe, config-5 N=1 vs N=2 equality
set -o pipefail
O=gpurun_out/r02b; mkdir -p $O
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
timeout -k 10 60 ./tools/peaks > $O/peaks.json 2>&1 || { cat $O/peaks.json; exit 1; }
cat $O/peaks.json
timeout -k 10 300 python -u bench.py --config 5 --sentences 32768 --no-cpu-baseline --no-roofline --dump $O/c5_n1.pt > $O/bench_cfg5_n1.json 2> $O/bench_cfg5_n1.err || { tail -30 $O/bench_cfg5_n1.err; exit 1; }
timeout -k 10 300 python -u bench.py --gpus 2 --share-gpu --backend gloo --config 5 --sentences 32768 --no-roofline --dump $O/c5_n2.pt > $O/bench_cfg5_n2.json 2> $O/bench_cfg5_n2.err || { tail -30 $O/bench_cfg5_n2.err; exit 1; }
python - <<'PY'
import torch, json
a = torch.load('gpurun_out/r02b/c5_n1.pt'); b = torch.load('gpurun_out/r02b/c5_n2.pt')
print('config 5 outputs N=1 vs N=2 equal:', torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]))
for f in ('bench_cfg5_n1', 'bench_cfg5_n2'):
    r = json.loads(open('gpurun_out/r02b/%s.json' % f).read().strip().splitlines()[-1])
    print(f, r['value'], r['ms_per_step'], r['output_checksum'])
PY
rm -f $O/*.pt
