#!/bin/bash
# one GPU call: no-alloc test, full GPU suite, peaks probe, bench lines (config 2, config 5 at N=1 and a 2-rank rehearsal)
set -o pipefail
mkdir -p gpurun_out/r02a
O=gpurun_out/r02a
timeout -k 10 300 python -u -m pytest tests/wip_no_alloc.py -x -v --timeout 120 --timeout-method thread > $O/noalloc.log 2>&1 || { tail -30 $O/noalloc.log; exit 1; }
tail -3 $O/noalloc.log
timeout -k 10 60 ./tools/peaks > $O/peaks.json 2>&1 || { cat $O/peaks.json; exit 1; }
cat $O/peaks.json
timeout -k 10 400 python -u bench.py --steps 5 --warmup 1 > $O/bench_cfg2.json 2> $O/bench_cfg2.err || { tail -30 $O/bench_cfg2.err; exit 1; }
cat $O/bench_cfg2.json
timeout -k 10 300 python -u bench.py --config 5 --sentences 32768 --no-cpu-baseline --dump $O/c5_n1.pt > $O/bench_cfg5_n1.json 2> $O/bench_cfg5_n1.err || { tail -30 $O/bench_cfg5_n1.err; exit 1; }
cat $O/bench_cfg5_n1.json
timeout -k 10 300 python -u bench.py --gpus 2 --share-gpu --backend gloo --config 5 --sentences 32768 --dump $O/c5_n2.pt > $O/bench_cfg5_n2.json 2> $O/bench_cfg5_n2.err || { tail -30 $O/bench_cfg5_n2.err; exit 1; }
cat $O/bench_cfg5_n2.json
python - <<'PY'
import torch
a = torch.load('gpurun_out/r02a/c5_n1.pt'); b = torch.load('gpurun_out/r02a/c5_n2.pt')
print('config 5 outputs N=1 vs N=2 equal:', torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]))
PY
rm -f $O/*.pt
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/gpu_tests.log 2>&1; rc=$?
tail -5 $O/gpu_tests.log
exit $rc
