#!/bin/bash
# r03_as: small batches (config 1, 256 sentences) decoded by several PROCESSES at a time on one GPU, each with 4 streams: does a
# second HIP runtime raise the launch rate that bounds the concurrent-batches regime?
set -o pipefail
mkdir -p gpurun_out/r03_as
for P in 1 2 4; do
  pids=""
  for i in $(seq 1 $P); do
    timeout -k 10 200 python tools/bench_configs.py cfg1 --sentences 256 --streams 4 > gpurun_out/r03_as/p${P}_$i.log 2>&1 &
    pids="$pids $!"
  done
  for p in $pids; do wait $p || { echo "a process of the $P-process run failed"; exit 1; }; done
  echo "processes=$P"
  grep -h '"case"' gpurun_out/r03_as/p${P}_*.log | python3 -c "
import sys, json
tot = 0.0
for l in sys.stdin:
    d = json.loads(l); tot += d['tokens_per_s']; print('  tokens/s %.0f  amortised ms/batch %.2f  latency median %.1f ms' % (d['tokens_per_s'], d['ms_per_batch_amortised'], d['batch_latency_ms_median']))
print('  sum %.0f tokens/s' % tot)"
done
