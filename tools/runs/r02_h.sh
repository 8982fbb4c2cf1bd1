#!/bin/bash
set -o pipefail
O=gpurun_out/r02h; mkdir -p $O
bash tools/make_profiles.sh r02h > $O/prof.log 2>&1 || { tail -20 $O/prof.log; exit 1; }
head -12 gpurun_out/prof_r02h/r02h_bench_kernel_stats.txt
cat gpurun_out/prof_r02h/r02h_bench_line_under_rocprof.json | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'])"
