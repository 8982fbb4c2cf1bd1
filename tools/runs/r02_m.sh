#!/bin/bash
set -o pipefail
O=gpurun_out/r02m; mkdir -p $O
timeout -k 10 400 python -u bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -30 $O/bench_default.err; exit 1; }
cat $O/bench_default.json
bash tools/make_profiles.sh r02 > $O/prof.log 2>&1 || { tail -20 $O/prof.log; exit 1; }
cat gpurun_out/prof_r02/r02_bench_kernel_stats.txt | head -30
grep -i "vocabulary\|logsoftmax" gpurun_out/prof_r02/r02_bench_pmc_fetch_size.txt gpurun_out/prof_r02/r02_bench_pmc_write_size.txt
timeout -k 10 300 python -u bench.py --config 5 --sentences 65536 --no-cpu-baseline > $O/bench_cfg5_64k.json 2> $O/bench_cfg5.err || { tail -30 $O/bench_cfg5.err; exit 1; }
cat $O/bench_cfg5_64k.json
