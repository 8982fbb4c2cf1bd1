#!/bin/bash
set -o pipefail
timeout -k 10 300 python -u bench.py > gpurun_out/r02_final_bench.json || exit 1
bash tools/make_profiles.sh r02 > gpurun_out/mp.log 2>&1 || { tail gpurun_out/mp.log; exit 1; }
head -12 gpurun_out/prof_r02/r02_bench_kernel_stats.txt
bash tools/prof_stats.sh r02_cfg1_256_sentences_kernel tools/bench_configs.py cfg1 --sentences 256 > gpurun_out/p1.log 2>&1 || { tail gpurun_out/p1.log; exit 1; }
head -8 gpurun_out/r02_cfg1_256_sentences_kernel_stats.txt
