#!/bin/bash
# r03_c: kernel statistics of config 4 with the statistics epilogue on and off
set -o pipefail
bash tools/prof_stats.sh r03_cfg4_stats tools/bench_configs.py cfg4 --dtype bf16 || exit 1
TST_NO_VOCAB_STATS=1 bash tools/prof_stats.sh r03_cfg4_nostats tools/bench_configs.py cfg4 --dtype bf16 || exit 1
head -12 gpurun_out/r03_cfg4_stats_stats.txt; head -12 gpurun_out/r03_cfg4_nostats_stats.txt
