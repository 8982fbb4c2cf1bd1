#!/bin/bash
# r03_e: kernel statistics of config 3 (which projection kernel runs, and how long)
set -o pipefail
bash tools/prof_stats.sh r03_cfg3 tools/bench_configs.py cfg3 || exit 1
head -14 gpurun_out/r03_cfg3_stats.txt
