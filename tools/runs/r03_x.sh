#!/bin/bash
# r03_x: utilisation table (matrix-pipe busy share, clock, fabric GB/s per kernel) of config 4 (bf16) on the final tree
set -o pipefail
bash tools/prof_util.sh r03_cfg4_bf16 tools/bench_configs.py cfg4 --dtype bf16 || exit 1
