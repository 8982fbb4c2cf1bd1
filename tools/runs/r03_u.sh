#!/bin/bash
# r03_u: kernel statistics of config 4 with the LayerNorm epilogue on from 16384 rows (the negative's kernel times)
set -o pipefail
TST_BF16_LN_FUSE=16384 bash tools/prof_stats.sh r03_cfg4_ln_fuse tools/bench_configs.py cfg4 --dtype bf16 || exit 1
head -16 gpurun_out/r03_cfg4_ln_fuse_stats.txt
