#!/bin/bash
# counters of the weights-in-registers kernel (clock, matrix-pipe busy share) on the layer shapes at 30000 rows, 20 launches per timing
export TST_GEMM_WREG=1 REPS=20
bash tools/prof_util.sh r03_wreg tools/bench_gemm_layers.py 30000
