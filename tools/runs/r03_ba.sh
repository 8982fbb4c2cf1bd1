#!/bin/bash
# r03_ba: the headline profile set (kernel statistics, FETCH_SIZE / WRITE_SIZE, SQ counters of bench.py) again on the last tree
bash tools/make_profiles.sh r03 && head -12 gpurun_out/prof_r03/r03_bench_kernel_stats.txt | cut -c1-170
