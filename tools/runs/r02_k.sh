#!/bin/bash
set -o pipefail
bash tools/prof_stats.sh r02_cfg1_256_sentences_kernel tools/bench_configs.py cfg1 --sentences 256 > gpurun_out/r02k.log 2>&1 || { tail gpurun_out/r02k.log; exit 1; }
head -45 gpurun_out/r02_cfg1_256_sentences_kernel_stats.txt
