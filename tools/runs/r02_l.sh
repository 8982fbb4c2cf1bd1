#!/bin/bash
set -o pipefail
bash tools/prof_stats.sh r02_cfg4_bf16_kernel tools/bench_configs.py cfg4 --dtype bf16 > gpurun_out/r02l.log 2>&1 || { tail gpurun_out/r02l.log; exit 1; }
head -24 gpurun_out/r02_cfg4_bf16_kernel_stats.txt
grep '"case"' gpurun_out/prof_r02_cfg4_bf16_kernel/trace.log
