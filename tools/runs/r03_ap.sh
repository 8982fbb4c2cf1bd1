#!/bin/bash
# r03_ap: with the per-token table, from how many rows is the one-kernel GRU cell ahead of recurrent product + update kernel?
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r03_ap.txt
for n in 512 1024 2048 4096 8192 100000000; do
  echo "TST_GRU_FUSED_MIN_ROWS=$n" >> gpurun_out/r03_ap.txt
  TST_GRU_FUSED_MIN_ROWS=$n timeout -k 10 500 python tools/bench_configs.py cfg3 2>&1 | grep -E '"case"' | grep "V=128" | cut -c10-170 >> gpurun_out/r03_ap.txt || exit 1
  TST_GRU_FUSED_MIN_ROWS=$n timeout -k 10 500 python tools/bench_configs.py cfg3 --sentences 1024 2>&1 | grep -E '"case"' | grep "V=128" | cut -c10-170 >> gpurun_out/r03_ap.txt || exit 1
done
cat gpurun_out/r03_ap.txt
