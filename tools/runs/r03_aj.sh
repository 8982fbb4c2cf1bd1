#!/bin/bash
# r03_aj: rows up to which the tail of a step is one workgroup (160) against 256 / 512 on config 1
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r03_aj.txt
for n in 160 256 512 160 256 512; do
  echo "TST_SMALL_TAIL_ROWS=$n" >> gpurun_out/r03_aj.txt
  TST_SMALL_TAIL_ROWS=$n timeout -k 10 200 python tools/bench_configs.py cfg1 --sentences 512 2>&1 | grep '"case"' | cut -c80-190 >> gpurun_out/r03_aj.txt || exit 1
done
cat gpurun_out/r03_aj.txt
