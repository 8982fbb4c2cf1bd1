#!/bin/bash
# r03_s: corpus walk by two staggered walkers (tail steps of a batch beside the large steps of the next) against a sequential walk
set -o pipefail
for d in 0 60 88 120; do timeout -k 10 200 python tools/probe/staggered_batches.py 8 $d 2>&1 | grep -v amdgpu.ids | tail -1 || exit 1; done | tee gpurun_out/r03_s_staggered.txt
