#!/bin/bash
# r03_q: the config-2 batch as 2 / 3 concurrent sub-batches on streams of their own, on the current tree
set -o pipefail
for p in 2 3; do timeout -k 10 200 python tools/probe/two_threads.py $p 2>&1 | grep -v amdgpu.ids | tail -3 || exit 1; done | tee gpurun_out/r03_q_two_threads.txt
