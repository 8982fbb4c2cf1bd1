#!/bin/bash
# r03_l: config 1 (256 / 128 sentences, greedy): one batch at a time against 2, 4, 8 concurrent batches on streams of their own
set -o pipefail
{
timeout -k 10 120 python tools/bench_configs.py cfg1 --sentences 256 || exit 1
for s in 2 4 8; do timeout -k 10 120 python tools/bench_configs.py cfg1 --sentences 256 --streams $s || exit 1; done
for s in 4 8; do timeout -k 10 120 python tools/bench_configs.py cfg1 --sentences 128 --streams $s || exit 1; done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_l_cfg1_streams.jsonl
