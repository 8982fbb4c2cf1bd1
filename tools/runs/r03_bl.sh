#!/bin/bash
# r03_bl: the lines of the other configurations on the last tree (ABI 109): config 1 (128 / 256 / 1024 sentences), config 3, config 4 (bf16)
mkdir -p gpurun_out
timeout -k 10 200 python tools/bench_configs.py cfg1 --sentences 1024 2>&1 | grep '"case"' > gpurun_out/r03_cfg1_lines.jsonl || exit 1
timeout -k 10 400 python tools/bench_configs.py cfg3 2>&1 | grep '"case"' > gpurun_out/r03_cfg3_lines.jsonl || exit 1
timeout -k 10 300 python tools/bench_configs.py cfg4 --dtype bf16 2>&1 | grep '"case"' > gpurun_out/r03_cfg4_lines.jsonl || exit 1
cut -c1-175 gpurun_out/r03_cfg1_lines.jsonl gpurun_out/r03_cfg3_lines.jsonl gpurun_out/r03_cfg4_lines.jsonl
