#!/bin/bash
# r03_b: vocabulary projection with selection statistics in its epilogue (bf16): parity tests, then config 4 with and without
set -o pipefail
mkdir -p gpurun_out/r03b
timeout -k 10 600 python -u -m pytest tests/test_gpu_lm_bf16.py -x -q --timeout 300 --timeout-method thread 2>&1 | tail -8 || exit 1
timeout -k 10 300 python -u tools/bench_configs.py cfg4 --dtype bf16 > gpurun_out/r03b/cfg4_stats.jsonl 2>&1 || exit 1
cat gpurun_out/r03b/cfg4_stats.jsonl
TST_NO_VOCAB_STATS=1 timeout -k 10 300 python -u tools/bench_configs.py cfg4 --dtype bf16 > gpurun_out/r03b/cfg4_nostats.jsonl 2>&1 || exit 1
cat gpurun_out/r03b/cfg4_nostats.jsonl
