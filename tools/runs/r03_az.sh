#!/bin/bash
# config 4 as 3 concurrent sub-batches runs 12 x slower than as 1 or 2 (without the profiler): which part is it?
echo "fp32, 3 parts (2048 sentences)"; timeout -k 10 300 python tools/bench_configs.py cfg4 --dtype fp32 --parts 3 --sentences 2048 2>&1 | grep '"case"' | cut -c60-200 || exit 1
echo "fp32, 1 part (2048 sentences)"; timeout -k 10 300 python tools/bench_configs.py cfg4 --dtype fp32 --parts 1 --sentences 2048 2>&1 | grep '"case"' | cut -c60-200 || exit 1
echo "bf16, 3 parts, 8 hardware queues"; GPU_MAX_HW_QUEUES=8 timeout -k 10 300 python tools/bench_configs.py cfg4 --dtype bf16 --parts 3 2>&1 | grep '"case"' | cut -c60-200 || exit 1
echo "bf16, 3 parts, non-persistent vocabulary projection"; TST_BF16_NO_PERSIST=1 timeout -k 10 300 python tools/bench_configs.py cfg4 --dtype bf16 --parts 3 2>&1 | grep '"case"' | cut -c60-200 || exit 1
