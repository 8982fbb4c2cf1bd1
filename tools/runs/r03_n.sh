#!/bin/bash
# r03_n: register-resident LayerNorm (rows of 256 / 512 floats): parity, then config 4 (bf16) throughput and kernel statistics
set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_lm_bf16.py -x -q -k "layernorm or bf16" --timeout 200 --timeout-method thread 2>&1 | tail -3 || exit 1
timeout -k 10 300 python tools/bench_configs.py cfg4 --dtype bf16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_n_cfg4_line.jsonl || exit 1
bash tools/prof_stats.sh r03_cfg4 tools/bench_configs.py cfg4 --dtype bf16 || exit 1
head -14 gpurun_out/r03_cfg4_stats.txt
