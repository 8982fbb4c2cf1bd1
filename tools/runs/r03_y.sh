#!/bin/bash
# r03_y: embedding + conditioning vector in one pass: parity (kernel, LM decode fixtures, config 4 sizes), config 4 line
set -o pipefail
timeout -k 10 400 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_lm_bf16.py tests/test_gpu_decode.py tests/test_gpu_config_sizes.py -x -q -k "embed or lm or tlm or config4 or bf16" --timeout 200 --timeout-method thread 2>&1 | tail -3 || exit 1
for i in 1 2; do timeout -k 10 300 python tools/bench_configs.py cfg4 --dtype bf16 2>&1 | grep '"case"' | cut -c1-220 || exit 1; done
