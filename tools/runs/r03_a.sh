#!/bin/bash
# r03_a: do the host-side changes hold on the GPU?  (packed-encoder version check, lagged loop sync, no-alloc free-memory
# assertion, sharded bench fields + oracle sample, config 3 V=32000 full-size parity) + baseline lines for config 2 and config 4
set -o pipefail
mkdir -p gpurun_out/r03a
timeout -k 10 900 python -u -m pytest tests/test_gpu_decode.py tests/test_gpu_sharded.py tests/test_zz_no_alloc.py tests/test_gpu_edge_cases.py -x -q --timeout 300 --timeout-method thread 2>&1 | tail -5 || exit 1
timeout -k 10 600 python -u -m pytest tests/test_gpu_config_sizes.py -x -q --timeout 400 --timeout-method thread -k "config3" 2>&1 | tail -5 || exit 1
timeout -k 10 300 python -u bench.py --no-cpu-baseline > gpurun_out/r03a/bench_cfg2.json 2> gpurun_out/r03a/bench_cfg2.err || exit 1
cut -c1-400 gpurun_out/r03a/bench_cfg2.json
timeout -k 10 300 python -u tools/bench_configs.py cfg4 --dtype bf16 > gpurun_out/r03a/cfg4.jsonl 2>&1 || exit 1
cat gpurun_out/r03a/cfg4.jsonl
