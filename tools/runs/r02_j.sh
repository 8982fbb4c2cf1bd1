#!/bin/bash
set -o pipefail
O=gpurun_out/r02j; mkdir -p $O
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_edge_cases.py -k "pack_sources or without_per_step" -x -q --timeout 120 --timeout-method thread > $O/k.log 2>&1 || { tail -40 $O/k.log; exit 1; }
tail -3 $O/k.log
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method thread > $O/gpu_tests.log 2>&1 || { tail -40 $O/gpu_tests.log; exit 1; }
tail -4 $O/gpu_tests.log
timeout -k 10 300 python -u tools/bench_configs.py cfg1 --sentences 1024 > $O/cfg1.jsonl 2>&1 || { tail -20 $O/cfg1.jsonl; exit 1; }
cat $O/cfg1.jsonl
TST_NO_LAG=1 timeout -k 10 300 python -u tools/bench_configs.py cfg1 --sentences 1024 > $O/cfg1_nolag.jsonl 2>&1 || { tail -20 $O/cfg1_nolag.jsonl; exit 1; }
cat $O/cfg1_nolag.jsonl
