#!/bin/bash
set -o pipefail
O=gpurun_out/r02d; mkdir -p $O
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py -k gru -x -q --timeout 120 --timeout-method thread > $O/gru_tests.log 2>&1 || { tail -40 $O/gru_tests.log; exit 1; }
tail -3 $O/gru_tests.log
timeout -k 10 200 python -u tools/bench_gru_cell.py > $O/gru_cell.jsonl 2>&1 || { tail -20 $O/gru_cell.jsonl; exit 1; }
cat $O/gru_cell.jsonl
timeout -k 10 300 python -u tools/bench_configs.py cfg3 > $O/cfg3_fused.jsonl 2>&1 || { tail -20 $O/cfg3_fused.jsonl; exit 1; }
cat $O/cfg3_fused.jsonl
TST_GRU_FUSED_MIN_ROWS=100000000 timeout -k 10 300 python -u tools/bench_configs.py cfg3 > $O/cfg3_three_call.jsonl 2>&1 || { tail -20 $O/cfg3_three_call.jsonl; exit 1; }
cat $O/cfg3_three_call.jsonl
