#!/bin/bash
set -o pipefail
timeout -k 10 600 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread 2>&1 | tail -2 || exit 1
for sw in 0 1 0 1; do
  echo "TST_GEMM_NO_K32=$sw"
  TST_GEMM_NO_K32=$sw REPS=3 timeout -k 10 200 python -u tools/probe/e2e_cfg2.py | tail -2 || exit 1
done
