#!/bin/bash
# r03_t: residual product + LayerNorm epilogue (bf16, whole-row tiles): parity, then config 4 with it from 16384 / 32768 rows on
# against the default path (TST_BF16_LN_FUSE), alternating
set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_gpu_lm_bf16.py -x -q -k "residual_product" --timeout 120 --timeout-method thread 2>&1 | tail -5 || exit 1
for f in 0 16384 0 16384 32768 12288; do
  echo "TST_BF16_LN_FUSE=$f"
  TST_BF16_LN_FUSE=$f timeout -k 10 300 python tools/bench_configs.py cfg4 --dtype bf16 2>&1 | grep '"case"' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('  tokens/s %.0f  ms_per_batch %.2f  row_steps %d  emitted %.4f' % (d['tokens_per_s'], d['ms_per_batch'], d['row_steps'], d['mean_emitted_len']))" || exit 1
done | tee gpurun_out/r03_t_ln_fuse.txt
