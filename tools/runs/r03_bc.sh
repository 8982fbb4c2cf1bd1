#!/bin/bash
# config 2 with the K = 256 layer products on the weights-in-registers kernel (TST_GEMM_WREG=1) against the tile kernels, alternating
for v in 0 1 0 1; do
  TST_GEMM_WREG=$v timeout -k 10 200 python bench.py --no-cpu-baseline --no-roofline --steps 8 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('TST_GEMM_WREG=$v', round(d['ms_per_step'],2), round(d['value']), d['output_checksum'])" || exit 1
done
