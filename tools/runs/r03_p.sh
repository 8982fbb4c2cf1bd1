#!/bin/bash
# r03_p: persistent form of the fp32 vocabulary projection (variant 4): parity, micro-benchmark against the one-tile-per-workgroup
# kernel, config 2 end to end with either (TST_VOCAB_VARIANT)
set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py -x -q -k "gemm_dma" --timeout 120 --timeout-method thread 2>&1 | tail -3 || exit 1
timeout -k 10 200 python tools/bench_gemm_dma.py 32768 15676 8192 4096 2>&1 | grep -v amdgpu.ids | sed 's/| dma v0.*| dma v3/| dma v3/' | tee gpurun_out/r03_p_microbench.txt || exit 1
for v in 3 4 3 4; do
  TST_VOCAB_VARIANT=$v timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/r03_p_v$v.json 2>gpurun_out/r03_p.err || { tail gpurun_out/r03_p.err; exit 1; }
  python - <<PY
import json
d = json.loads(open('gpurun_out/r03_p_v$v.json').read().strip().splitlines()[-1])
print('variant $v ms_per_step %.2f tokens/s %.0f vocab %.1f TF checksum %d' % (d['ms_per_step'], d['value'], d['roofline']['achieved'], d['output_checksum']))
PY
done
