#!/bin/bash
# r03_ah: one-wave strip kernel with chunks requested two ahead (K >= 512, one-column strips) and the residual values requested
# before the k loop, against the build before it (abl_out/libtst_base.so): parity, small-GEMM timings, config 1
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_kernels.py -x -q -k "gemm" --timeout 300 --timeout-method thread 2>&1 | tail -2 || exit 1
timeout -k 10 200 python tools/bench_gemm_small.py 64,128,256 2>&1 | grep -E "proj|ff2" | cut -c1-90 > gpurun_out/r03_ah_new.txt || exit 1
TST_HIP_LIB=abl_out/libtst_base.so timeout -k 10 200 python tools/bench_gemm_small.py 64,128,256 2>&1 | grep -E "proj|ff2" | cut -c1-90 > gpurun_out/r03_ah_base.txt || exit 1
paste -d'\n' gpurun_out/r03_ah_new.txt gpurun_out/r03_ah_base.txt
for lib in "" "TST_HIP_LIB=abl_out/libtst_base.so" "" "TST_HIP_LIB=abl_out/libtst_base.so"; do
  echo "[$lib]"
  env $lib timeout -k 10 200 python tools/bench_configs.py cfg1 --sentences 256 2>&1 | grep '"case"' | cut -c80-190 || exit 1
done
