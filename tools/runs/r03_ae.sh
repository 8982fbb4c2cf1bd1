#!/bin/bash
# r03_ae: what the atomics of the fused arg-max kernel cost: builds with no atomics (va1), without the read that filters them
# (va2), with plain stores in their place (va4), with workgroup-scope atomics (va8)
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r03_ae.txt
for v in 1 2 4 8; do
  TST_HIP_LIB=abl_out/libtst_va$v.so timeout -k 10 120 python tools/bench_vocab_argmax.py >> gpurun_out/r03_ae.txt 2>&1 || { tail gpurun_out/r03_ae.txt; exit 1; }
done
grep rows gpurun_out/r03_ae.txt | cut -c1-200
