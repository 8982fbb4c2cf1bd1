#!/bin/bash
# Builds libtst_hip.so (gfx950) next to the Python package. No fast-math, no fp contraction: the kernels
# rely on a fixed fp32 operation order.
set -e
cd "$(dirname "$0")"
OUT=../libtst_hip.so
FLAGS="--offload-arch=gfx950 -O3 -fPIC -ffp-contract=off -std=c++17 -Wall -Wno-unused-result"
for f in capi gemm_f32 gemm_bf16 rowops attn select gumbel step; do
  if [ ! -f $f.o ] || [ $f.hip -nt $f.o ] || [ tst_common.h -nt $f.o ] || [ ../../include/tst_hip.h -nt $f.o ]; then
    /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o $f.o &
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT capi.o gemm_f32.o gemm_bf16.o rowops.o attn.o select.o gumbel.o step.o
echo "built $OUT"
