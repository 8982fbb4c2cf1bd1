#!/bin/bash
# Builds libtst_hip.so (gfx950) next to the Python package. No fast-math, no fp contraction: the kernels
# rely on a fixed fp32 operation order.  A translation unit that fails to compile fails the build: its stale
# object is removed first, and every compile job's status is checked.
set -e
cd "$(dirname "$0")"
OUT=../libtst_hip.so
FLAGS="--offload-arch=gfx950 -O3 -fPIC -ffp-contract=off -std=c++17 -Wall -Wno-unused-result"
UNITS="capi gemm_f32 gemm_f32_wreg gemm_f32_dma chain_f32 gemm_bf16 gru_fused rowops attn attn_mfma select gumbel step"
pids=()
for f in $UNITS; do
  if [ ! -f $f.o ] || [ $f.hip -nt $f.o ] || [ tst_common.h -nt $f.o ] || [ ../../include/tst_hip.h -nt $f.o ]; then
    rm -f $f.o
    /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o $f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do
  wait "$p" || { echo "build.sh: a compile job failed" >&2; exit 1; }
done
OBJS=""
for f in $UNITS; do OBJS="$OBJS $f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJS
echo "built $OUT"
