#!/bin/bash
# r03_h: what the chain kernel's k loops wait for: variants without DMA waits / DMA / operand reads / LDS waits
set -o pipefail
for a in 0 4 7; do
  lib=abl_out/libtst_abl$a.so
  [ $a = 0 ] && lib=text-style-transfer_amd/libtst_hip.so
  echo "== CHAIN_ABL=$a"
  TST_HIP_LIB=$PWD/$lib timeout -k 10 120 python tools/bench_chain.py 2048 16384 2>/dev/null | grep -E "ff1|q\|k\|v" || exit 1
done
