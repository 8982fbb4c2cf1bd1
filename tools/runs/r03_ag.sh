#!/bin/bash
# r03_ag: config 1 with 1 / 4 / 8 concurrent 256-sentence batches, the fused arg-max tail and the host-visible count on and off
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r03_ag.txt
for st in 4 8; do
  for sw in "" "TST_NO_ARGMAX_KEYS=1 TST_NO_HOST_COUNT=1" "" "TST_NO_ARGMAX_KEYS=1 TST_NO_HOST_COUNT=1"; do
    echo "streams=$st switches=[$sw]" >> gpurun_out/r03_ag.txt
    env $sw timeout -k 10 200 python tools/bench_configs.py cfg1 --sentences 256 --streams $st 2>&1 | grep '"case"' | cut -c1-400 >> gpurun_out/r03_ag.txt || exit 1
  done
done
cat gpurun_out/r03_ag.txt
