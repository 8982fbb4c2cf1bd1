#!/bin/bash
# r03_at: the embedding of a small fp32 step formed inside layer 0's LayerNorm + q|k|v kernel (one launch less): decode tests,
# config 1 with it on / off (TST_NO_EMBED_FUSE=1), 1 and 8 concurrent batches
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_gpu_argmax_keys.py tests/test_gpu_edge_cases.py tests/test_gpu_decode.py tests/test_gpu_fuzz.py -x -q --timeout 300 --timeout-method thread 2>&1 | tail -3 || exit 1
: > gpurun_out/r03_at.txt
for sw in "" "TST_NO_EMBED_FUSE=1" "" "TST_NO_EMBED_FUSE=1"; do
  echo "[$sw]" >> gpurun_out/r03_at.txt
  env $sw timeout -k 10 200 python tools/bench_configs.py cfg1 --sentences 256 2>&1 | grep '"case"' | cut -c80-190 >> gpurun_out/r03_at.txt || exit 1
  env $sw timeout -k 10 200 python tools/bench_configs.py cfg1 --sentences 256 --streams 8 2>&1 | grep '"case"' | cut -c100-330 >> gpurun_out/r03_at.txt || exit 1
done
cat gpurun_out/r03_at.txt
