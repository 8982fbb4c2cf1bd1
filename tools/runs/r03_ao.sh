#!/bin/bash
# r03_ao: GRU layer 0 with its input pre-activations looked up per token (gi table): parity tests, config 3 with the table on / off
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_gpu_gi_table.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r03_ao_tests.log 2>&1 || { tail -40 gpurun_out/r03_ao_tests.log; exit 1; }
tail -2 gpurun_out/r03_ao_tests.log
: > gpurun_out/r03_ao.txt
for sw in "" "TST_NO_GI_TABLE=1"; do
  echo "[$sw]" >> gpurun_out/r03_ao.txt
  env $sw timeout -k 10 500 python tools/bench_configs.py cfg3 2>&1 | grep -E '"case"' | cut -c1-260 >> gpurun_out/r03_ao.txt || exit 1
  env $sw SENTENCES=128 timeout -k 10 100 python tools/probe/cfg3_small.py 2>&1 | grep batch >> gpurun_out/r03_ao.txt || exit 1
done
cat gpurun_out/r03_ao.txt
