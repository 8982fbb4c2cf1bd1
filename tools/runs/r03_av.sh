#!/bin/bash
# a longer sweep of the seeded random-configuration tests than the suite runs: $1 = first seed, $2 = multiple of the suite's count per test
mkdir -p gpurun_out
TST_FUZZ_FIRST=${1:-1000} TST_FUZZ_SCALE=${2:-10} timeout -k 10 1000 python -u -m pytest tests/test_gpu_fuzz.py -q --timeout 120 --timeout-method thread -p no:cacheprovider > gpurun_out/r03_fuzz_sweep.txt 2>&1
rc=$?
tail -15 gpurun_out/r03_fuzz_sweep.txt
exit $rc
