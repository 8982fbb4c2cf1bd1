#!/bin/bash
# r03_ai: is a single 256-sentence batch bound by the GPU or by the host?  kernel-trace of three batches: share of the span of the
# last one in which a kernel was running, and the gaps between consecutive kernels
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_ai
rm -rf $out && mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace -d $out/t -- python3 tools/probe/cfg1_once.py > $out/log.txt 2>&1 || { tail -20 $out/log.txt; exit 1; }
python3 tools/probe/gpu_busy.py $out/t 1 | tee $out/busy.txt
rm -rf $out/t
