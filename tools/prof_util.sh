#!/bin/bash
# Runs on the GPU box: FETCH_SIZE, WRITE_SIZE and SQ counter passes (a run each) of one python command, per-kernel table of
# matrix-pipe busy share, clock and fabric GB/s in gpurun_out/<tag>_utilisation.txt.
#   tools/prof_util.sh <tag> <script.py> [args...]
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/util_$tag
rm -rf $out && mkdir -p $out
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $out/fetch -- python3 "$@" > $out/fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $out/write -- python3 "$@" > $out/write.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $out/sq -- python3 "$@" > $out/sq.log 2>&1
python3 tools/summarise_prof.py util $out/sq $out/fetch $out/write > gpurun_out/${tag}_utilisation.txt
rm -rf $out/fetch $out/write $out/sq
head -24 gpurun_out/${tag}_utilisation.txt
