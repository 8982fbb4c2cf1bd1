#!/bin/bash
# Runs on the GPU box: kernel-trace statistics of one python command, summary in gpurun_out/<tag>_stats.txt.
#   tools/prof_stats.sh <tag> <script.py> [args...]
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
rm -rf $out && mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/trace -- python3 "$@" > $out/trace.log 2>&1
python3 tools/summarise_prof.py stats $out/trace > gpurun_out/${tag}_stats.txt
tail -3 $out/trace.log
rm -rf $out/trace
