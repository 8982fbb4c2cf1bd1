#!/bin/bash
# clock and matrix-pipe busy share of every variant of tools/probe/mfma_feed.hip (one counter pass)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_mfma_feed
rm -rf $out && mkdir -p $out
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d $out/sq -- tools/probe/bin/mfma_feed > $out/log.txt 2>&1 || { tail -5 $out/log.txt; exit 1; }
grep TFLOP $out/log.txt
python3 tools/summarise_prof.py util $out/sq $out/sq $out/sq | cut -c1-140
rm -rf $out/sq
