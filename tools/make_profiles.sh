#!/bin/bash
# Runs on the GPU box: rocprofv3 passes over one bench.py command (kernel trace, then FETCH_SIZE, WRITE_SIZE and the SQ
# counters in runs of their own) and text summaries under gpurun_out/; copy the summaries into profiles/ afterwards.
#   tools/make_profiles.sh <tag>
set -e
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline"
out=gpurun_out/prof_$tag
rm -rf $out && mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace -- $B > $out/trace.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $out/fetch -- $B > $out/fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $out/write -- $B > $out/write.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $out/sq -- $B > $out/sq.log 2>&1
python3 tools/summarise_prof.py util $out/sq $out/fetch $out/write > $out/${tag}_bench_utilisation.txt
python3 tools/summarise_prof.py stats $out/trace > $out/${tag}_bench_kernel_stats.txt
python3 tools/summarise_prof.py pmc $out/fetch FETCH_SIZE > $out/${tag}_bench_pmc_fetch_size.txt
python3 tools/summarise_prof.py pmc $out/write WRITE_SIZE > $out/${tag}_bench_pmc_write_size.txt
grep -h '"metric"' $out/trace.log > $out/${tag}_bench_line_under_rocprof.json
rm -rf $out/trace $out/fetch $out/write $out/sq
ls -la $out
