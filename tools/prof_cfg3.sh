#!/bin/bash
# Runs on the GPU box: config 3 (GRU decoder E = H = 1024, 8192 sentences, V = 128 and 32000, greedy and beam 4) under
# rocprofv3: kernel-trace statistics of the four decodes, then SQ / GRBM counters and FETCH_SIZE / WRITE_SIZE of the
# one-kernel GRU cell (tools/bench_gru_cell.py) in passes of their own.  Summaries in gpurun_out/<tag>_cfg3_*.txt
set -e
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_cfg3_$tag
rm -rf $out && mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace -- python3 tools/bench_configs.py cfg3 > $out/trace.log 2>&1
python3 tools/summarise_prof.py stats $out/trace > gpurun_out/${tag}_cfg3_8192_sentences_kernel_stats.txt
grep -h '"case"\|"kernel"' $out/trace.log > gpurun_out/${tag}_cfg3_lines_under_rocprof.jsonl
C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
timeout -s KILL 120 rocprofv3 --pmc $C -d $out/sq -- python3 tools/bench_gru_cell.py > $out/sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $out/fetch -- python3 tools/bench_gru_cell.py > $out/fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $out/write -- python3 tools/bench_gru_cell.py > $out/write.log 2>&1
{
  echo "rocprofv3 --pmc <counters> -- python3 tools/bench_gru_cell.py   (sums over all dispatches of the kernel: 21 launches at each of 128 ... 32768 rows)"
  for c in $C; do python3 tools/summarise_prof.py pmc $out/sq $c | grep -i "gru_fused" | head -1 | awk -v c=$c '{print c, $0}'; done
  python3 tools/summarise_prof.py pmc $out/fetch FETCH_SIZE | grep -i "gru_fused" | head -1 | awk '{print "FETCH_SIZE(KiB; x2 for 16-byte streaming reads)", $0}'
  python3 tools/summarise_prof.py pmc $out/write WRITE_SIZE | grep -i "gru_fused" | head -1 | awk '{print "WRITE_SIZE(KiB)", $0}'
} > gpurun_out/${tag}_cfg3_gru_cell_pmc.txt
rm -rf $out
cat gpurun_out/${tag}_cfg3_8192_sentences_kernel_stats.txt | head -30
cat gpurun_out/${tag}_cfg3_gru_cell_pmc.txt
cat gpurun_out/${tag}_cfg3_lines_under_rocprof.jsonl
