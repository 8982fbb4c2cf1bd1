#!/bin/bash
# r03_i: SQ counters of the chain kernel at 16384 rows (what its k loops spend their cycles on)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03_i
rm -rf $out && mkdir -p $out
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"
timeout -s KILL 120 rocprofv3 --pmc $C -d $out/sq -- python3 tools/bench_chain.py 16384 > $out/sq.log 2>&1 || { tail -5 $out/sq.log; exit 1; }
for c in $C; do python3 tools/summarise_prof.py pmc $out/sq $c | grep -E "chain_kernel|counter|kernel " | head -6; done > gpurun_out/r03_i_chain_pmc.txt
rm -rf $out/sq
cat gpurun_out/r03_i_chain_pmc.txt
