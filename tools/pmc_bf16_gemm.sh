#!/bin/bash
# Runs on the GPU box: SQ / GRBM counters of the bf16 vocabulary projection (one pass), per-dispatch averages in
# gpurun_out/<tag>_bf16_gemm_pmc_sq.txt
set -e
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_bf16_$tag
rm -rf $out && mkdir -p $out
C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
# second pass: the LDS array and the vector-memory path
C2="SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INST_CYCLES_VMEM_RD TA_TA_BUSY_sum GRBM_GUI_ACTIVE"
timeout -s KILL 120 rocprofv3 --pmc $C -d $out/sq -- python3 tools/bench_gemm_bf16.py 65536 vocab > $out/sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc $C2 -d $out/sq2 -- python3 tools/bench_gemm_bf16.py 65536 vocab > $out/sq2.log 2>&1
{
  echo "rocprofv3 --pmc $C -- python3 tools/bench_gemm_bf16.py 65536 vocab"
  grep median $out/sq.log
  for c in $C; do python3 tools/summarise_prof.py pmc $out/sq $c | grep -i "gemm_bf16" | head -1 | awk -v c=$c '{print c, $0}'; done
  echo "rocprofv3 --pmc $C2 -- python3 tools/bench_gemm_bf16.py 65536 vocab"
  grep median $out/sq2.log
  for c in $C2; do python3 tools/summarise_prof.py pmc $out/sq2 $c | grep -i "gemm_bf16" | head -1 | awk -v c=$c '{print c, $0}'; done
} > gpurun_out/${tag}_bf16_gemm_pmc_sq.txt
rm -rf $out/sq $out/sq2
cat gpurun_out/${tag}_bf16_gemm_pmc_sq.txt
