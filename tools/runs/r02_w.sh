#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_layers; rm -rf $out; mkdir -p $out
C1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
C2="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE"
timeout -s KILL 120 rocprofv3 --pmc $C1 -d $out/a -- python3 tools/bench_gemm_layers.py 15676 > $out/a.log 2>&1 || { tail -5 $out/a.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc $C2 -d $out/b -- python3 tools/bench_gemm_layers.py 15676 > $out/b.log 2>&1 || { tail -5 $out/b.log; exit 1; }
for c in $C1; do python3 tools/summarise_prof.py pmc $out/a $c | grep "gemm_f32_kernel" | awk -v c=$c '{print c, $0}'; done
for c in $C2; do python3 tools/summarise_prof.py pmc $out/b $c | grep "gemm_f32_kernel" | awk -v c=$c '{print c, $0}'; done
rm -rf $out/a $out/b
