#!/bin/bash
# L2 / L1 counters of the bf16 vocabulary projection
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_bf16_l2; rm -rf $out; mkdir -p $out
C="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_BUSY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE"
timeout -s KILL 120 rocprofv3 --pmc $C -d $out/a -- python3 tools/bench_gemm_bf16.py 65536 vocab > $out/a.log 2>&1 || { tail -5 $out/a.log; exit 1; }
grep median $out/a.log
for c in $C; do python3 tools/summarise_prof.py pmc $out/a $c | grep -i "gemm_bf16" | head -1 | awk -v c=$c '{print c, $0}'; done
rm -rf $out/a
