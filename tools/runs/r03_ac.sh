#!/bin/bash
# r03_ac: kernel statistics of config 1 (256 sentences) with the fused LayerNorm + projection + arg-max kernel on and off
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_ac
rm -rf $out && mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/on -- python3 tools/bench_configs.py cfg1 --sentences 256 > $out/on.log 2>&1 || { tail -20 $out/on.log; exit 1; }
python3 tools/summarise_prof.py stats $out/on > $out/r03_cfg1_keys_on_kernel_stats.txt
export TST_NO_ARGMAX_KEYS=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/off -- python3 tools/bench_configs.py cfg1 --sentences 256 > $out/off.log 2>&1 || { tail -20 $out/off.log; exit 1; }
python3 tools/summarise_prof.py stats $out/off > $out/r03_cfg1_keys_off_kernel_stats.txt
rm -rf $out/on $out/off
head -16 $out/r03_cfg1_keys_on_kernel_stats.txt | cut -c1-150; head -16 $out/r03_cfg1_keys_off_kernel_stats.txt | cut -c1-150
