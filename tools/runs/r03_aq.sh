#!/bin/bash
# r03_aq: whole GPU suite, smoke and the config 3 lines + kernel statistics with the per-token table of the GRU's layer 0 (ABI 108)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r03_aq_tests.log 2>&1 || { tail -30 gpurun_out/r03_aq_tests.log; exit 1; }
tail -2 gpurun_out/r03_aq_tests.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 || exit 1
timeout -k 10 500 python tools/bench_configs.py cfg3 2>&1 | grep -E '"case"' > gpurun_out/r03_cfg3_lines.jsonl || exit 1
cut -c1-230 gpurun_out/r03_cfg3_lines.jsonl
out=gpurun_out/prof_aq
rm -rf $out && mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/t -- python3 tools/bench_configs.py cfg3 > $out/log.txt 2>&1 || { tail -20 $out/log.txt; exit 1; }
python3 tools/summarise_prof.py stats $out/t > gpurun_out/r03_cfg3_8192_sentences_kernel_stats.txt
rm -rf $out/t
head -14 gpurun_out/r03_cfg3_8192_sentences_kernel_stats.txt | cut -c1-150
