#!/bin/bash
# r03_ak: records on the final tree (ABI 108): GPU suite, smoke, default bench line, rocprofv3 profile set of the bench command,
# config 1 / 4 lines
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r03_ak_tests.log 2>&1 || { tail -30 gpurun_out/r03_ak_tests.log; exit 1; }
tail -2 gpurun_out/r03_ak_tests.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 || exit 1
timeout -k 10 300 python bench.py > gpurun_out/r03_bench_default_line.json 2>gpurun_out/r03_ak_bench.err || { tail -30 gpurun_out/r03_ak_bench.err; exit 1; }
python -c "import json; d=json.loads(open('gpurun_out/r03_bench_default_line.json').read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], d['value'], d['roofline']['achieved'], d['roofline_select']['achieved'], d['peaks_measured'])"
bash tools/make_profiles.sh r03 > gpurun_out/r03_ak_prof.log 2>&1 || { tail -20 gpurun_out/r03_ak_prof.log; exit 1; }
head -12 gpurun_out/prof_r03/r03_bench_kernel_stats.txt
timeout -k 10 200 python tools/bench_configs.py cfg1 --sentences 256 2>&1 | grep '"case"' > gpurun_out/r03_cfg1_lines.jsonl || exit 1
timeout -k 10 300 python tools/bench_configs.py cfg4 --dtype bf16 2>&1 | grep '"case"' > gpurun_out/r03_cfg4_lines.jsonl || exit 1
cut -c1-200 gpurun_out/r03_cfg1_lines.jsonl gpurun_out/r03_cfg4_lines.jsonl
