#!/bin/bash
# r03_ax: whole GPU suite as the driver runs it (120 s per test), smoke and the default bench line on the final tree
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r03_ax_tests.log 2>&1 || { tail -30 gpurun_out/r03_ax_tests.log; exit 1; }
tail -2 gpurun_out/r03_ax_tests.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 || exit 1
timeout -k 10 400 python bench.py > gpurun_out/r03_ax_bench.json 2> gpurun_out/r03_ax_bench.err || { tail -20 gpurun_out/r03_ax_bench.err; exit 1; }
cut -c1-400 gpurun_out/r03_ax_bench.json
