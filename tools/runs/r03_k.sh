#!/bin/bash
# r03_k: fixed and per-row cost of a config-2 decode step; estimate of what a full-batch corpus walk would save
set -o pipefail
timeout -k 10 200 python tools/probe/step_cost_fit.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_k_step_cost_fit.txt
