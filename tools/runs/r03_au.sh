#!/bin/bash
# the GRU cell alone at E = H = 1024: one-kernel and multi-call forms, with the input product and with the per-token table
mkdir -p gpurun_out
timeout -k 10 300 python tools/bench_gru_cell.py 2>&1 | grep rows | tee gpurun_out/r03_gru_cell_table.jsonl
