#!/bin/bash
# config 4 (bf16) as 1 / 2 / 3 / 4 sub-batches that decode at the same time on streams of their own
for p in 1 2 3 4; do
  timeout -k 10 200 python tools/bench_configs.py cfg4 --dtype bf16 --parts $p 2>&1 | grep '"case"' | cut -c1-260 || exit 1
done
