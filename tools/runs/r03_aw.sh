#!/bin/bash
# counters (matrix-pipe busy share, clock, fabric bytes) of the kernels added last: small greedy batches (ln_vocab_argmax_kernel)
# and config 3 with the per-token table (gru_fused_kernel<true, true>)
bash tools/prof_util.sh r03_cfg1 tools/bench_configs.py cfg1 --sentences 256 && bash tools/prof_util.sh r03_cfg3 tools/bench_configs.py cfg3
