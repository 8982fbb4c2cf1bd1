#!/bin/bash
# config 2 with the vocabulary projection on the weights-in-registers kernel (TST_VOCAB_VARIANT=5) against the LDS-DMA kernel (3), alternating
for v in 3 5 3 5 3 5; do
  TST_VOCAB_VARIANT=$v timeout -k 10 200 python bench.py --no-cpu-baseline --steps 8 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('TST_VOCAB_VARIANT=$v', round(d['ms_per_step'],2), round(d['value']), d['output_checksum'], 'projection %.1f TF %.3f ms per launch' % (d['roofline']['achieved'], d['roofline']['avg_launch_ms']))" || exit 1
done
