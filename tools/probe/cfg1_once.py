"""One batch of config 1 (256 sentences, greedy) after a warm-up batch: for kernel-trace profiles of the small-batch regime."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import tst_amd
from tst_amd import synthetic

B = int(os.environ.get('SENTENCES', '256'))
dev = torch.device('cuda', 0)
model, vocab = bench.build_model(dev, 5.5)
src = synthetic.multi30k_like_sources(B, bench.VOCAB, seed=1235).to(dev)
for it in range(int(os.environ.get('BATCHES', '3'))):
    torch.cuda.synchronize()
    t0 = time.time()
    encs, mask = model.encoder(src)
    st = tst_amd.InferenceState({'model': model.decoder, 'inputs': encs, 'enc_mask': mask, 'vocab': vocab, 'max_len': bench.MAX_LEN})
    tok, lens, _ = st.decode_packed()
    lens.cpu()
    torch.cuda.synchronize()
    print('batch %d: %.2f ms, %d decode steps' % (it, (time.time() - t0) * 1e3, st.loop.steps_done), flush=True)
