"""cProfile of the 256-sentence transformer-mt batch (config 1) after warm-up: where the host time of a batch goes."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, tst_amd
from tst_amd import synthetic

dev = torch.device('cuda', 0)
model, vocab = bench.build_model(dev, 5.5)
src = synthetic.multi30k_like_sources(256, bench.VOCAB, seed=1235).to(dev)
model.decoder.event_log = None


def once():
    encs, enc_mask = model.encoder(src)
    st = tst_amd.InferenceState({'model': model.decoder, 'inputs': encs, 'enc_mask': enc_mask, 'vocab': vocab, 'max_len': 50})
    tok, lens, _ = st.decode_packed()
    return tok.cpu(), lens.cpu()


for _ in range(3):
    once()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    once()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(32)
