"""cProfile of the 128-sentence GRU batch (cyclegan sizes) after warm-up: where the host time of a batch goes."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tst_amd
from tst_amd import synthetic

B = int(os.environ.get('SENTENCES', '128'))
dev = torch.device('cuda', 0)
vocab = tst_amd.Vocab.synthetic(128, extra=('|',))
model = tst_amd.RNNDecoder(1024, 1024, vocab, device=dev)
model.load_state_dict(synthetic.synthetic_state_dict(model, vocab.stoi['|'], 0.5, 2.0, seed=42))
z = torch.randn((B, 1024), generator=torch.Generator().manual_seed(1237)).to(dev)


def once():
    st = tst_amd.InferenceState({'model': model, 'inputs': z, 'vocab': vocab, 'max_len': 100, 'eos_token': '|'})
    tok, lens, _ = st.decode_packed()
    return lens.cpu(), st.loop.steps_done


for _ in range(5):
    once()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(20):
    _, steps = once()
torch.cuda.synchronize()
print('%.3f ms per batch, %d decode steps' % ((time.time() - t0) / 20 * 1e3, steps))
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    once()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
