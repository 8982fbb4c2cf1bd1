"""cProfile of one small decode call after warm-up (where does the host time go?)."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tst_amd
from tst_amd import synthetic

dev = torch.device('cuda', 0)
vocab = tst_amd.Vocab.synthetic(128, extra=('|',))
model = tst_amd.RNNDecoder(1024, 1024, vocab, device=dev)
model.load_state_dict(synthetic.synthetic_state_dict(model, vocab.stoi['|'], 0.5, 2.0, seed=42))
z = torch.randn((128, 1024), generator=torch.Generator().manual_seed(1237)).to(dev)
for _ in range(3):
    tst_amd.simple_inference(model, z, vocab, max_len=100, eos_token='|')
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    tst_amd.simple_inference(model, z, vocab, max_len=100, eos_token='|')
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
