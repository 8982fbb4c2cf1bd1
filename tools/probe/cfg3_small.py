"""GRU decoder (cyclegan sizes, character vocabulary) at the reference's batch size: wall time per batch, for kernel-trace runs."""
import os
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
for it in range(3):
    torch.cuda.synchronize()
    t0 = time.time()
    out = tst_amd.simple_inference(model, z, vocab, max_len=100, eos_token='|')
    torch.cuda.synchronize()
    print('batch %d: %.2f ms, %d tokens' % (it, (time.time() - t0) * 1e3, sum(len(s) - 1 for s in out)), flush=True)
