"""Diagnostic: per decode step, how many rows are stepped and how many of them belong to finished beams
(rows whose model step is wasted).  Usage: python tools/diag_rows.py [batch] [eos_bias]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import tst_amd
from tst_amd import synthetic

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
eos_bias = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
model, vocab = bench.build_model(torch.device('cuda', 0), eos_bias)
src = synthetic.multi30k_like_sources(batch, bench.VOCAB, seed=1236).cuda()
encs, mask = model.encoder(src)
st = tst_amd.InferenceState({'model': model.decoder, 'inputs': encs, 'enc_mask': mask, 'vocab': vocab, 'max_len': 50,
                             'beam_size': 4})
sess = st._session()
orig = sess.step
log = []


def step(rows, n, t):
    fin = int(rows.fin[:n].sum().item())
    log.append((t, n, fin))
    return orig(rows, n, t)


sess.step = step
st._session = lambda: sess
st.inference()
tot = sum(n for _, n, _ in log)
dead = sum(f for _, _, f in log)
print('step rows finished_rows')
for t, n, f in log:
    print(t, n, f)
print('total row-steps %d, of which finished-beam rows %d (%.1f%%); step-0 surplus %d' % (tot, dead, 100.0 * dead / tot, log[0][1] * 3 // 4))
