"""Counts, per decode step of the benchmark batch, the rows that belong to already finished hypotheses."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, tst_amd
from tst_amd import synthetic
from tst_amd.engine import DecodeLoop
from tst_amd import kernels as K

model, vocab = bench.build_model(torch.device('cuda', 0), 5.5)
src = synthetic.multi30k_like_sources(8192, bench.VOCAB, seed=1236).cuda()
encs, mask = model.encoder(src)
st = tst_amd.InferenceState({'model': model.decoder, 'inputs': encs, 'enc_mask': mask, 'vocab': vocab, 'max_len': 50, 'beam_size': 4})
sess = st._session()
tot = fin = 0
orig = sess.step
def step(rows, n, t):
    global tot, fin
    tot += n
    fin += int(rows.fin[:n].sum())
    return orig(rows, n, t)
sess.step = step
loop = DecodeLoop(sess, 8192, 4, 50, 2, 3, 1, 0.0, torch.device('cuda', 0))
loop.run()
print('row-steps %d, of finished hypotheses %d (%.1f%%)' % (tot, fin, 100.0 * fin / tot))
