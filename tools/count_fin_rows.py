"""Counts, per decode step of the benchmark batch, the rows that belong to already finished hypotheses and
the live rows that can no longer win (a finished hypothesis of the same sentence has a strictly higher score)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, tst_amd
from tst_amd import synthetic
from tst_amd.engine import DecodeLoop

K = 4
model, vocab = bench.build_model(torch.device('cuda', 0), 5.5)
src = synthetic.multi30k_like_sources(8192, bench.VOCAB, seed=1236).cuda()
encs, mask = model.encoder(src)
st = tst_amd.InferenceState({'model': model.decoder, 'inputs': encs, 'enc_mask': mask, 'vocab': vocab, 'max_len': 50, 'beam_size': K})
sess = st._session()
tot = fin = dead = 0
per_step = []
orig = sess.step
def step(rows, n, t):
    global tot, fin, dead
    tot += n
    f = rows.fin[:n]
    nf = int(f.sum())
    nd = 0
    if t > 0:
        sc = rows.score[:n].view(-1, K)
        fm = f.view(-1, K) != 0
        best_fin = torch.where(fm, sc, torch.full_like(sc, float('-inf'))).max(1, keepdim=True)[0]
        nd = int(((~fm) & (sc < best_fin)).sum())
    fin += nf
    dead += nd
    per_step.append((t, n, nf, nd))
    return orig(rows, n, t)
sess.step = step
loop = DecodeLoop(sess, 8192, K, 50, 2, 3, 1, 0.0, torch.device('cuda', 0))
loop.run()
print('row-steps %d, of finished hypotheses %d (%.1f%%), of live hypotheses that cannot win %d (%.1f%%)'
      % (tot, fin, 100.0 * fin / tot, dead, 100.0 * dead / tot))
for r in per_step[::5]:
    print('step %2d rows %6d finished %5d dead %5d' % r)
