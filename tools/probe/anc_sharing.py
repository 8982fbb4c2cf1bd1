"""Probe: how often do all beams of a sentence read the SAME cache row for a past position?  (If often, a wave that serves
the beams of a sentence together could load those keys / values once.)  Config 2 model, 1024 sentences, beam 4; and the
config 4 language model, 512 sentences, beam 8."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import tst_amd
from tst_amd import synthetic


def watch(session, k, stats):
    inner = session.step

    def step(rows, n, t, soft=None):
        if t > 0 and n % k == 0 and session.rows_per_sentence == k:
            anc = rows.anc[:n, :t].view(n // k, k, t)
            same = (anc == anc[:, :1, :]).all(dim=1)          # [sentences, positions]
            stats[0] += int(same.sum())
            stats[1] += same.numel()
            pairs = (anc[:, :, None, :] == anc[:, None, :, :]).float().mean()
            stats[2] += float(pairs) * same.numel()
        return inner(rows, n, t, soft) if soft is not None else inner(rows, n, t)
    session.step = step


dev = torch.device('cuda', 0)
model, vocab = bench.build_model(dev, 5.5)
src = synthetic.multi30k_like_sources(1024, bench.VOCAB, seed=1236).to(dev)
encs, mask = model.encoder(src)
st = tst_amd.InferenceState({'model': model.decoder, 'inputs': encs, 'enc_mask': mask, 'vocab': vocab, 'max_len': 50, 'beam_size': 4})
sess = st._session()
stats = [0, 0, 0.0]
watch(sess, 4, stats)
st._session = lambda n_positions=None: sess
st.decode_packed()
print('config 2 (beam 4): all beams share the cache row at %.1f %% of (sentence, past position) pairs; a pair of beams shares at %.1f %%'
      % (100.0 * stats[0] / stats[1], 100.0 * stats[2] / stats[1]))

hp = dict(model_size=512, n_layers=6, n_heads=8, d_ff=2048, pe_max_len=64)
lm = tst_amd.TransformerLM(tst_amd.HP(hp), vocab, device=dev, dtype='bf16')
lm.load_state_dict(synthetic.synthetic_state_dict(lm, 3, 5.0, 2.0, seed=42))
z = (0.3 * torch.randn((512, 512), generator=torch.Generator().manual_seed(1238))).to(dev)
st = tst_amd.InferenceState({'model': lm, 'inputs': z, 'vocab': vocab, 'max_len': 50, 'beam_size': 8})
sess = st._session()
sess.rows_per_sentence = 1
stats = [0, 0, 0.0]
watch(sess, 8, stats)
st._session = lambda n_positions=None: sess
st.decode_packed()
print('config 4 (beam 8): all beams share at %.1f %%; a pair of beams shares at %.1f %%' % (100.0 * stats[0] / stats[1], 100.0 * stats[2] / stats[1]))
