"""Calls of InferenceState with the input variants a caller of the reference might pass (CPU tensors, other float types,
non-contiguous views, mask layouts, a foreign vocabulary object, a side stream, numpy integers): each must give the outputs of
the plain call or a clean Python exception."""
import os
import sys
import traceback

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tst_amd as T
from tst_amd import synthetic

V = 300
vocab = T.Vocab.synthetic(V)
hp = dict(d_model=256, dropout=0.1, n_layers=2, pe_max_len=60, n_heads=2, d_ff=1024)
model = T.Transformer(T.HP(hp), vocab, vocab)
model.load_state_dict(synthetic.synthetic_state_dict(model, 3, 3.0, 2.0, seed=7))
src = synthetic.multi30k_like_sources(50, V, seed=9).cuda()
encs, mask = model.encoder(src)
encs, mask = encs.clone(), mask.clone()


def dec(**kw):
    d = {'model': model.decoder, 'inputs': encs, 'enc_mask': mask, 'vocab': vocab, 'max_len': 12, 'beam_size': 2}
    d.update(kw)
    return [s.cpu().tolist() for s in T.InferenceState(d).inference()]


base = dec()
gru = T.RNNDecoder(64, 128, vocab)
gru.load_state_dict(synthetic.synthetic_state_dict(gru, 3, 1.0, 2.0, seed=3))
z = torch.randn((40, 128), generator=torch.Generator().manual_seed(2)).cuda()


def gdec(**kw):
    d = {'model': gru, 'inputs': z, 'vocab': vocab, 'max_len': 10}
    d.update(kw)
    return [s.cpu().tolist() for s in T.InferenceState(d).inference()]


gbase = gdec()


class ForeignVocab:
    def __init__(self, v):
        self.stoi, self.itos = dict(v.stoi), list(v.itos)

    def __len__(self):
        return len(self.itos)


big = torch.zeros((50, encs.shape[1], 512), device='cuda')
big[:, :, ::2] = encs
side = torch.cuda.Stream()


def on_side():
    with torch.cuda.stream(side):
        out = dec()
    side.synchronize()
    return out


cases = {
    'cpu inputs': lambda: dec(inputs=encs.cpu(), enc_mask=mask.cpu()) == base,
    'float64 inputs': lambda: dec(inputs=encs.double()) == base,
    'float16 gru state': lambda: len(gdec(inputs=z.half())) == 40,
    'non-contiguous inputs': lambda: dec(inputs=big[:, :, ::2]) == base,
    'mask [B,S] bool': lambda: dec(enc_mask=mask.reshape(50, -1).bool()) == base,
    'mask uint8': lambda: dec(enc_mask=mask.to(torch.uint8)) == base,
    'mask int64': lambda: dec(enc_mask=mask.long()) == base,
    'requires_grad inputs': lambda: dec(inputs=encs.clone().requires_grad_(True)) == base,
    'numpy integer options': lambda: dec(max_len=np.int64(12), beam_size=np.int32(2)) == base,
    'foreign vocabulary object': lambda: dec(vocab=ForeignVocab(vocab)) == base,
    'side stream': lambda: on_side() == base,
    'gru cpu inputs': lambda: gdec(inputs=z.cpu()) == gbase,
    'gru float64 inputs': lambda: gdec(inputs=z.double()) == gbase,
    'gru non-contiguous inputs': lambda: gdec(inputs=torch.cat([z, z], 1)[:, :128]) == gbase,
    'gru 3-d state [1,B,H]': lambda: gdec(inputs=z.unsqueeze(0), inputs_batch_dim=1) == gbase,
    'twice on one state': lambda: (lambda st: [s.cpu().tolist() for s in st.inference()] == [s.cpu().tolist() for s in st.inference()])(
        T.InferenceState({'model': gru, 'inputs': z, 'vocab': vocab, 'max_len': 10})),
    'max_len beyond pe table (clean error)': lambda: _raises(lambda: dec(max_len=100), ValueError),
    'unknown eos token (clean error)': lambda: _raises(lambda: dec(eos_token='nope'), AssertionError),
}


def _raises(fn, exc):
    try:
        fn()
    except exc:
        return True
    return False


bad = 0
for name, fn in cases.items():
    try:
        ok = fn()
        print('%-40s %s' % (name, 'ok' if ok else 'DIFFERENT OUTPUT'), flush=True)
        bad += not ok
    except Exception as e:   # noqa: BLE001
        bad += 1
        print('%-40s RAISED %s: %s' % (name, type(e).__name__, str(e)[:200]), flush=True)
        traceback.print_exc(limit=3)
print('RESULT', 'all ok' if bad == 0 else '%d problems' % bad)
