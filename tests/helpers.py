"""Shared helpers of the test-suite: fixture loading and model reconstruction from fixture configs."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def load_fixture(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    fx = {k: z[k] for k in z.files}
    if 'config' in fx:
        fx['config'] = json.loads(bytes(fx['config']).decode())
    if 'tokens' in fx:
        fx['seqs'] = [fx['tokens'][i, :n].tolist() for i, n in enumerate(fx['lens'])]
    return fx


class replay_rand:
    """Context manager: ``torch.rand`` returns the recorded uniform draws of a fixture, one array per call (the
    reference draws its Gumbel noise with ``torch.rand``: /root/reference/src/utils/gumbel.py:12)."""

    def __init__(self, draws):
        self.draws, self.i = list(draws), 0

    def __enter__(self):
        import torch
        self.real = torch.rand

        def fake(*size, **kw):
            shape = tuple(size[0]) if len(size) == 1 and not isinstance(size[0], int) else tuple(size)
            u = torch.from_numpy(np.array(self.draws[self.i]))
            assert tuple(u.shape) == shape, (self.i, tuple(u.shape), shape)
            self.i += 1
            return u.to(kw['device']) if 'device' in kw else u
        torch.rand = fake
        return self

    def __exit__(self, *exc):
        import torch
        torch.rand = self.real


def torch_model_for(cfg):
    """Rebuilds the torch specification model a fixture was made with (weights are a pure function of
    the config: ``oracle.torch_ref.synthetic_init``)."""
    from oracle import torch_ref as R
    kind = cfg['kind']
    if kind in ('transformer', 'gumbel_tfm'):
        vocab = R.synthetic_vocab(cfg['V'])
        model = R.synthetic_init(R.Transformer(R.HP(cfg['hp']), vocab, vocab), vocab.stoi['<eos>'], cfg['eos_bias'],
                                 cfg['out_scale'], seed=cfg['weight_seed'])
        return model, vocab
    if kind == 'tlm':
        vocab = R.synthetic_vocab(cfg['V'])
        model = R.synthetic_init(R.TransformerLM(R.HP(cfg['hp']), vocab), vocab.stoi['<eos>'], cfg['eos_bias'],
                                 cfg['out_scale'], seed=cfg['weight_seed'])
        return model, vocab
    vocab = R.synthetic_vocab(cfg['V'], extra=('|',))
    if kind in ('gru', 'gumbel_gru'):
        model = R.RNNDecoder(cfg['E'], cfg['H'], vocab)
        eos = vocab.stoi[cfg.get('eos_token', '<eos>')]
    elif kind == 'rnnlm_stateful':
        model = R.RNNLM(cfg['H'], vocab, cfg['n_layers'])
        eos = vocab.stoi['|']
    elif kind == 'conditional_lm':
        model = R.ConditionalLM(cfg['H'], vocab, cfg['n_layers'])
        eos = vocab.stoi['|']
    else:
        raise ValueError(kind)
    return R.synthetic_init(model, eos, cfg['eos_bias'], cfg['out_scale'], seed=cfg['weight_seed']), vocab


def logits_close(a, b, rtol=1e-4, floor=0.1):
    """The fp32 tolerance of this build: |a - b| <= rtol * max(|b|, floor), elementwise: 1e-4 RELATIVE for every logit of
    magnitude >= 0.1, and the bound frozen at its value for |b| = 0.1 (1e-5 absolute) below that.  The floor is needed, not
    a loosening: the rounding error of a K-term fp32 dot product is absolute (about K * 2**-24 * sum|a_k w_k|, a few 1e-6
    here) whatever the result's magnitude, random-init logits cross zero (a tenth of a row lies within +-0.1), and a purely
    relative bound is undefined there."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    err = np.abs(a - b) - rtol * np.maximum(np.abs(b), floor)
    return bool((err <= 0).all()), float(np.abs(a - b).max())
