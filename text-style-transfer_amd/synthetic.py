"""Deterministic synthetic weights and inputs for benchmarks and demos (there is no network for real
checkpoints or datasets).  Weights: every parameter, in checkpoint order, from one
``numpy.random.RandomState(seed)`` stream — matrices N(0, 1/fan_in) (``out.weight`` times ``out_scale``),
LayerNorm gains 1 + N(0, 0.01), other vectors N(0, 0.01) — plus ``eos_bias`` on the EOS output bias, since a
random-init model otherwise almost never ends a sentence.  The reference experiments fix
``random_seed: 42`` (``/root/reference/experiments/transformer-mt/config.yml:13``)."""
import math
import re

import numpy as np
import torch


def synthetic_state_dict(model, eos_idx=None, eos_bias=0.0, out_scale=1.0, seed=42):
    """Returns a state dict for ``model`` (anything with ``state_dict()``) filled as described above."""
    rs = np.random.RandomState(seed)
    sd = model.state_dict()
    out = {}
    for name, p in sd.items():
        if name == 'pe' or name.endswith('.pe'):
            out[name] = p
            continue
        draw = torch.from_numpy(rs.standard_normal(tuple(p.shape)).astype(np.float32))
        if p.dim() >= 2:
            std = 1.0 / math.sqrt(p.shape[-1])
            if name.endswith('out.weight'):
                std *= out_scale
            out[name] = draw * std
        elif re.search(r'(^|\.)ln\d*\.weight$', name):
            out[name] = 1.0 + 0.01 * draw
        else:
            out[name] = 0.01 * draw
    if eos_idx is not None:
        key = 'out.bias' if 'out.bias' in out else 'decoder.out.bias'
        out[key][eos_idx] += eos_bias
    return out


def multi30k_like_sources(n_sent, vocab_size, seed, pad=1, bos=2, eos=3, mean=13.0, std=4.0, lo=4, hi=40):
    """Synthetic source batch shaped like Multi30k sentences: ``clamp(round(N(mean, std)), lo, hi)`` content
    tokens (uniform over the non-special ids) between BOS and EOS, right-padded.  int64 [n_sent, S]."""
    rs = np.random.RandomState(seed)
    lens = np.clip(np.rint(rs.normal(mean, std, size=n_sent)), lo, hi).astype(np.int64)
    S = int(lens.max()) + 2
    src = np.full((n_sent, S), pad, dtype=np.int64)
    body = rs.randint(4, vocab_size, size=(n_sent, S))
    col = np.arange(S)[None, :]
    src[:, 0] = bos
    inside = (col >= 1) & (col <= lens[:, None])
    src[inside] = body[inside]
    src[np.arange(n_sent), lens + 1] = eos
    return torch.from_numpy(src)
