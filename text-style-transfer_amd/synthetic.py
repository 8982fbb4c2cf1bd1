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


def multi30k_like_lengths(n_sent, seed, mean=13.0, std=4.0, lo=4, hi=40):
    """Content-token counts of ``multi30k_like_sources(n_sent, ..., seed)`` without building the sentences (int64 [n_sent]);
    a sentence has two more tokens, BOS and EOS."""
    rs = np.random.RandomState(seed)
    return np.clip(np.rint(rs.normal(mean, std, size=n_sent)), lo, hi).astype(np.int64)


def multi30k_like_sources(n_sent, vocab_size, seed, pad=1, bos=2, eos=3, mean=13.0, std=4.0, lo=4, hi=40, rows=None,
                          chunk=65536):
    """Synthetic source batch shaped like Multi30k sentences: ``clamp(round(N(mean, std)), lo, hi)`` content
    tokens (uniform over the non-special ids) between BOS and EOS, right-padded.  int64 [n_sent, S].
    ``rows`` (ascending int64 indices) returns only those sentences of the same corpus, [len(rows), S] in that order: the
    random stream is drawn chunk by chunk and the other rows are dropped, so a rank of a sharded run never holds the corpus."""
    rs = np.random.RandomState(seed)
    lens = np.clip(np.rint(rs.normal(mean, std, size=n_sent)), lo, hi).astype(np.int64)
    S = int(lens.max()) + 2
    col = np.arange(S)[None, :]
    if rows is None:
        keep_all, rows = True, np.arange(n_sent, dtype=np.int64)
    else:
        keep_all, rows = False, np.asarray(rows, dtype=np.int64)
        if rows.size and (np.any(np.diff(rows) <= 0) or rows[0] < 0 or rows[-1] >= n_sent):
            raise ValueError('rows must be ascending indices into the corpus')
    src = np.full((rows.size, S), pad, dtype=np.int64)
    done = 0
    for c0 in range(0, n_sent, n_sent if keep_all else chunk):
        c1 = n_sent if keep_all else min(n_sent, c0 + chunk)
        body = rs.randint(4, vocab_size, size=(c1 - c0, S))     # the stream of the whole corpus, a chunk at a time
        hi_i = np.searchsorted(rows, c1)
        mine = rows[done:hi_i]
        if mine.size:
            ln = lens[mine]
            inside = (col >= 1) & (col <= ln[:, None])
            part = src[done:hi_i]
            part[:, 0] = bos
            part[inside] = body[mine - c0][inside]
            part[np.arange(mine.size), ln + 1] = eos
        done = hi_i
    return torch.from_numpy(src)
