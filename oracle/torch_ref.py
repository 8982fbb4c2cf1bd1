"""TEST INFRASTRUCTURE — CPU reference of the decode hot path in stock PyTorch (fp32).

Nothing in the product package may import this file; it is used by tests/, by the fixture
generator in tests/golden/, by ``__graft_entry__.smoke()`` and by the ``cpu_baseline`` leg of
``bench.py``.

Two things live here:

* Model specifications.  The reference's model sources (``src/models``) are an un-vendored git
  submodule (``/root/reference/.gitmodules:1-3``) and absent, so the arithmetic of the models is
  specified here from the call-site contracts:
    - ``Transformer(hp, vocab_src, vocab_trg)``, ``.encoder(src) -> (encs, enc_mask)``,
      ``.decoder(encs, ys, enc_mask) -> [B, t, V]`` raw logits
      (``/root/reference/src/trainers/transformer_mt_trainer.py:55,74-76,91-98``; sizes from
      ``/root/reference/experiments/transformer-mt/config.yml:21-27``)
    - ``RNNDecoder(emb, hid, vocab, dropword_p)``, ``decoder(z[B,H], x[B,T]) -> [B,T,V]``,
      attributes ``.embed`` and ``.gru`` (``/root/reference/src/trainers/cyclegan_trainer.py:64``,
      ``/root/reference/src/prod/style_model.py:39,64-69``)
    - ``RNNLM``: ``lm(z, x, return_z=True) -> (logits, z')``
      (``/root/reference/src/prod/dialog_model.py:97-120``); ``ConditionalLM`` with ``style=`` (``:52,99-100``)
    - ``RNNEncoder``, ``Generator`` (``/root/reference/src/trainers/cyclegan_trainer.py:63,337-348``), ``FFN``
      (``/root/reference/src/prod/style_model.py:40,67``), ``TransformerLM`` (``/root/reference/src/trainers/lm_trainer.py:54,72-74``)
  Parity of this arithmetic against the absent submodule is unpinned.

* ``RefInferenceState`` / ``ref_simple_inference``: a restatement of the decode loop of
  ``/root/reference/src/inference.py:11-181`` (full-prefix recompute at every step, batch compaction,
  BOS-led / EOS-stripped outputs).  It is pinned against the real reference loop by the golden
  fixtures under ``tests/golden/`` (made by ``tests/golden/make_golden.py``, which imports the real
  ``src/inference.py``).
"""
import math
import re

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------
# Vocabulary (legacy-torchtext behaviour: specials first, unknown strings map to the unk id)
# --------------------------------------------------------------------------------------------
class _Stoi(dict):
    def __init__(self, mapping, unk_id):
        super().__init__(mapping)
        self._unk = unk_id

    def __missing__(self, key):
        return self._unk


class RefVocab:
    def __init__(self, tokens, specials=('<unk>', '<pad>', '<bos>', '<eos>')):
        self.itos = list(specials) + [t for t in tokens if t not in specials]
        self.stoi = _Stoi({t: i for i, t in enumerate(self.itos)}, 0)

    def __len__(self):
        return len(self.itos)


def synthetic_vocab(size, extra=()):
    """<unk>,<pad>,<bos>,<eos> at ids 0..3, then ``extra`` tokens, then 'w<i>' fillers up to ``size``."""
    toks = list(extra)
    toks += ['w%d' % i for i in range(size - 4 - len(toks))]
    v = RefVocab(toks)
    assert len(v) == size
    return v


def embed_tokens(embedding, ys):
    """Embedding of token ids ``[B, t]`` or, for the soft decoding mode, the expected embedding of token
    distributions ``[B, t, V]`` (a floating tensor): ``dist @ weight``."""
    if ys.is_floating_point():
        return torch.matmul(ys, embedding.weight)
    return embedding(ys)


def ref_gumbel_softmax(dists, u, temperature=1):
    """``gumbel_softmax_sample`` of ``/root/reference/src/utils/gumbel.py:11-21`` with the uniform draws ``u`` given
    (the reference takes them from ``torch.rand``): softmax((dists + G) / temperature), G = -log(-log(u + eps) + eps)."""
    noise = -torch.log(-torch.log(u + 1e-20) + 1e-20)
    return F.softmax((dists + noise) / temperature, dim=-1)


class SoftInput(nn.Module):
    """Adapter under which the gumbel branch of the reference loop runs
    (``/root/reference/src/inference.py:104-105,136-138``): with ``gumbel=True`` the loop concatenates the
    ``[B, 1, V]`` distributions along the LAST axis, so after t steps ``active_seqs`` is ``[B, 1, t * V]``; this
    adapter views it as t distributions over V tokens and calls the wrapped model, whose embedding accepts
    distributions (``embed_tokens``)."""

    def __init__(self, model, vocab_size):
        super().__init__()
        self.model, self.V = model, vocab_size

    def forward(self, inputs, active, *args, **kwargs):
        return self.model(inputs, active.reshape(active.shape[0], -1, self.V), *args, **kwargs)


# --------------------------------------------------------------------------------------------
# Transformer (pre-norm encoder-decoder, sinusoidal positions, raw-logit output)
# --------------------------------------------------------------------------------------------
def sinusoid_table(n_pos, d_model):
    pos = torch.arange(n_pos, dtype=torch.float64).unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float64) * (-math.log(10000.0) / d_model))
    pe = torch.zeros(n_pos, d_model, dtype=torch.float64)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe.to(torch.float32)


class HP(dict):
    """Attribute-access dict, the shape firelab configs have (``config.hp.transformer.d_model``)."""
    __getattr__ = dict.__getitem__


def mt_hp(**over):
    hp = HP(d_model=256, dropout=0.1, n_layers=2, pe_max_len=60, n_heads=2, d_ff=1024)
    hp.update(over)
    return hp


class Attention(nn.Module):
    def __init__(self, d_model, n_heads):
        super().__init__()
        assert d_model % n_heads == 0
        self.h, self.dh = n_heads, d_model // n_heads
        self.q = nn.Linear(d_model, d_model)
        self.k = nn.Linear(d_model, d_model)
        self.v = nn.Linear(d_model, d_model)
        self.o = nn.Linear(d_model, d_model)

    def forward(self, x_q, x_kv, mask):
        """x_q [B,Tq,d], x_kv [B,Tk,d], mask broadcastable to [B,1,Tq,Tk] (True = may attend)."""
        B, Tq, _ = x_q.shape
        Tk = x_kv.shape[1]
        q = self.q(x_q).view(B, Tq, self.h, self.dh).transpose(1, 2)
        k = self.k(x_kv).view(B, Tk, self.h, self.dh).transpose(1, 2)
        v = self.v(x_kv).view(B, Tk, self.h, self.dh).transpose(1, 2)
        s = torch.matmul(q, k.transpose(-1, -2)) * (1.0 / math.sqrt(self.dh))
        if mask is not None:
            s = s.masked_fill(~mask, -1e9)
        p = F.softmax(s, dim=-1)
        y = torch.matmul(p, v).transpose(1, 2).reshape(B, Tq, self.h * self.dh)
        return self.o(y)


class FeedForward(nn.Module):
    def __init__(self, d_model, d_ff):
        super().__init__()
        self.w1 = nn.Linear(d_model, d_ff)
        self.w2 = nn.Linear(d_ff, d_model)

    def forward(self, x):
        return self.w2(F.relu(self.w1(x)))


class EncoderLayer(nn.Module):
    def __init__(self, d_model, n_heads, d_ff):
        super().__init__()
        self.ln1 = nn.LayerNorm(d_model)
        self.attn = Attention(d_model, n_heads)
        self.ln2 = nn.LayerNorm(d_model)
        self.ff = FeedForward(d_model, d_ff)

    def forward(self, x, mask):
        h = self.ln1(x)
        x = x + self.attn(h, h, mask)
        return x + self.ff(self.ln2(x))


class DecoderLayer(nn.Module):
    def __init__(self, d_model, n_heads, d_ff):
        super().__init__()
        self.ln1 = nn.LayerNorm(d_model)
        self.self_attn = Attention(d_model, n_heads)
        self.ln2 = nn.LayerNorm(d_model)
        self.cross_attn = Attention(d_model, n_heads)
        self.ln3 = nn.LayerNorm(d_model)
        self.ff = FeedForward(d_model, d_ff)

    def forward(self, x, mem, causal, mem_mask):
        h = self.ln1(x)
        x = x + self.self_attn(h, h, causal)
        x = x + self.cross_attn(self.ln2(x), mem, mem_mask)
        return x + self.ff(self.ln3(x))


class TransformerEncoder(nn.Module):
    def __init__(self, hp, vocab):
        super().__init__()
        d = hp.d_model
        self.d_model = d
        self.pad_idx = vocab.stoi['<pad>']
        self.embed = nn.Embedding(len(vocab), d)
        self.register_buffer('pe', sinusoid_table(hp.pe_max_len, d))
        self.layers = nn.ModuleList([EncoderLayer(d, hp.n_heads, hp.d_ff) for _ in range(hp.n_layers)])
        self.ln = nn.LayerNorm(d)

    def forward(self, src):
        """src [B,S] ids -> (encs [B,S,d], enc_mask [B,1,S] bool, True where src is not padding)."""
        enc_mask = (src != self.pad_idx).unsqueeze(1)
        x = self.embed(src) * math.sqrt(self.d_model) + self.pe[:src.shape[1]]
        for layer in self.layers:
            x = layer(x, enc_mask.unsqueeze(1))
        return self.ln(x), enc_mask


class TransformerDecoder(nn.Module):
    def __init__(self, hp, vocab):
        super().__init__()
        d = hp.d_model
        self.d_model = d
        self.embed = nn.Embedding(len(vocab), d)
        self.register_buffer('pe', sinusoid_table(hp.pe_max_len, d))
        self.layers = nn.ModuleList([DecoderLayer(d, hp.n_heads, hp.d_ff) for _ in range(hp.n_layers)])
        self.ln = nn.LayerNorm(d)
        self.out = nn.Linear(d, len(vocab))

    def forward(self, encs, ys, *enc_mask):
        """encs [B,S,d], ys [B,t] ids, enc_mask [B,1,S] (or its B rows, the way
        ``/root/reference/src/inference.py:124,132`` star-unpacks it) -> raw logits [B,t,V]."""
        if len(enc_mask) == 0:
            mem_mask = None
        elif len(enc_mask) == 1 and enc_mask[0].dim() == 3:
            mem_mask = enc_mask[0].unsqueeze(1)
        else:
            mem_mask = torch.stack(list(enc_mask)).unsqueeze(1)
        t = ys.shape[1]
        causal = torch.ones(t, t, dtype=torch.bool, device=ys.device).tril().view(1, 1, t, t)
        x = embed_tokens(self.embed, ys) * math.sqrt(self.d_model) + self.pe[:t]
        for layer in self.layers:
            x = layer(x, encs, causal, mem_mask)
        return self.out(self.ln(x))


class Transformer(nn.Module):
    def __init__(self, hp, vocab_src, vocab_trg):
        super().__init__()
        self.encoder = TransformerEncoder(hp, vocab_src)
        self.decoder = TransformerDecoder(hp, vocab_trg)

    def forward(self, src, trg):
        encs, enc_mask = self.encoder(src)
        return self.decoder(encs, trg[:, :-1], enc_mask)


class LMLayer(nn.Module):
    def __init__(self, d_model, n_heads, d_ff):
        super().__init__()
        self.ln1 = nn.LayerNorm(d_model)
        self.self_attn = Attention(d_model, n_heads)
        self.ln2 = nn.LayerNorm(d_model)
        self.ff = FeedForward(d_model, d_ff)

    def forward(self, x, causal):
        h = self.ln1(x)
        x = x + self.self_attn(h, h, causal)
        return x + self.ff(self.ln2(x))


class TransformerLM(nn.Module):
    """``TransformerLM(hp.transformer, vocab)``, ``lm(z [B, model_size], x [B, T]) -> [B, T, V]`` and
    ``lm.inference(src, vocab, eos_token, max_len)`` (``/root/reference/src/trainers/lm_trainer.py:54,72-74,86``;
    ``model_size: 512`` at ``/root/reference/experiments/lm/config.yml:25-26``).  Decoder-only pre-norm
    layers; the conditioning vector z is added to the input embedding of every position."""

    def __init__(self, hp, vocab):
        super().__init__()
        d = hp.get('model_size', hp.get('d_model'))
        self.d_model = d
        self.n_heads = hp.get('n_heads', 8)
        self.embed = nn.Embedding(len(vocab), d)
        self.layers = nn.ModuleList([LMLayer(d, self.n_heads, hp.get('d_ff', 4 * d)) for _ in range(hp.get('n_layers', 6))])
        self.ln = nn.LayerNorm(d)
        self.out = nn.Linear(d, len(vocab))
        self.register_buffer('pe', sinusoid_table(hp.get('pe_max_len', 512), d))

    def forward(self, z, x):
        t = x.shape[1]
        causal = torch.ones(t, t, dtype=torch.bool, device=x.device).tril().view(1, 1, t, t)
        h = embed_tokens(self.embed, x) * math.sqrt(self.d_model) + self.pe[:t] + z.unsqueeze(1)
        for layer in self.layers:
            h = layer(h, causal)
        return self.out(self.ln(h))

    def inference(self, src, vocab, eos_token='<eos>', max_len=100):
        """Greedy continuation of the 1-D prefix ``src`` with z = 0; returns the generated ids (no prefix, no EOS)."""
        eos = vocab.stoi[eos_token]
        seq = src.reshape(1, -1).long()
        z = torch.zeros(1, self.d_model)
        out = []
        with torch.no_grad():
            for _ in range(max_len):
                tok = int(self.forward(z, seq)[0, -1].max(dim=-1)[1])
                if tok == eos:
                    break
                out.append(tok)
                seq = torch.cat([seq, torch.tensor([[tok]])], dim=1)
        return torch.tensor(out, dtype=torch.long)


# --------------------------------------------------------------------------------------------
# GRU models
# --------------------------------------------------------------------------------------------
class RNNDecoder(nn.Module):
    def __init__(self, emb_size, hid_size, vocab, dropword_p=0.0):
        super().__init__()
        self.embed = nn.Embedding(len(vocab), emb_size)
        self.gru = nn.GRU(emb_size, hid_size, batch_first=True)
        self.out = nn.Linear(hid_size, len(vocab))

    def forward(self, z, x):
        hid, _ = self.gru(embed_tokens(self.embed, x), z.unsqueeze(0))
        return self.out(hid)


class RNNEncoder(nn.Module):
    def __init__(self, emb_size, hid_size, vocab, dropword_p=0.0, noise=0.0):
        super().__init__()
        self.embed = nn.Embedding(len(vocab), emb_size)
        self.gru = nn.GRU(emb_size, hid_size, batch_first=True)

    def forward(self, x):
        return self.gru(self.embed(x))[1].squeeze(0)


class Generator(nn.Module):
    """Restatement of ``Generator`` at ``/root/reference/src/trainers/cyclegan_trainer.py:337-348``: a GRU that is fed
    ``n_steps`` copies of z, starts at z and returns its final state."""

    def __init__(self, size, n_steps):
        super().__init__()
        self.n_steps = n_steps
        self.gru = nn.GRU(size, size, batch_first=True)

    def forward(self, z):
        x = z.unsqueeze(1).repeat(1, self.n_steps, 1)
        return self.gru(x, z.unsqueeze(0))[1].squeeze(0)


class FFN(nn.Module):
    """Specification of ``FFN(sizes)`` as the pre-decode path uses it (``merge_z`` at
    ``/root/reference/src/prod/style_model.py:40,67``): linear layers, ReLU between them, none after the last.  The
    reference's class is in the absent model-zoo submodule; the one-layer form is a linear map under any choice."""

    def __init__(self, sizes, dropout=0.0):
        super().__init__()
        self.layers = nn.ModuleList([nn.Linear(a, b) for a, b in zip(sizes[:-1], sizes[1:])])

    def forward(self, x):
        for i, layer in enumerate(self.layers):
            x = layer(x)
            if i + 1 < len(self.layers):
                x = torch.relu(x)
        return x


class RNNLM(nn.Module):
    """Multi-layer GRU language model with carried state ``z [n_layers, B, size]``; constructor as at
    ``/root/reference/src/prod/dialog_model.py:49``: ``RNNLM(size, vocab, n_layers=...)``."""

    def __init__(self, size, vocab, n_layers=1):
        super().__init__()
        self.embed = nn.Embedding(len(vocab), size)
        self.gru = nn.GRU(size, size, num_layers=n_layers, batch_first=True)
        self.out = nn.Linear(size, len(vocab))

    def forward(self, z, x, return_z=False):
        hid, z_new = self.gru(self.embed(x), z)
        logits = self.out(hid)
        return (logits, z_new) if return_z else logits


class ConditionalLM(nn.Module):
    """Specification of ``ConditionalLM(size, vocab)`` as called at ``/root/reference/src/prod/dialog_model.py:52,99-100``:
    ``lm(z [2, B, size], x, style=s, return_z=True) -> (logits, z')``.  The reference's class is in the absent model-zoo
    submodule; this build's choice: a GRU language model (two layers, as the caller's z has) whose token embeddings
    get a learned style embedding added; ``style`` is one index for the whole batch or one per sentence."""

    def __init__(self, size, vocab, n_layers=2, n_styles=2):
        super().__init__()
        self.embed = nn.Embedding(len(vocab), size)
        self.style_embed = nn.Embedding(n_styles, size)
        self.gru = nn.GRU(size, size, num_layers=n_layers, batch_first=True)
        self.out = nn.Linear(size, len(vocab))

    def forward(self, z, x, style=0, return_z=False):
        styles = torch.as_tensor(style, dtype=torch.long, device=x.device).reshape(-1).expand(x.shape[0])
        hid, z_new = self.gru(self.embed(x) + self.style_embed(styles).unsqueeze(1), z)
        logits = self.out(hid)
        return (logits, z_new) if return_z else logits


class WeightedLMEnsemble(nn.Module):
    """Restatement of ``WeightedLMEnsemble`` at ``/root/reference/src/prod/throwaway_models.py:9-36``: every member gets
    its own state ``zs[i]``; the prediction is log(sum_i w_i * softmax(logits_i))."""

    def __init__(self, models, weights):
        super().__init__()
        assert len(models) == len(weights)
        assert sum(weights) == 1
        self.models = nn.ModuleList(models)
        self.weights = weights

    def forward(self, zs, x, return_z=False):
        if return_z:
            results = [m(z, x, return_z) for m, z in zip(self.models, zs)]
            return self.weight([r[0] for r in results]), torch.stack([r[1] for r in results])
        return self.weight([m(z, x) for m, z in zip(self.models, zs)])

    def weight(self, logits):
        probs = [F.softmax(l, dim=2) for l in logits]
        return sum([w * p for w, p in zip(self.weights, probs)]).log()


def synthetic_init(module, eos_idx=None, eos_bias=0.0, out_scale=1.0, seed=42):
    """Deterministic random init used by fixtures, tests and the benchmark.

    Every parameter, in ``named_parameters()`` order, is drawn from one
    ``numpy.random.RandomState(seed)`` (a frozen stream, so the weights are the same on every
    machine; the reference experiments use ``random_seed: 42``,
    ``/root/reference/experiments/transformer-mt/config.yml:13``): matrices N(0, 1/fan_in) (a
    parameter named ``out.weight`` additionally times ``out_scale``), LayerNorm gains
    1 + N(0, 0.01), every other vector N(0, 0.01).  ``eos_bias`` is then added to the output bias at
    ``eos_idx`` (random-init models otherwise almost never emit EOS).
    """
    import numpy as np
    rs = np.random.RandomState(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            draw = torch.from_numpy(rs.standard_normal(tuple(p.shape)).astype(np.float32))
            if p.dim() >= 2:
                std = 1.0 / math.sqrt(p.shape[-1])
                if name.endswith('out.weight'):
                    std *= out_scale
                p.copy_(draw * std)
            elif re.search(r'(^|\.)ln\d*\.weight$', name):
                p.copy_(1.0 + 0.01 * draw)
            else:
                p.copy_(0.01 * draw)
        if eos_idx is not None:
            out = module.out if hasattr(module, 'out') else module.decoder.out
            out.bias[eos_idx] += eos_bias
    return module.eval()


# --------------------------------------------------------------------------------------------
# Decode loop, restating /root/reference/src/inference.py
# --------------------------------------------------------------------------------------------
class RefInferenceState:
    """Restatement of ``InferenceState`` (``/root/reference/src/inference.py:11-167``), greedy and
    sampling selection, stateful (``substitute_inputs``) and full-prefix modes, and the gumbel branch
    (``:104-105,136-138`` with ``/root/reference/src/utils/gumbel.py:11-21``; it needs a caller-provided
    ``active_seqs`` of distributions ``[B, 1, V]`` and a model that takes them, see ``SoftInput``).  Keeps the
    reference's data flow: the whole prefix is fed to the model at every step
    (``:132``), finished rows leave the batch (``:102-120``), a finished sequence is stored as it
    was *before* its EOS was appended (``:94-100``), unfinished rows are stored after ``max_len``
    steps (``:65-69``)."""

    def __init__(self, state):
        self.model = state['model']
        self.vocab = state['vocab']
        self.inputs = state['inputs']
        self.bos_idx = self.vocab.stoi[state.get('bos_token', '<bos>')]
        self.eos_idx = self.vocab.stoi[state.get('eos_token', '<eos>')]
        self.max_len = state.get('max_len', 100)
        self.temperature = state.get('temperature', 1)
        self.sample_type = state.get('sample_type', 'max')
        self.kwargs = state.get('kwargs', {})
        self.enc_mask = state.get('enc_mask')
        self.inputs_batch_dim = state.get('inputs_batch_dim', 0)
        self.substitute_inputs = state.get('substitute_inputs', False)
        self.batch_size = self.inputs.size(self.inputs_batch_dim)
        self.gumbel = state.get('gumbel', False)
        self.active = state.get('active_seqs', torch.full((self.batch_size, 1), self.bos_idx, dtype=torch.long))
        self.finished = [None] * self.batch_size
        self.active_idx = torch.arange(self.batch_size)
        self.step_logits = []  # last-position logits per step (rows = active rows of that step)
        self.step_rows = []
        assert self.max_len > 0 and self.batch_size > 0
        unk = self.vocab.stoi[state.get('unk_token', '<unk>')]
        assert self.bos_idx != unk and self.eos_idx != unk

    def _select(self, logits):
        if self.sample_type == 'max':
            return logits.max(dim=-1)[1]
        if self.sample_type == 'sample':
            return torch.distributions.Categorical(logits=logits / self.temperature).sample()
        raise NotImplementedError

    def inference(self, keep_logits=0):
        with torch.no_grad():
            for step in range(self.max_len):
                args = [] if self.enc_mask is None else [self.enc_mask]
                if self.substitute_inputs:
                    logits, self.inputs = self.model(self.inputs, self.active[:, step:], *args, **self.kwargs)
                    logits = logits[:, -1]
                else:
                    logits = self.model(self.inputs, self.active, *args, **self.kwargs)[:, -1]
                if step < keep_logits:
                    self.step_logits.append(logits.clone())
                    self.step_rows.append(self.active_idx.clone())
                nxt = self._select(logits)
                nxt_x = nxt
                if self.gumbel:
                    # softmax of the logits, Gumbel noise from torch's global generator, softmax again
                    dists = F.softmax(logits, dim=1)
                    nxt_x = ref_gumbel_softmax(dists, torch.rand(dists.size()), self.temperature)
                done = nxt == self.eos_idx
                for i, seq in zip(self.active_idx[done].tolist(), self.active[done]):
                    self.finished[i] = seq
                self.active = torch.cat([self.active, nxt_x.unsqueeze(1)], dim=-1)[~done]
                self.active_idx = self.active_idx[~done]
                self.inputs = self.inputs.index_select(self.inputs_batch_dim, (~done).nonzero().squeeze(1))
                if self.enc_mask is not None:
                    self.enc_mask = self.enc_mask[~done]
                if all(s is not None for s in self.finished):
                    break
            for i, seq in zip(self.active_idx.tolist(), self.active):
                self.finished[i] = seq
        return self.finished


def ref_simple_inference(model, z, vocab, max_len=100, eos_token='<eos>'):
    """``simple_inference`` of ``/root/reference/src/inference.py:170-181``."""
    out = RefInferenceState({'model': model, 'inputs': z, 'vocab': vocab,
                             'max_len': max_len, 'eos_token': eos_token}).inference()
    return [x.cpu().numpy().tolist() for x in out]


def ref_beam_search(step_fn, n_sent, beam, max_len, bos, eos, alpha=0.0):
    """Beam search as this build specifies it (the reference has none:
    ``/root/reference/README.md:27,30``, ``/root/reference/src/inference.py:143-149``), written with
    torch ops for cross-checking the canonical oracle on small cases.

    ``step_fn(prefixes) -> logits[rows, V]`` gets a list of token lists (one per live+frozen row,
    sentence-major, ``beam`` rows per sentence) and returns last-position raw logits.

    Per step and sentence: every unfinished beam offers its ``beam`` best tokens by raw logit (ties:
    lowest id) with value ``score + (logit - logsumexp(logits))``; a finished beam offers itself with
    its score unchanged; at step 0 only beam 0 offers.  The ``beam`` best offers (ties: lowest beam,
    then lowest rank) become the new beams in that order.  A sentence ends when all its beams are
    finished or after ``max_len`` steps; the result is the beam maximising
    ``score / ((5 + n_emitted) / 6) ** alpha`` (ties: lowest beam), BOS-led with EOS stripped;
    ``n_emitted`` counts the emitted tokens, EOS included.
    """
    hyps = [[{'toks': [bos], 'score': 0.0 if j == 0 else -float('inf'), 'fin': False, 'n': 0}
             for j in range(beam)] for _ in range(n_sent)]
    alive = list(range(n_sent))
    results = [None] * n_sent
    for step in range(max_len):
        prefixes = [h['toks'] for s in alive for h in hyps[s]]
        logits = step_fn(prefixes, alive)
        lse = torch.logsumexp(logits.double(), dim=-1)
        still = []
        for a, s in enumerate(alive):
            offers = []
            for j, h in enumerate(hyps[s]):
                row = a * beam + j
                if h['fin']:
                    offers.append((h['score'], j, 0, None))
                    continue
                if step == 0 and j > 0:
                    continue
                vals, idx = torch.sort(logits[row], descending=True, stable=True)
                for rk in range(beam):
                    lp = torch.tensor(float(vals[rk]) - float(lse[row]), dtype=torch.float32)
                    sc = float(torch.tensor(h['score'], dtype=torch.float32) + lp)
                    offers.append((sc, j, rk, int(idx[rk])))
            offers.sort(key=lambda o: (-o[0], o[1], o[2]))
            new = []
            for sc, j, rk, tok in offers[:beam]:
                h = hyps[s][j]
                if tok is None:
                    new.append(dict(h, toks=list(h['toks'])))
                elif tok == eos:
                    new.append({'toks': list(h['toks']), 'score': sc, 'fin': True, 'n': h['n'] + 1})
                else:
                    new.append({'toks': h['toks'] + [tok], 'score': sc, 'fin': False, 'n': h['n'] + 1})
            hyps[s] = new
            if all(h['fin'] for h in new) or step == max_len - 1:
                def rank(h):
                    return h['score'] / (((5.0 + h['n']) / 6.0) ** alpha) if alpha != 0.0 else h['score']
                best = max(range(beam), key=lambda j: (rank(new[j]), -j))
                results[s] = (new[best]['toks'], new[best]['score'])
            else:
                still.append(s)
        alive = still
        if not alive:
            break
    return results
