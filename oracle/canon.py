"""TEST INFRASTRUCTURE — Python side of the canonical-arithmetic oracle (``oracle/canon.c``).

Only tests/, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import this.

The arithmetic (GEMMs, LayerNorm, attention, GRU gates, log-sum-exp, top-k) is done by the C library;
this file holds the model wiring with an incremental KV cache / carried GRU state, the greedy decode
loop restating ``/root/reference/src/inference.py:94-167`` and the build-defined beam search (the
reference has none: ``/root/reference/README.md:27,30``).  KV caches are kept per hypothesis and are
physically re-gathered on every batch compaction / beam reorder, i.e. without the slot indirection the
GPU path uses, so that the two implementations only share the specification.
"""
import ctypes
import math
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libtst_oracle.so')
GCC_CMD = ['gcc', '-O3', '-fPIC', '-shared', '-ffp-contract=off', '-mavx2', '-mfma', '-fopenmp',
           '-o', LIB_PATH, os.path.join(HERE, 'canon.c'), '-lm']


def build(force=False):
    src = os.path.join(HERE, 'canon.c')
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(GCC_CMD)
    return LIB_PATH


_lib = None
_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_i64 = ctypes.c_int64


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.orc_exp.restype = ctypes.c_float
        _lib.orc_exp.argtypes = [ctypes.c_float]
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_i32p)


def fmap(which, x):
    """which: 0 exp, 1 log, 2 sigmoid, 3 tanh"""
    x, px = _f(x)
    y = np.empty_like(x)
    lib().orc_map(ctypes.c_int(which), px, y.ctypes.data_as(_f32p), _i64(x.size))
    return y


def gemm(A, Wt, bias=None, res=None, act=0):
    """A [M,K] @ Wt [K,N] (+bias) -> relu? -> (+res)"""
    A, pA = _f(A)
    Wt, pW = _f(Wt)
    M, K = A.shape
    N = Wt.shape[1]
    C = np.empty((M, N), dtype=np.float32)
    pb = pr = None
    if bias is not None:
        bias, pb = _f(bias)
    if res is not None:
        res, pr = _f(res)
    lib().orc_gemm(pA, _i64(K), pW, _i64(N), pb, pr, _i64(N), C.ctypes.data_as(_f32p), _i64(N), _i64(M), _i64(N),
                   _i64(K), ctypes.c_int(act))
    return C


def embed_posadd(tok, emb, pe_row, scale):
    tok, pt = _i(tok)
    emb, pe_ = _f(emb)
    d = emb.shape[1]
    out = np.empty((tok.shape[0], d), dtype=np.float32)
    pp = None
    if pe_row is not None:
        pe_row, pp = _f(pe_row)
    lib().orc_embed_posadd(pt, pe_, pp, ctypes.c_float(scale), out.ctypes.data_as(_f32p), _i64(tok.shape[0]), _i64(d))
    return out


def layernorm(x, g, b, eps=1e-5):
    x, px = _f(x)
    g, pg = _f(g)
    b, pb = _f(b)
    y = np.empty_like(x)
    lib().orc_layernorm(px, pg, pb, ctypes.c_float(eps), y.ctypes.data_as(_f32p), _i64(x.shape[0]), _i64(x.shape[1]))
    return y


def attention(q, K, V, kv_row, mask, n, heads, chain=False):
    """q [R,d]; K,V [E,n_stride,d]; kv_row [R] or None; mask [E,n_stride] (nonzero = attend) or None.
    ``chain``: the score dot products are single fma chains instead of 16 partial sums (the order of the matrix-core
    cross-attention kernel, which is not on the decode path)."""
    q, pq = _f(q)
    K, pK = _f(K)
    V, pV = _f(V)
    R, d = q.shape
    dh = d // heads
    out = np.empty_like(q)
    pk = pm = None
    if kv_row is not None:
        kv_row, pk = _i(kv_row)
    if mask is not None:
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        pm = mask.ctypes.data_as(_u8p)
    lib().orc_attention(pq, pK, pV, pk, pm, _i64(R), _i64(n), _i64(K.shape[1]), _i64(heads), _i64(dh),
                        ctypes.c_float(np.float32(1.0 / math.sqrt(dh))), out.ctypes.data_as(_f32p), ctypes.c_int(int(bool(chain))))
    return out


def gru_cell(gi, gh, h):
    gi, pi = _f(gi)
    gh, ph = _f(gh)
    h, phh = _f(h)
    out = np.empty_like(h)
    lib().orc_gru_cell(pi, ph, phh, out.ctypes.data_as(_f32p), _i64(h.shape[0]), _i64(h.shape[1]))
    return out


def argmax_rows(x):
    x, px = _f(x)
    out = np.empty(x.shape[0], dtype=np.int32)
    lib().orc_argmax_rows(px, _i64(x.shape[1]), _i64(x.shape[0]), _i64(x.shape[1]), out.ctypes.data_as(_i32p))
    return out


def logsoftmax_topk(x, k):
    x, px = _f(x)
    R, V = x.shape
    val = np.empty((R, k), dtype=np.float32)
    idx = np.empty((R, k), dtype=np.int32)
    lse = np.empty(R, dtype=np.float32)
    lib().orc_logsoftmax_topk(px, _i64(V), _i64(R), _i64(V), ctypes.c_int(k), val.ctypes.data_as(_f32p),
                              idx.ctypes.data_as(_i32p), lse.ctypes.data_as(_f32p))
    return val, idx, lse


def sample_rows(x, temperature, seed, step, sent):
    x, px = _f(x)
    sent, ps = _i(sent)
    out = np.empty(x.shape[0], dtype=np.int32)
    lib().orc_sample_rows(px, _i64(x.shape[1]), _i64(x.shape[0]), _i64(x.shape[1]), ctypes.c_float(temperature),
                          ctypes.c_uint32(seed), ctypes.c_uint32(step), ps, out.ctypes.data_as(_i32p))
    return out


# --------------------------------------------------------------------------------------------
# models (weights come from a torch state_dict; Linear weights are stored transposed, [in, out])
# --------------------------------------------------------------------------------------------
def _np(t):
    return np.ascontiguousarray(t.detach().cpu().numpy().astype(np.float32))


class _Lin:
    def __init__(self, sd, prefix):
        self.wt = np.ascontiguousarray(_np(sd[prefix + '.weight']).T)
        self.b = _np(sd[prefix + '.bias'])

    def __call__(self, x, res=None, act=0):
        return gemm(x, self.wt, self.b, res, act)


class CanonTransformerDecoder:
    """Incremental (KV-cached) evaluation of ``oracle.torch_ref.TransformerDecoder``."""

    def __init__(self, torch_decoder, n_heads):
        sd = torch_decoder.state_dict()
        self.d = sd['embed.weight'].shape[1]
        self.heads = n_heads
        self.emb = _np(sd['embed.weight'])
        self.pe = _np(sd['pe'])
        self.scale = float(np.float32(math.sqrt(self.d)))
        self.layers = []
        i = 0
        while 'layers.%d.ln1.weight' % i in sd:
            p = 'layers.%d.' % i
            L = {n: (_np(sd[p + n + '.weight']), _np(sd[p + n + '.bias'])) for n in ('ln1', 'ln2', 'ln3')}
            for n in ('q', 'k', 'v', 'o'):
                L['s' + n] = _Lin(sd, p + 'self_attn.' + n)
                L['c' + n] = _Lin(sd, p + 'cross_attn.' + n)
            L['w1'], L['w2'] = _Lin(sd, p + 'ff.w1'), _Lin(sd, p + 'ff.w2')
            self.layers.append(L)
            i += 1
        self.ln = (_np(sd['ln.weight']), _np(sd['ln.bias']))
        self.out = _Lin(sd, 'out')
        self.V = self.out.b.shape[0]

    def session(self, encs, enc_mask, rows_per_sentence, max_len):
        return _TransformerSession(self, np.asarray(encs, np.float32), enc_mask, rows_per_sentence, max_len)


class _TransformerSession:
    def __init__(self, m, encs, enc_mask, k, max_len):
        self.m = m
        B, S, d = encs.shape
        self.mask = None if enc_mask is None else np.asarray(enc_mask).reshape(B, S).astype(np.uint8)
        flat = encs.reshape(B * S, d)
        self.ck = [L['ck'](flat).reshape(B, S, d) for L in m.layers]
        self.cv = [L['cv'](flat).reshape(B, S, d) for L in m.layers]
        self.S = S
        R = B * k
        self.k = k
        self.sent = np.repeat(np.arange(B, dtype=np.int32), k)
        self.sk = [np.zeros((R, max_len, d), np.float32) for _ in m.layers]
        self.sv = [np.zeros((R, max_len, d), np.float32) for _ in m.layers]

    def reorder(self, idx):
        idx = np.asarray(idx, dtype=np.int64)
        self.sent = self.sent[idx]
        self.sk = [a[idx] for a in self.sk]
        self.sv = [a[idx] for a in self.sv]

    def step(self, tok, t):
        m = self.m
        x = embed_posadd(tok, m.emb, m.pe[t], m.scale)
        for li, L in enumerate(m.layers):
            h = layernorm(x, *L['ln1'])
            q = L['sq'](h)
            self.sk[li][:, t] = L['sk'](h)
            self.sv[li][:, t] = L['sv'](h)
            a = attention(q, self.sk[li], self.sv[li], None, None, t + 1, m.heads)
            x = L['so'](a, res=x)
            h = layernorm(x, *L['ln2'])
            a = attention(L['cq'](h), self.ck[li], self.cv[li], self.sent, self.mask, self.S, m.heads)
            x = L['co'](a, res=x)
            h = layernorm(x, *L['ln3'])
            x = L['w2'](L['w1'](h, act=1), res=x)
        return m.out(layernorm(x, *m.ln))


def bf16_round(a):
    """fp32 array rounded to the nearest bf16 (ties to even), returned as fp32."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)) << np.uint32(16)
    return r.astype(np.uint32).view(np.float32)


class CanonTransformerLM:
    """Incremental (KV-cached) evaluation of ``oracle.torch_ref.TransformerLM``.  ``bf16=True`` emulates the
    bf16 storage points of the GPU's bf16 path on the CPU: weights, LayerNorm outputs, the q|k|v cache,
    attention outputs, the feed-forward activation and the logits are rounded to bf16; all sums stay fp32.
    (The matrix cores add the 16 products of one instruction in their own order, so the GPU's bf16 results
    agree with this emulation up to an occasional bf16 rounding flip, not bit for bit.)"""

    def __init__(self, torch_lm, bf16=False):
        sd = torch_lm.state_dict()
        self.bf16 = bf16
        self.d = sd['embed.weight'].shape[1]
        self.heads = torch_lm.n_heads
        self.emb, self.pe = _np(sd['embed.weight']), _np(sd['pe'])
        self.scale = float(np.float32(math.sqrt(self.d)))
        self.layers = []
        i = 0
        while 'layers.%d.ln1.weight' % i in sd:
            p = 'layers.%d.' % i
            L = {n: (_np(sd[p + n + '.weight']), _np(sd[p + n + '.bias'])) for n in ('ln1', 'ln2')}
            for n in ('q', 'k', 'v', 'o'):
                L[n] = self._lin(sd, p + 'self_attn.' + n)
            L['w1'], L['w2'] = self._lin(sd, p + 'ff.w1'), self._lin(sd, p + 'ff.w2')
            self.layers.append(L)
            i += 1
        self.ln = (_np(sd['ln.weight']), _np(sd['ln.bias']))
        self.out = self._lin(sd, 'out')
        self.V = self.out.b.shape[0]

    def _lin(self, sd, prefix):
        lin = _Lin(sd, prefix)
        if self.bf16:
            lin.wt = bf16_round(lin.wt)
        return lin

    def rnd(self, a):
        return bf16_round(a) if self.bf16 else a

    def session(self, z, rows_per_sentence=1, max_len=512):
        return _LMSession(self, np.asarray(z, np.float32), rows_per_sentence, max_len)


class _LMSession:
    def __init__(self, m, z, k, max_len):
        self.m = m
        B, d = z.shape
        self.z = np.repeat(z, k, axis=0)
        R = B * k
        self.sk = [np.zeros((R, max_len, d), np.float32) for _ in m.layers]
        self.sv = [np.zeros((R, max_len, d), np.float32) for _ in m.layers]

    def reorder(self, idx):
        idx = np.asarray(idx, dtype=np.int64)
        self.z = self.z[idx]
        self.sk = [a[idx] for a in self.sk]
        self.sv = [a[idx] for a in self.sv]

    def step(self, tok, t):
        m, rnd = self.m, self.m.rnd
        x = embed_posadd(tok, m.emb, m.pe[t], m.scale) + self.z
        for li, L in enumerate(m.layers):
            h = rnd(layernorm(x, *L['ln1']))
            q = rnd(L['q'](h))
            self.sk[li][:, t] = rnd(L['k'](h))
            self.sv[li][:, t] = rnd(L['v'](h))
            a = rnd(attention(q, self.sk[li], self.sv[li], None, None, t + 1, m.heads))
            x = L['o'](a, res=x)
            h = rnd(layernorm(x, *L['ln2']))
            x = L['w2'](rnd(L['w1'](h, act=1)), res=x)
        return rnd(m.out(rnd(layernorm(x, *m.ln))))


class CanonTransformerEncoder:
    """Evaluation of ``oracle.torch_ref.TransformerEncoder`` with the canonical arithmetic."""

    def __init__(self, torch_encoder, n_heads):
        sd = torch_encoder.state_dict()
        self.d = sd['embed.weight'].shape[1]
        self.heads, self.pad_idx = n_heads, torch_encoder.pad_idx
        self.emb, self.pe = _np(sd['embed.weight']), _np(sd['pe'])
        self.scale = float(np.float32(math.sqrt(self.d)))
        self.layers = []
        i = 0
        while 'layers.%d.ln1.weight' % i in sd:
            p = 'layers.%d.' % i
            L = {n: (_np(sd[p + n + '.weight']), _np(sd[p + n + '.bias'])) for n in ('ln1', 'ln2')}
            for n in ('q', 'k', 'v', 'o'):
                L[n] = _Lin(sd, p + 'attn.' + n)
            L['w1'], L['w2'] = _Lin(sd, p + 'ff.w1'), _Lin(sd, p + 'ff.w2')
            self.layers.append(L)
            i += 1
        self.ln = (_np(sd['ln.weight']), _np(sd['ln.bias']))

    def __call__(self, src):
        src = np.asarray(src)
        B, S = src.shape
        mask = (src != self.pad_idx)
        x = np.concatenate([embed_posadd(src[:, s], self.emb, self.pe[s], self.scale)[:, None] for s in range(S)], 1)
        x = x.reshape(B * S, self.d)
        sent = np.repeat(np.arange(B, dtype=np.int32), S)
        for L in self.layers:
            h = layernorm(x, *L['ln1'])
            k = L['k'](h).reshape(B, S, self.d)
            v = L['v'](h).reshape(B, S, self.d)
            a = attention(L['q'](h), k, v, sent, mask, S, self.heads)
            x = L['o'](a, res=x)
            x = L['w2'](L['w1'](layernorm(x, *L['ln2']), act=1), res=x)
        return layernorm(x, *self.ln).reshape(B, S, self.d), mask[:, None, :]


class CanonGRUDecoder:
    """Carried-state evaluation of ``oracle.torch_ref.RNNDecoder`` / ``RNNLM`` (n_layers GRU + Linear)."""

    def __init__(self, torch_module):
        sd = torch_module.state_dict()
        self.emb = _np(sd['embed.weight'])
        self.cells = []
        i = 0
        while 'gru.weight_ih_l%d' % i in sd:
            self.cells.append(dict(
                wih=np.ascontiguousarray(_np(sd['gru.weight_ih_l%d' % i]).T), bih=_np(sd['gru.bias_ih_l%d' % i]),
                whh=np.ascontiguousarray(_np(sd['gru.weight_hh_l%d' % i]).T), bhh=_np(sd['gru.bias_hh_l%d' % i])))
            i += 1
        self.out = _Lin(sd, 'out')
        self.V = self.out.b.shape[0]
        self.style_emb = _np(sd['style_embed.weight']) if 'style_embed.weight' in sd else None   # ConditionalLM

    def session(self, z, rows_per_sentence=1, style=None):
        """``style`` (ConditionalLM): one style index for the batch or one per sentence; the style embedding is added to
        every token embedding (one fp32 add per element)."""
        z = np.asarray(z, np.float32)
        if z.ndim == 2:
            z = z[None]
        sess = _GRUSession(self, np.repeat(z, rows_per_sentence, axis=1))
        if self.style_emb is not None:
            styles = np.broadcast_to(np.asarray(0 if style is None else style, np.int64).reshape(-1), (z.shape[1],))
            sess.extra = np.repeat(self.style_emb[styles], rows_per_sentence, axis=0)
        return sess


def _gru_weights(sd, prefix='gru.'):
    return dict(wih=np.ascontiguousarray(_np(sd[prefix + 'weight_ih_l0']).T), bih=_np(sd[prefix + 'bias_ih_l0']),
                whh=np.ascontiguousarray(_np(sd[prefix + 'weight_hh_l0']).T), bhh=_np(sd[prefix + 'bias_hh_l0']))


def rnn_encoder(torch_module, x):
    """``oracle.torch_ref.RNNEncoder`` in canonical arithmetic: final state of the GRU over all positions of x [B, T]."""
    sd = torch_module.state_dict()
    c, emb = _gru_weights(sd), _np(sd['embed.weight'])
    x = np.asarray(x)
    h = np.zeros((x.shape[0], c['whh'].shape[0]), np.float32)
    for t in range(x.shape[1]):
        gi = gemm(embed_posadd(x[:, t].astype(np.int32), emb, None, 1.0), c['wih'], c['bih'])
        h = gru_cell(gi, gemm(h, c['whh'], c['bhh']), h)
    return h


def generator(torch_module, z):
    """``oracle.torch_ref.Generator`` in canonical arithmetic."""
    c = _gru_weights(torch_module.state_dict())
    z = np.ascontiguousarray(np.asarray(z, np.float32))
    gi = gemm(z, c['wih'], c['bih'])
    h = z.copy()
    for _ in range(torch_module.n_steps):
        h = gru_cell(gi, gemm(h, c['whh'], c['bhh']), h)
    return h


def ffn(torch_module, x):
    """``oracle.torch_ref.FFN`` in canonical arithmetic (k-ordered fma chains; widths padded with zeros to a multiple
    of 4 add only fma(0, 0, acc) terms, which change nothing)."""
    h = np.ascontiguousarray(np.asarray(x, np.float32))
    n = len(torch_module.layers)
    for i, layer in enumerate(torch_module.layers):
        w = layer.weight.detach().numpy().astype(np.float32)
        h = gemm(h, np.ascontiguousarray(w.T), layer.bias.detach().numpy().astype(np.float32), None, 0 if i == n - 1 else 1)
    return h


class _GRUSession:
    def __init__(self, m, h):
        self.m, self.h = m, [np.ascontiguousarray(h[i]) for i in range(h.shape[0])]

    extra = None   # per-row vector added to the token embedding

    def reorder(self, idx):
        idx = np.asarray(idx, dtype=np.int64)
        self.h = [h[idx] for h in self.h]
        if self.extra is not None:
            self.extra = self.extra[idx]

    def step(self, tok, t):
        m = self.m
        x = embed_posadd(tok, m.emb, None, 1.0)
        if self.extra is not None:
            x = (x + self.extra[:x.shape[0]]).astype(np.float32)
        for i, c in enumerate(m.cells):
            gi = gemm(x, c['wih'], c['bih'])
            gh = gemm(self.h[i], c['whh'], c['bhh'])
            self.h[i] = gru_cell(gi, gh, self.h[i])
            x = self.h[i]
        return m.out(x)


# --------------------------------------------------------------------------------------------
# decode loops
# --------------------------------------------------------------------------------------------
def greedy_decode(sess, B, max_len, bos, eos, keep_logits=0, sample=None):
    """Greedy loop with the data flow of /root/reference/src/inference.py:154-167: rows whose new token
    is EOS are stored without it and leave the batch; rows alive after ``max_len`` steps are stored as
    they are (1 + max_len tokens).  ``sample=(temperature, seed)`` replaces the arg-max by the sampling of
    ``:147``.  Returns (sequences, [logits per kept step], [rows per kept step])."""
    active = [[bos] for _ in range(B)]
    idx = np.arange(B)
    finished = [None] * B
    cur = np.full(B, bos, dtype=np.int32)
    logs, rows = [], []
    for t in range(max_len):
        logits = sess.step(cur, t)
        if t < keep_logits:
            logs.append(logits.copy())
            rows.append(idx.copy())
        if sample is None:
            nxt = argmax_rows(logits)
        else:  # sample = (temperature, seed): Gumbel-max draw keyed by the original sentence index
            nxt = sample_rows(logits, sample[0], sample[1], t, idx)
        done = nxt == eos
        for i, row in enumerate(idx):
            if done[i]:
                finished[row] = active[i]
        keep = np.nonzero(~done)[0]
        active = [active[i] + [int(nxt[i])] for i in keep]
        idx = idx[keep]
        cur = nxt[keep]
        if len(keep) == 0:
            break
        if len(keep) != len(done):
            sess.reorder(keep)
    for i, row in enumerate(idx):
        finished[row] = active[i]
    return finished, logs, rows


def length_penalty_table(max_len, alpha):
    """lp[n] = ((5 + n) / 6) ** alpha for n emitted tokens, rounded to fp32 (all ones for alpha == 0)."""
    return np.array([((5.0 + n) / 6.0) ** alpha for n in range(max_len + 2)], dtype=np.float32)


def beam_decode(sess, B, beam, max_len, bos, eos, alpha=0.0, keep_logits=0):
    """Beam search as specified in ``oracle.torch_ref.ref_beam_search``; all score arithmetic in fp32:
    offer = score + (logit - lse).  Finished sentences leave the batch.  Returns
    (sequences, scores, [logits per kept step], [sentences per kept step])."""
    f32 = np.float32
    lp = length_penalty_table(max_len, alpha)
    neg = f32(-np.inf)
    hyps = [[dict(toks=[bos], score=f32(0.0) if j == 0 else neg, fin=False, n=0) for j in range(beam)] for _ in range(B)]
    alive = list(range(B))
    results, scores = [None] * B, np.zeros(B, dtype=np.float32)
    logs, sents = [], []
    for t in range(max_len):
        cur = np.array([h['toks'][-1] if not h['fin'] else eos for s in alive for h in hyps[s]], dtype=np.int32)
        logits = sess.step(cur, t)
        if t < keep_logits:
            logs.append(logits.copy())
            sents.append(np.array(alive))
        val, idx, lse = logsoftmax_topk(logits, beam)
        parents, still = [], []
        for a, s in enumerate(alive):
            offers = []
            for j, h in enumerate(hyps[s]):
                row = a * beam + j
                if h['fin']:
                    offers.append((h['score'], j, 0, None))
                elif t == 0 and j > 0:
                    continue
                else:
                    for rk in range(beam):
                        offers.append((f32(h['score'] + f32(val[row, rk] - lse[row])), j, rk, int(idx[row, rk])))
            offers.sort(key=lambda o: (-o[0], o[1], o[2]))
            new = []
            for sc, j, rk, tok in offers[:beam]:
                h = hyps[s][j]
                if tok is None:
                    new.append(dict(h, toks=list(h['toks'])))
                elif tok == eos:
                    new.append(dict(toks=list(h['toks']), score=sc, fin=True, n=h['n'] + 1))
                else:
                    new.append(dict(toks=h['toks'] + [tok], score=sc, fin=False, n=h['n'] + 1))
                new[-1]['parent'] = a * beam + j
            hyps[s] = new
            if all(h['fin'] for h in new) or t == max_len - 1:
                best = max(range(beam), key=lambda j: (f32(new[j]['score'] / lp[new[j]['n']]), -j))
                results[s], scores[s] = new[best]['toks'], new[best]['score']
            else:
                still.append(s)
                parents += [h['parent'] for h in new]
        alive = still
        if not alive:
            break
        sess.reorder(parents)
    return results, scores, logs, sents
