"""Transformer encoder-decoder whose decode steps run on the HIP kernels.

Call surface as used by the reference (``/root/reference/src/trainers/transformer_mt_trainer.py:55,91-98``):
``Transformer(hp.transformer, vocab_src, vocab_trg)``, ``.encoder(src) -> (encs, enc_mask)``,
``.decoder(encs, ys, *enc_mask) -> [B, t, V]`` raw logits; ``hp`` carries ``d_model, n_layers, n_heads, d_ff,
pe_max_len, dropout`` (``/root/reference/experiments/transformer-mt/config.yml:21-27``).  The model source
of the reference is an absent submodule; the architecture is this build's own: pre-norm layers, sinusoidal
positions added to ``embedding * sqrt(d_model)``, final LayerNorm, linear output to raw logits.

Parameters are fp32 torch tensors on the GPU, named like an ``nn.Module`` of the same structure so that
``state_dict()`` / ``load_state_dict()`` interchange with ordinary checkpoints.  For the kernels the
weights are re-laid once: every ``nn.Linear`` weight transposed to [in, out] (the GEMM streams it along
``out``), q|k|v of self-attention fused to one [d, 3d] matrix, k|v of cross-attention to [d, 2d].
"""
import math
from collections import OrderedDict

import torch

from . import kernels as K


class HP(dict):
    """Attribute-access dict (the shape of a firelab config node)."""
    __getattr__ = dict.__getitem__


def sinusoid_table(n_pos, d_model):
    pos = torch.arange(n_pos, dtype=torch.float64).unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float64) * (-math.log(10000.0) / d_model))
    pe = torch.zeros(n_pos, d_model, dtype=torch.float64)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe.to(torch.float32)


def tensor_version(t):
    """Version counter of a tensor (moves with every in-place edit), or None for tensors that keep none (tensors made under
    ``torch.inference_mode()``): callers then assume nothing about the contents."""
    try:
        return t._version
    except RuntimeError:
        return None


class HipModule:
    """Minimal parameter container with the ``nn.Module`` methods the reference's callers use."""

    def __init__(self, device):
        self.device = torch.device(device)
        self._params = OrderedDict()
        self._prepared = None

    # The kernels read weights in their own layouts (transposed, concatenated, padded, bf16, per-token tables), made once by
    # prepared() and kept here.  An in-place edit of a parameter tensor (an optimizer step over parameters(), p.copy_(...))
    # bumps that tensor's version counter, which drops the kept copies, so the next call converts the edited weights.
    def _weights_token(self):
        return tuple(tensor_version(v) for v in self._params.values())

    @property
    def _prepared(self):
        d = self.__dict__
        if d.get('_prep') is not None and d.get('_prep_token') != self._weights_token():
            d['_prep'] = None
        return d.get('_prep')

    @_prepared.setter
    def _prepared(self, value):
        self.__dict__['_prep'] = value
        self.__dict__['_prep_token'] = None if value is None else self._weights_token()

    def _add(self, name, shape, std=None, fill=None):
        with torch.inference_mode(False):      # parameters are ordinary tensors (with version counters) wherever the model is made
            if fill is not None:
                t = torch.full(shape, float(fill), dtype=torch.float32)
            else:
                t = torch.randn(shape, dtype=torch.float32) * std
            self._params[name] = t.to(self.device)

    def state_dict(self):
        return OrderedDict((k, v.clone()) for k, v in self._params.items())

    def load_state_dict(self, sd, strict=True):
        missing = [k for k in self._params if k not in sd]
        extra = [k for k in sd if k not in self._params]
        if strict and (missing or extra):
            raise KeyError('state_dict mismatch: missing %s, unexpected %s' % (missing, extra))
        for k in self._params:
            if k in sd:
                if tuple(sd[k].shape) != tuple(self._params[k].shape):
                    raise ValueError('shape mismatch for %s: %s vs %s' % (k, tuple(sd[k].shape), tuple(self._params[k].shape)))
                with torch.inference_mode(False):
                    self._params[k] = sd[k].detach().to(device=self.device, dtype=torch.float32).contiguous().clone()
        self._prepared = None
        return self

    def parameters(self):
        return list(self._params.values())

    def eval(self):
        return self

    def train(self, mode=True):
        return self

    def to(self, device):
        self.device = torch.device(device)
        with torch.inference_mode(False):
            self._params = OrderedDict((k, v.to(self.device)) for k, v in self._params.items())
        self._prepared = None
        return self

    def cuda(self):
        return self.to('cuda')


def _t(w):
    """nn.Linear weight [out, in] -> [in, ld] with ld = out rounded up to a multiple of 4 (zero filled)."""
    out_f, in_f = w.shape
    ld = (out_f + 3) // 4 * 4
    wt = torch.zeros((in_f, ld), dtype=torch.float32, device=w.device)
    wt[:, :out_f] = w.t()
    return wt[:, :out_f]


def _pad_rows4(m):
    rows = (m.shape[0] + 3) // 4 * 4
    if rows == m.shape[0]:
        return m
    out = torch.zeros((rows, m.shape[1]), dtype=m.dtype, device=m.device)
    out[:m.shape[0]] = m
    return out


def _normalise_mask(enc_mask, B, S, device):
    """Accepts [B,1,S] / [B,S] bool or byte masks, or the B rows [1,S] that
    ``/root/reference/src/inference.py:124,132`` produces by star-unpacking; returns uint8 [B,S] or None."""
    if enc_mask is None or (isinstance(enc_mask, (tuple, list)) and len(enc_mask) == 0):
        return None
    if isinstance(enc_mask, (tuple, list)):
        enc_mask = enc_mask[0] if (len(enc_mask) == 1 and enc_mask[0].dim() == 3) else torch.stack(list(enc_mask))
    m = enc_mask.reshape(B, S)
    return m.to(device=device, dtype=torch.uint8).contiguous()


class TransformerDecoder(HipModule):
    def __init__(self, hp, vocab, device='cuda'):
        super().__init__(device)
        hp = HP(hp)
        self.hp, self.vocab = hp, vocab
        d, V, ff = hp.d_model, len(vocab), hp.d_ff
        self.d, self.V, self.heads, self.n_layers = d, V, hp.n_heads, hp.n_layers
        self.event_log = None  # set to a timing.LaunchTimer to time every vocabulary projection (work = flops)
        if d % hp.n_heads or (d // hp.n_heads) % 4 or d % 4:
            raise ValueError('d_model / n_heads must be a multiple of 4')
        s = 1.0 / math.sqrt(d)
        self._add('embed.weight', (V, d), std=s)
        with torch.inference_mode(False):
            self._params['pe'] = sinusoid_table(hp.pe_max_len, d).to(self.device)
        for i in range(hp.n_layers):
            p = 'layers.%d.' % i
            for ln, att in (('ln1', 'self_attn'), ('ln2', 'cross_attn')):
                self._add(p + ln + '.weight', (d,), fill=1.0)
                self._add(p + ln + '.bias', (d,), fill=0.0)
                for n in 'qkvo':
                    self._add(p + att + '.' + n + '.weight', (d, d), std=s)
                    self._add(p + att + '.' + n + '.bias', (d,), fill=0.0)
            self._add(p + 'ln3.weight', (d,), fill=1.0)
            self._add(p + 'ln3.bias', (d,), fill=0.0)
            self._add(p + 'ff.w1.weight', (ff, d), std=s)
            self._add(p + 'ff.w1.bias', (ff,), fill=0.0)
            self._add(p + 'ff.w2.weight', (d, ff), std=1.0 / math.sqrt(ff))
            self._add(p + 'ff.w2.bias', (d,), fill=0.0)
        self._add('ln.weight', (d,), fill=1.0)
        self._add('ln.bias', (d,), fill=0.0)
        self._add('out.weight', (V, d), std=s)
        self._add('out.bias', (V,), fill=0.0)
        # nn.Module orders ln1, self_attn, ln2, cross_attn, ln3, ff; keep that order for checkpoints
        self._params = OrderedDict((k, self._params[k]) for k in self._ordered_names())

    def _ordered_names(self):
        names = ['embed.weight']
        for i in range(self.n_layers):
            p = 'layers.%d.' % i
            for ln, att in (('ln1', 'self_attn'), ('ln2', 'cross_attn')):
                names += [p + ln + '.weight', p + ln + '.bias']
                for n in 'qkvo':
                    names += [p + att + '.' + n + '.weight', p + att + '.' + n + '.bias']
            names += [p + 'ln3.weight', p + 'ln3.bias', p + 'ff.w1.weight', p + 'ff.w1.bias', p + 'ff.w2.weight',
                      p + 'ff.w2.bias']
        return names + ['ln.weight', 'ln.bias', 'out.weight', 'out.bias', 'pe']

    def prepared(self):
        if self._prepared is None:
            P, W = self._params, {}
            W['emb'], W['pe'] = P['embed.weight'].contiguous(), P['pe'].contiguous()
            W['layers'] = []
            for i in range(self.n_layers):
                p = 'layers.%d.' % i
                L = {}
                for n in ('ln1', 'ln2', 'ln3'):
                    L[n] = (P[p + n + '.weight'], P[p + n + '.bias'])
                sa, ca = p + 'self_attn.', p + 'cross_attn.'
                L['wqkv'] = _t(torch.cat([P[sa + 'q.weight'], P[sa + 'k.weight'], P[sa + 'v.weight']], 0))
                L['bqkv'] = torch.cat([P[sa + 'q.bias'], P[sa + 'k.bias'], P[sa + 'v.bias']]).contiguous()
                L['wo'], L['bo'] = _t(P[sa + 'o.weight']), P[sa + 'o.bias']
                L['cwq'], L['cbq'] = _t(P[ca + 'q.weight']), P[ca + 'q.bias']
                L['cwkv'] = _t(torch.cat([P[ca + 'k.weight'], P[ca + 'v.weight']], 0))
                L['cbkv'] = torch.cat([P[ca + 'k.bias'], P[ca + 'v.bias']]).contiguous()
                L['cwo'], L['cbo'] = _t(P[ca + 'o.weight']), P[ca + 'o.bias']
                L['w1'], L['b1'] = _t(P[p + 'ff.w1.weight']), P[p + 'ff.w1.bias']
                L['w2'], L['b2'] = _t(P[p + 'ff.w2.weight']), P[p + 'ff.w2.bias']
                W['layers'].append(L)
            W['ln'] = (P['ln.weight'], P['ln.bias'])
            W['wout'], W['bout'] = _t(P['out.weight']), P['out.bias']
            self._prepared = W
        return self._prepared

    def soft_embedding_matrix(self):
        """Embedding matrix with the vocabulary padded to a multiple of 4 rows (zeros): the right-hand side of the
        product that embeds token distributions."""
        W = self.prepared()
        if 'emb_pad' not in W:
            W['emb_pad'] = _pad_rows4(W['emb'])
        return W['emb_pad']

    # ---- step-wise protocol used by the decode loop -------------------------------------------------
    def session(self, encs, enc_mask, rows_per_sentence, max_len):
        return TransformerSession(self, encs, enc_mask, rows_per_sentence, max_len)

    # ---- reference call protocol: whole-prefix logits ------------------------------------------------
    def __call__(self, encs, ys, *enc_mask):
        """encs [B,S,d], ys [B,t] ids -> raw logits [B,t,V] (teacher forcing; every position attends to its
        prefix).  Evaluated as t incremental steps over the KV cache."""
        B, t = ys.shape
        sess = self.session(encs, enc_mask, 1, t)
        rows = K.RowState(B, t, self.device, True, 0)
        rows.sent.copy_(torch.arange(B, dtype=torch.int32, device=self.device))
        rows.anc.copy_(torch.arange(B, dtype=torch.int32, device=self.device).unsqueeze(1).expand(B, t))
        out = torch.empty((B, t, self.V), dtype=torch.float32, device=self.device)
        ys32 = ys.to(device=self.device, dtype=torch.int32)
        for p in range(t):
            rows.tok.copy_(ys32[:, p])
            out[:, p] = sess.step(rows, B, p)[:B, :self.V]
        return out


class TransformerSession:
    """Per-batch decode state of a ``TransformerDecoder``: cross-attention k|v of the encoder outputs
    (computed once), the self-attention q|k|v cache [layer][position][row slot][3d] and work buffers."""
    needs_anc = True
    takes_soft = True
    incremental = True
    lag_ok = True     # rows beyond the true count may be stepped on stale state without harm (engine.DecodeLoop.lagged)

    def __init__(self, dec, encs, enc_mask, k, max_len):
        self.dec, self.V = dec, dec.V
        W = dec.prepared()
        dev = dec.device
        if max_len > W['pe'].shape[0]:
            raise ValueError('max_len %d exceeds pe_max_len %d' % (max_len, W['pe'].shape[0]))
        packed = getattr(encs, '_tst_packed', None)
        if (packed is None or packed is not getattr(enc_mask, '_tst_packed', None) or packed.rows.device != dev
                or None in packed.versions or packed.versions != (tensor_version(encs), tensor_version(enc_mask))):
            packed = None     # not the untouched outputs of this package's encoder (other objects, or edited in place): pack here
        encs = encs.to(device=dev, dtype=torch.float32).contiguous()
        B, S, d = encs.shape
        if d != dec.d:
            raise ValueError('encoder width %d != d_model %d' % (d, dec.d))
        self.B, self.S, self.k = B, S, k
        R = B * k
        self.R = R
        self.mask = packed.mask if packed is not None else _normalise_mask(enc_mask, B, S, dev)
        f32 = dict(dtype=torch.float32, device=dev)
        flat = encs.view(B * S, d)
        # k|v of the encoder outputs, for the non-padding source positions only: rows packed sentence after sentence,
        # kvrow[sentence, position] = packed row (a batch padded to its longest sentence is mostly padding)
        self.kvrow = None
        n_src = B * S
        if packed is not None:
            if 0 < packed.n < B * S:
                self.kvrow, flat, n_src = packed.kvrow, packed.rows, packed.n
        elif self.mask is not None:
            live = self.mask.view(-1) != 0
            idx = live.nonzero().squeeze(1)
            if 0 < idx.numel() < B * S:
                self.kvrow = (torch.cumsum(live, 0, dtype=torch.int32) - 1).view(B, S).contiguous()
                flat = flat.index_select(0, idx)
                n_src = int(idx.numel())
        self.kv_cross = []
        for L in W['layers']:
            kv = torch.empty((n_src, 2 * d), **f32)
            K.gemm(flat, L['cwkv'], L['cbkv'], kv, n_src)
            self.kv_cross.append(kv if self.kvrow is not None else kv.view(B, S, 2 * d))
        self.cache = [torch.empty((max_len, R, 3 * d), **f32) for _ in W['layers']]
        self.x = torch.empty((R, d), **f32)
        self.h = torch.empty((R, d), **f32)
        self.q = torch.empty((R, d), **f32)
        self.a = torch.empty((R, d), **f32)
        self.ff = torch.empty((R, dec.hp.d_ff), **f32)
        self.logits = torch.empty((R, (dec.V + 3) // 4 * 4), **f32)
        # k-major target of the final LayerNorm for steps of many rows (the vocabulary projection then stages both operands by
        # LDS-DMA); every row of it is long enough for the last 128-row tile
        self.ht = torch.empty((d, (R + 127) // 128 * 128), **f32) if (R >= K.DMA_MIN_ROWS and d % 16 == 0 and d <= 512) else None
        self.scale = float(torch.tensor(math.sqrt(d), dtype=torch.float32))
        self.max_len = max_len
        self._plans = {}
        self.rows_per_sentence = 1   # adjacent rows that belong to one sentence in the coming step (the loop updates it)

    def reorder(self, row_parent, n_rows, identity):
        pass  # the KV cache is addressed through the ancestor table; nothing moves

    def _plan(self, anc_ld):
        """The native launch plan of a step (one per ancestor-table width)."""
        plan = self._plans.get(anc_ld)
        if plan is None:
            dec, W = self.dec, self.dec.prepared()
            layers = []
            for li, L in enumerate(W['layers']):
                layers.append(dict(ln_self_g=L['ln1'][0], ln_self_b=L['ln1'][1], wqkv=L['wqkv'], bqkv=L['bqkv'], wo=L['wo'], bo=L['bo'],
                                   ln_cross_g=L['ln2'][0], ln_cross_b=L['ln2'][1], cwq=L['cwq'], cbq=L['cbq'], cwo=L['cwo'], cbo=L['cbo'],
                                   ln_ff_g=L['ln3'][0], ln_ff_b=L['ln3'][1], w1=L['w1'], b1=L['b1'], w2=L['w2'], b2=L['b2'],
                                   cache=self.cache[li], kv_cross=self.kv_cross[li]))
            plan = K.StepPlan(d=dec.d, heads=dec.heads, d_ff=dec.hp.d_ff, V=self.V, bf16=False, n_pos=self.max_len, slab_rows=self.R,
                              S=self.S, anc_ld=anc_ld, emb=W['emb'], pe=W['pe'], scale=self.scale, z=None, mask=self.mask, kvrow=self.kvrow,
                              ln=W['ln'],
                              wout=W['wout'], bout=W['bout'], x=self.x, h=self.h, q=self.q, a=self.a, ff=self.ff, logits=self.logits,
                              layers=layers, ht=self.ht)
            self._plans[anc_ld] = plan
        return plan

    def keys_ok(self, n):
        """Whether a step of n rows can end in arg-max keys instead of logits (``step(..., keys=)``)."""
        return self.dec.event_log is None and K.vocab_argmax_ok(n, self.V, self.dec.d)

    def step(self, rows, n, t, soft=None, keys=None):
        """Logits of the next token for rows 0..n-1 at position t.  ``soft`` ([>= n, Vp] fp32 token distributions,
        see ``kernels.soft_embed``) replaces the lookup of ``rows.tok`` by the expected embedding.  The launches of
        the step are issued by one native call (two when the vocabulary projection is being timed).
        ``keys`` (int64 [>= n], only when ``keys_ok(n)``): the step ends in the arg-max keys of its logits
        (``kernels.ln_vocab_argmax``) and the returned logits buffer is NOT written."""
        dec = self.dec
        plan = self._plan(rows.anc.shape[1])
        plan.c.group = self.rows_per_sentence      # set by the decode loop: 1 at the first step, then the beam size
        if keys is not None:
            if soft is not None or not self.keys_ok(n):
                raise ValueError('arg-max keys need token inputs and a shape keys_ok accepts')
            plan.set_keys(keys)
            try:
                plan.run(rows, n, t, K.STEP_EMBED | K.STEP_LAYERS | K.STEP_VOCAB)
            finally:
                plan.set_keys(None)
            return self.logits
        body = K.STEP_LAYERS
        if soft is None:
            body |= K.STEP_EMBED
        else:
            K.soft_embed(soft, dec.soft_embedding_matrix(), self.scale, dec.prepared()['pe'][t], self.x, n)
        log = dec.event_log
        if log is None:
            plan.run(rows, n, t, body | K.STEP_VOCAB)
        else:  # measurement aid: HIP events on the launch stream around the vocabulary projection
            plan.run(rows, n, t, body)
            log.begin()
            plan.run(rows, n, t, K.STEP_VOCAB)
            log.end(2.0 * n * dec.d * self.V)
        return self.logits


class Transformer(HipModule):
    """``Transformer(hp, vocab_src, vocab_trg)``; the encoder half runs the same kernels at prefill shapes."""

    def __init__(self, hp, vocab_src, vocab_trg, device='cuda'):
        super().__init__(device)
        from .encoder import TransformerEncoder
        self.encoder = TransformerEncoder(hp, vocab_src, device)
        self.decoder = TransformerDecoder(hp, vocab_trg, device)

    def state_dict(self):
        sd = OrderedDict(('encoder.' + k, v) for k, v in self.encoder.state_dict().items())
        sd.update(('decoder.' + k, v) for k, v in self.decoder.state_dict().items())
        return sd

    def load_state_dict(self, sd, strict=True):
        self.encoder.load_state_dict({k[8:]: v for k, v in sd.items() if k.startswith('encoder.')}, strict)
        self.decoder.load_state_dict({k[8:]: v for k, v in sd.items() if k.startswith('decoder.')}, strict)
        return self

    def parameters(self):
        return self.encoder.parameters() + self.decoder.parameters()

    def to(self, device):
        self.encoder.to(device)
        self.decoder.to(device)
        return self

    def __call__(self, src, trg):
        encs, enc_mask = self.encoder(src)
        return self.decoder(encs, trg[:, :-1], enc_mask)
