"""Transformer language model / auto-encoder decoder conditioned on a vector ``z``, stepped on the HIP kernels.

Call surface as used by the reference (``/root/reference/src/trainers/lm_trainer.py:54,72-74,86``):
``TransformerLM(hp.transformer, vocab)``, ``lm(z [B, model_size], x [B, T]) -> [B, T, V]`` raw logits and
``lm.inference(src, vocab, eos_token=..., max_len=...)``; the only size the reference's config gives is
``model_size: 512`` (``/root/reference/experiments/lm/config.yml:25-26``).  The model source of the reference
is an absent submodule; the architecture is this build's own: decoder-only pre-norm layers (self-attention
+ feed-forward), input ``embedding * sqrt(d) + position + z`` (the conditioning vector is added at every
position), final LayerNorm, linear output to raw logits.  Defaults: 6 layers, 8 heads, d_ff = 4 d.

``dtype='bf16'`` stores weights, GEMM operands, the q|k|v cache and the logits in bf16 (fp32 accumulation on
the bf16 matrix cores, fp32 residual stream, LayerNorm statistics, softmax and scores); ``'fp32'`` runs the
fixed-order fp32 kernels.
"""
import math
import os

import torch

from . import kernels as K
from .transformer import _pad_rows4, HP, HipModule, _t, sinusoid_table


# TST_VOCAB_STATS=1: beam search on bf16 logits takes the statistics route described in TransformerLMSession.__init__.  Off by
# default: measured on config 4 the exponentials in the projection's epilogue cost what the shorter selection pass saves.
VOCAB_STATS = os.environ.get('TST_VOCAB_STATS', '0') == '1'


class TransformerLM(HipModule):
    def __init__(self, hp, vocab, device='cuda', dtype=None):
        super().__init__(device)
        hp = HP(hp)
        d = hp.get('model_size', hp.get('d_model'))
        if d is None:
            raise KeyError('hp.model_size')
        self.hp, self.vocab = hp, vocab
        self.d, self.V = d, len(vocab)
        self.n_layers, self.heads = hp.get('n_layers', 6), hp.get('n_heads', 8)
        self.d_ff, self.pe_max_len = hp.get('d_ff', 4 * d), hp.get('pe_max_len', 512)
        self.dtype = dtype or hp.get('dtype', 'fp32')
        if self.dtype not in ('fp32', 'bf16'):
            raise ValueError("dtype must be 'fp32' or 'bf16'")
        if d % self.heads or (d // self.heads) % 4 or d % 8 or self.d_ff % 8:
            raise ValueError('model_size / n_heads must be a multiple of 4, model_size and d_ff multiples of 8')
        self.event_log = None  # set to a timing.LaunchTimer to time every vocabulary projection (work = flops)
        s = 1.0 / math.sqrt(d)
        self._add('embed.weight', (self.V, d), std=s)
        for i in range(self.n_layers):
            p = 'layers.%d.' % i
            self._add(p + 'ln1.weight', (d,), fill=1.0)
            self._add(p + 'ln1.bias', (d,), fill=0.0)
            for n in 'qkvo':
                self._add(p + 'self_attn.' + n + '.weight', (d, d), std=s)
                self._add(p + 'self_attn.' + n + '.bias', (d,), fill=0.0)
            self._add(p + 'ln2.weight', (d,), fill=1.0)
            self._add(p + 'ln2.bias', (d,), fill=0.0)
            self._add(p + 'ff.w1.weight', (self.d_ff, d), std=s)
            self._add(p + 'ff.w1.bias', (self.d_ff,), fill=0.0)
            self._add(p + 'ff.w2.weight', (d, self.d_ff), std=1.0 / math.sqrt(self.d_ff))
            self._add(p + 'ff.w2.bias', (d,), fill=0.0)
        self._add('ln.weight', (d,), fill=1.0)
        self._add('ln.bias', (d,), fill=0.0)
        self._add('out.weight', (self.V, d), std=s)
        self._add('out.bias', (self.V,), fill=0.0)
        with torch.inference_mode(False):
            self._params['pe'] = sinusoid_table(self.pe_max_len, d).to(self.device)

    # kernel-side weight copies are kept per arithmetic type, so switching back and forth converts nothing twice;
    # assigning None (load_state_dict, to) drops them all
    @property
    def _prepared(self):
        store = self.__dict__.setdefault('_prepared_by_dtype', {})
        if store and self.__dict__.get('_prep_token') != self._weights_token():     # a parameter was edited in place
            store.clear()
        return store.get(self.__dict__.get('dtype'))

    @_prepared.setter
    def _prepared(self, value):
        store = self.__dict__.setdefault('_prepared_by_dtype', {})
        if value is None:
            store.clear()
        else:
            store[self.dtype] = value
            self.__dict__['_prep_token'] = self._weights_token()

    def prepared(self):
        """Weights in kernel layout: fp32 -> transposed [in, out]; bf16 -> as stored [out, in], rounded to bf16."""
        if self._prepared is None:
            P, W = self._params, {}
            bf = self.dtype == 'bf16'
            lay = (lambda w: w.to(torch.bfloat16).contiguous()) if bf else _t
            W['emb'], W['pe'] = P['embed.weight'].contiguous(), P['pe'].contiguous()
            W['layers'] = []
            for i in range(self.n_layers):
                p = 'layers.%d.' % i
                sa = p + 'self_attn.'
                L = {'ln1': (P[p + 'ln1.weight'], P[p + 'ln1.bias']), 'ln2': (P[p + 'ln2.weight'], P[p + 'ln2.bias'])}
                L['wqkv'] = lay(torch.cat([P[sa + 'q.weight'], P[sa + 'k.weight'], P[sa + 'v.weight']], 0))
                L['bqkv'] = torch.cat([P[sa + 'q.bias'], P[sa + 'k.bias'], P[sa + 'v.bias']]).contiguous()
                L['wo'], L['bo'] = lay(P[sa + 'o.weight']), P[sa + 'o.bias']
                L['w1'], L['b1'] = lay(P[p + 'ff.w1.weight']), P[p + 'ff.w1.bias']
                L['w2'], L['b2'] = lay(P[p + 'ff.w2.weight']), P[p + 'ff.w2.bias']
                W['layers'].append(L)
            W['ln'] = (P['ln.weight'], P['ln.bias'])
            W['wout'], W['bout'] = lay(P['out.weight']), P['out.bias']
            self._prepared = W
        return self._prepared

    # ---- step-wise protocol used by the decode loop -------------------------------------------------
    def set_dtype(self, dtype):
        """Switches between fp32 and bf16 arithmetic (the parameters stay fp32; the kernel-side copies are rebuilt)."""
        if dtype not in ('fp32', 'bf16'):
            raise ValueError("dtype must be 'fp32' or 'bf16'")
        self.dtype = dtype

    def session(self, z, enc_mask=None, rows_per_sentence=1, max_len=None):
        return TransformerLMSession(self, z, rows_per_sentence, max_len if max_len is not None else self.pe_max_len)

    # ---- reference call protocol: whole-prefix logits ------------------------------------------------
    def __call__(self, z, x):
        """z [B, model_size], x [B, t] ids -> raw logits [B, t, V] (every position attends to its prefix);
        evaluated as t incremental steps over the KV cache."""
        B, t = x.shape
        sess = self.session(z, None, 1, t)
        rows = _identity_rows(B, t, self.device)
        out = torch.empty((B, t, self.V), dtype=torch.float32, device=self.device)
        x32 = x.to(device=self.device, dtype=torch.int32)
        for p in range(t):
            rows.tok.copy_(x32[:, p])
            out[:, p] = sess.step(rows, B, p)[:B, :self.V]
        return out

    def inference(self, src, vocab=None, eos_token='<eos>', max_len=100):
        """Greedy continuation of one token sequence ``src`` (1-D ids) with z = 0, as the reference's
        validation calls it (``/root/reference/src/trainers/lm_trainer.py:86``): the prefix is fed through
        the KV cache, then up to ``max_len`` tokens are generated until ``eos_token``.  Returns the
        generated ids (prefix and EOS not included) as a 1-D LongTensor.  What the absent reference model
        returns exactly is unknown; this format is this build's choice."""
        vocab = vocab or self.vocab
        eos = vocab.stoi[eos_token]
        src = torch.as_tensor(src).reshape(-1).to(device=self.device, dtype=torch.int32)
        n_prefix = int(src.numel())
        if n_prefix < 1:
            raise ValueError('inference needs at least one prefix token')
        total = n_prefix + max_len
        sess = self.session(torch.zeros((1, self.d), device=self.device), None, 1, total)
        rows = _identity_rows(1, total, self.device)
        am = torch.zeros(1, dtype=torch.int32, device=self.device)
        out = []
        for p in range(total - 1):
            rows.tok.copy_(src[p:p + 1] if p < n_prefix else am)
            logits = sess.step(rows, 1, p)
            if p >= n_prefix - 1:
                K.argmax_rows(logits, 1, self.V, am)
                tok = int(am.item())
                if tok == eos or len(out) >= max_len:
                    break
                out.append(tok)
        return torch.tensor(out, dtype=torch.long, device=self.device)


def _identity_rows(B, T, device):
    rows = K.RowState(B, max(T, 1), device, True, 0)
    ar = torch.arange(B, dtype=torch.int32, device=device)
    rows.sent.copy_(ar)
    rows.anc.copy_(ar.unsqueeze(1).expand(B, max(T, 1)))
    return rows


class TransformerLMSession:
    """Per-batch decode state of a ``TransformerLM``: the conditioning vectors, the self-attention q|k|v cache
    [layer][position][row slot][3d] (addressed through the ancestor table, so beam reorders move nothing)
    and work buffers."""
    needs_anc = True
    incremental = True
    takes_soft = True
    lag_ok = True     # rows beyond the true count may be stepped on stale state without harm (engine.DecodeLoop.lagged)

    def __init__(self, lm, z, k, max_len):
        self.lm, self.V = lm, lm.V
        W = lm.prepared()
        dev = lm.device
        if max_len > W['pe'].shape[0]:
            raise ValueError('max_len %d exceeds pe_max_len %d' % (max_len, W['pe'].shape[0]))
        z = z.to(device=dev, dtype=torch.float32).contiguous()
        if z.dim() != 2 or z.shape[1] != lm.d:
            raise ValueError('z must be [B, model_size=%d], got %s' % (lm.d, tuple(z.shape)))
        self.z = z
        B, d = z.shape
        R = B * k
        self.R = R
        act = torch.bfloat16 if lm.dtype == 'bf16' else torch.float32
        f32 = dict(dtype=torch.float32, device=dev)
        lo = dict(dtype=act, device=dev)
        self.cache = [torch.empty((max_len, R, 3 * d), **lo) for _ in W['layers']]
        self.x = torch.empty((R, d), **f32)
        self.h = torch.empty((R, d), **lo)
        self.a = torch.empty((R, d), **lo)
        self.ff = torch.empty((R, lm.d_ff), **lo)
        self.logits = torch.empty((R, (lm.V + 7) // 8 * 8), **lo)
        self.scale = float(torch.tensor(math.sqrt(d), dtype=torch.float32))
        self._plans = {}
        # beam search on bf16 logits: the vocabulary projection also leaves, per row and slice of 64 columns, the largest logit
        # and the sum of exponentials, and the selection pass reads those instead of the rows (steps of a shape the
        # statistics kernel serves; other steps run the two plain kernels).  ``stats`` = (buffer, inner dimension) or None
        self.stats = None
        if VOCAB_STATS and lm.dtype == 'bf16' and k > 1 and K.vocab_stats_ok(R, lm.V, d):
            self.stats = (torch.empty(((R + 31) // 32, (lm.V + 63) // 64, 64), **f32), d)

    def reorder(self, row_parent, n_rows, identity):
        pass  # the KV cache is addressed through the ancestor table; nothing moves

    def _plan(self, anc_ld):
        """The native launch plan of a step (one per ancestor-table width)."""
        plan = self._plans.get(anc_ld)
        if plan is None:
            lm, W = self.lm, self.lm.prepared()
            layers = [dict(ln_self_g=L['ln1'][0], ln_self_b=L['ln1'][1], wqkv=L['wqkv'], bqkv=L['bqkv'], wo=L['wo'], bo=L['bo'],
                           ln_ff_g=L['ln2'][0], ln_ff_b=L['ln2'][1], w1=L['w1'], b1=L['b1'], w2=L['w2'], b2=L['b2'],
                           cache=self.cache[li]) for li, L in enumerate(W['layers'])]
            plan = K.StepPlan(d=lm.d, heads=lm.heads, d_ff=lm.d_ff, V=self.V, bf16=lm.dtype == 'bf16', n_pos=self.cache[0].shape[0],
                              slab_rows=self.R, S=0, anc_ld=anc_ld, emb=W['emb'], pe=W['pe'], scale=self.scale, z=self.z, mask=None,
                              ln=W['ln'], wout=W['wout'], bout=W['bout'], x=self.x, h=self.h, q=None, a=self.a, ff=self.ff,
                              logits=self.logits, layers=layers, part=None if self.stats is None else self.stats[0])
            self._plans[anc_ld] = plan
        return plan

    def step(self, rows, n, t, soft=None):
        """Logits of the next token; ``soft`` (token distributions [>= n, Vp] fp32) replaces the lookup of ``rows.tok``.
        The launches of the step are issued by one native call (two when the vocabulary projection is being timed)."""
        lm = self.lm
        plan = self._plan(rows.anc.shape[1])
        body = K.STEP_LAYERS
        if soft is None:
            body |= K.STEP_EMBED
        else:
            W = lm.prepared()
            if 'emb_pad' not in W:
                W['emb_pad'] = _pad_rows4(W['emb'])
            K.soft_embed(soft, W['emb_pad'], self.scale, W['pe'][t], self.x, n)
            K.add_rows(self.x, self.z, rows.sent, n)
        log = lm.event_log
        if log is None:
            plan.run(rows, n, t, body | K.STEP_VOCAB)
        else:  # measurement aid: HIP events on the launch stream around the vocabulary projection
            plan.run(rows, n, t, body)
            log.begin()
            plan.run(rows, n, t, K.STEP_VOCAB)
            log.end(2.0 * n * lm.d * self.V)
        return self.logits
