"""Transformer encoder on the HIP kernels: the pass that produces the decode loop's ``inputs``
(``encs, enc_mask = transformer.encoder(src)``, ``/root/reference/src/trainers/transformer_mt_trainer.py:91``).

Pre-norm layers as in the decoder.  Self-attention over a sentence reuses the one-query attention kernel:
every non-padding (sentence, position) is a row whose keys/values are the non-padding positions of its own
sentence.
"""
import math

import torch

from . import kernels as K
from .transformer import HP, HipModule, _t, sinusoid_table, tensor_version


class TransformerEncoder(HipModule):
    def __init__(self, hp, vocab, device='cuda'):
        super().__init__(device)
        hp = HP(hp)
        self.hp, self.vocab = hp, vocab
        d, V, ff = hp.d_model, len(vocab), hp.d_ff
        self.d, self.V, self.heads, self.n_layers = d, V, hp.n_heads, hp.n_layers
        self.pad_idx = vocab.stoi['<pad>']
        s = 1.0 / math.sqrt(d)
        self._add('embed.weight', (V, d), std=s)
        for i in range(hp.n_layers):
            p = 'layers.%d.' % i
            self._add(p + 'ln1.weight', (d,), fill=1.0)
            self._add(p + 'ln1.bias', (d,), fill=0.0)
            for n in 'qkvo':
                self._add(p + 'attn.' + n + '.weight', (d, d), std=s)
                self._add(p + 'attn.' + n + '.bias', (d,), fill=0.0)
            self._add(p + 'ln2.weight', (d,), fill=1.0)
            self._add(p + 'ln2.bias', (d,), fill=0.0)
            self._add(p + 'ff.w1.weight', (ff, d), std=s)
            self._add(p + 'ff.w1.bias', (ff,), fill=0.0)
            self._add(p + 'ff.w2.weight', (d, ff), std=1.0 / math.sqrt(ff))
            self._add(p + 'ff.w2.bias', (d,), fill=0.0)
        self._add('ln.weight', (d,), fill=1.0)
        self._add('ln.bias', (d,), fill=0.0)
        with torch.inference_mode(False):
            self._params['pe'] = sinusoid_table(hp.pe_max_len, d).to(self.device)

    def prepared(self):
        if self._prepared is None:
            P, W = self._params, {}
            W['emb'], W['pe'] = P['embed.weight'].contiguous(), P['pe'].contiguous()
            W['layers'] = []
            for i in range(self.n_layers):
                p = 'layers.%d.' % i
                a = p + 'attn.'
                W['layers'].append(dict(
                    ln1=(P[p + 'ln1.weight'], P[p + 'ln1.bias']), ln2=(P[p + 'ln2.weight'], P[p + 'ln2.bias']),
                    wq=_t(P[a + 'q.weight']), bq=P[a + 'q.bias'],
                    wkv=_t(torch.cat([P[a + 'k.weight'], P[a + 'v.weight']], 0)),
                    bkv=torch.cat([P[a + 'k.bias'], P[a + 'v.bias']]).contiguous(),
                    wo=_t(P[a + 'o.weight']), bo=P[a + 'o.bias'],
                    w1=_t(P[p + 'ff.w1.weight']), b1=P[p + 'ff.w1.bias'],
                    w2=_t(P[p + 'ff.w2.weight']), b2=P[p + 'ff.w2.bias']))
            W['ln'] = (P['ln.weight'], P['ln.bias'])
            self._prepared = W
        return self._prepared

    def __call__(self, src):
        """src [B,S] token ids -> (encs [B,S,d] fp32, enc_mask [B,1,S] bool; True where src is not padding).
        Only the non-padding positions are computed (rows packed sentence after sentence; a batch padded to its
        longest sentence is mostly padding); ``encs`` is zero at padding positions, which nothing downstream reads."""
        W, dev, d = self.prepared(), self.device, self.d
        src = src.to(dev)
        B, S = src.shape
        if S > W['pe'].shape[0]:
            raise ValueError('source length %d exceeds pe_max_len %d' % (S, W['pe'].shape[0]))
        f32 = dict(dtype=torch.float32, device=dev)
        packed = K.PackedSources(src.to(torch.int64).contiguous(), self.pad_idx)    # one kernel sequence, one host read
        n = packed.n
        enc_mask = packed.mask.view(B, 1, S).to(torch.bool)
        encs = torch.zeros((B * S, d), **f32)
        if n == 0:
            return encs.view(B, S, d), enc_mask
        mask8, kvrow, tok, pos, sent = packed.mask, packed.kvrow, packed.tok, packed.pos, packed.sent
        x = torch.empty((n, d), **f32)
        h, q, a = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        kv = torch.empty((n, 2 * d), **f32)
        ff = torch.empty((n, self.hp.d_ff), **f32)
        scale = float(torch.tensor(math.sqrt(d), dtype=torch.float32))
        K.embed_posadd(tok, W['emb'], W['pe'], scale, x, n, pos=pos)
        for L in W['layers']:
            K.layernorm(x, L['ln1'][0], L['ln1'][1], h, n)
            K.gemm(h, L['wq'], L['bq'], q, n)
            K.gemm(h, L['wkv'], L['bkv'], kv, n)
            K.attn_cross_step(q, kv, mask8, sent, a, n, d, self.heads, kvrow=kvrow)
            K.gemm(a, L['wo'], L['bo'], x, n, res=x)
            K.layernorm(x, L['ln2'][0], L['ln2'][1], h, n)
            K.gemm(h, L['w1'], L['b1'], ff, n, act=1)
            K.gemm(ff, L['w2'], L['b2'], x, n, res=x)
        K.layernorm(x, W['ln'][0], W['ln'][1], h, n)
        K.scatter_rows(h, packed.padded_row, encs, n)
        encs = encs.view(B, S, d)
        # the decoder session of this package can take the packed rows as they are (no second packing pass) when it is handed
        # these two tensor objects unchanged: same objects AND same contents (the version counters recorded here move with
        # every in-place edit, e.g. ``encs.add_(style)`` or ``enc_mask[..., j] = False``, after which the session repacks
        # from the tensors it is given)
        packed.rows = h
        packed.versions = (tensor_version(encs), tensor_version(enc_mask))    # None under torch.inference_mode(): the session repacks
        encs._tst_packed = enc_mask._tst_packed = packed
        return encs, enc_mask
