"""GRU decoder / language model whose decode steps run on the HIP kernels.

Call surfaces as used by the reference: ``RNNDecoder(emb_size, hid_size, vocab, dropword_p)`` with
``decoder(z [B,H], x [B,T]) -> [B,T,V]`` (``/root/reference/src/trainers/cyclegan_trainer.py:64``,
``/root/reference/src/inference.py:170-177``) and ``RNNLM(size, vocab, n_layers)`` with
``lm(z [n_layers,B,H], x, return_z=True) -> (logits, z')`` (``/root/reference/src/prod/dialog_model.py:49,108-120``).
Before the loop (``/root/reference/src/trainers/cyclegan_trainer.py:321-332``): ``RNNEncoder(emb, hid, vocab, dropword_p,
noise=...)`` with ``encoder(x [B,T]) -> z [B,H]`` (``:63``) and ``Generator(size, n_steps)`` with ``gen(z) -> z'``
(``:337-348``: a GRU fed n_steps copies of z, started at z).
Architecture (this build's own, the reference's model source is an absent submodule): embedding ->
n_layers GRU (PyTorch gate order and formula) -> linear to raw logits.  The decode loop carries the hidden
state instead of re-running the prefix (``/root/reference/src/inference.py:132`` re-runs it).
"""
import math
import os
from collections import OrderedDict

import torch

from . import kernels as K
from .transformer import HipModule, _pad_rows4, _t


class _GRUModel(HipModule):
    def __init__(self, emb_size, hid_size, vocab, n_layers, device):
        super().__init__(device)
        self.vocab = vocab
        self.E, self.H, self.V, self.n_layers = emb_size, hid_size, len(vocab), n_layers
        if emb_size % 4 or hid_size % 4:
            raise ValueError('embedding and hidden sizes must be multiples of 4')
        s = 1.0 / math.sqrt(hid_size)
        self._add('embed.weight', (self.V, emb_size), std=1.0)
        for l in range(n_layers):
            inp = emb_size if l == 0 else hid_size
            self._add('gru.weight_ih_l%d' % l, (3 * hid_size, inp), std=s)
            self._add('gru.weight_hh_l%d' % l, (3 * hid_size, hid_size), std=s)
            self._add('gru.bias_ih_l%d' % l, (3 * hid_size,), std=s)
            self._add('gru.bias_hh_l%d' % l, (3 * hid_size,), std=s)
        self._add('out.weight', (self.V, hid_size), std=s)
        self._add('out.bias', (self.V,), fill=0.0)

    def prepared(self):
        if self._prepared is None:
            P, W = self._params, {}
            W['emb'] = P['embed.weight'].contiguous()
            W['cells'] = [dict(wih=_t(P['gru.weight_ih_l%d' % l]), bih=P['gru.bias_ih_l%d' % l],
                               whh=_t(P['gru.weight_hh_l%d' % l]), bhh=P['gru.bias_hh_l%d' % l])
                          for l in range(self.n_layers)]
            W['wout'], W['bout'] = _t(P['out.weight']), P['out.bias']
            self._prepared = W
        return self._prepared

    def soft_embedding_matrix(self):
        W = self.prepared()
        if 'emb_pad' not in W:
            W['emb_pad'] = _pad_rows4(W['emb'])
        return W['emb_pad']

    def gi_table(self):
        """[V, 3H]: layer 0's input pre-activations (bias_ih + embedding @ W_ih) of every vocabulary entry, made once per
        set of weights, or None when it would not pay (see ``GI_TABLE_MAX_BYTES``).  A decode step whose inputs are token ids
        then looks the rows up instead of multiplying the embeddings: same values (the same product kernel made them)."""
        W = self.prepared()
        if 'gi_tab' not in W:
            nbytes = self.V * 3 * self.H * 4
            W['gi_tab'] = (K.gi_table(W['emb'], W['cells'][0]['wih'], W['cells'][0]['bih'])
                           if GI_TABLE_MAX_BYTES > 0 and nbytes <= GI_TABLE_MAX_BYTES else None)
        return W['gi_tab']

    def session(self, z, enc_mask=None, rows_per_sentence=1, max_len=None):
        return GRUSession(self, z, rows_per_sentence)

    # The two sub-modules that callers of the reference use directly (``decoder.embed(ids)``, ``decoder.gru(embs, z)[1]``
    # at /root/reference/src/prod/style_model.py:64,68; ``lm.embed(text)``, ``lm.gru(embs)[1]`` at
    # /root/reference/src/prod/dialog_model.py:105-106), with the calling conventions of nn.Embedding and of a
    # batch_first nn.GRU.
    def embed(self, ids):
        """ids [B, T] -> embeddings [B, T, E]"""
        W = self.prepared()
        B, T = ids.shape
        out = torch.empty((B * T, self.E), dtype=torch.float32, device=self.device)
        K.embed_posadd(ids.to(device=self.device, dtype=torch.int32).contiguous().view(-1), W['emb'], None, 1.0, out, B * T)
        return out.view(B, T, self.E)

    def gru(self, embs, h0=None):
        """embs [B, T, E], h0 [n_layers, B, H] (zeros if omitted) -> (outputs of the last layer [B, T, H], final states
        [n_layers, B, H]).  Layer by layer: the input products of all positions of a layer are one GEMM."""
        W = self.prepared()
        B, T, _ = embs.shape
        f32 = dict(dtype=torch.float32, device=self.device)
        x = embs.to(**f32).transpose(0, 1).contiguous().view(T * B, -1)       # position-major rows t * B + b
        h_n = []
        gh = torch.empty((B, 3 * self.H), **f32)
        for l, c in enumerate(W['cells']):
            gi = torch.empty((T * B, 3 * self.H), **f32)
            K.gemm(x, c['wih'], c['bih'], gi, T * B)
            h = torch.zeros((B, self.H), **f32) if h0 is None else h0[l].to(**f32).contiguous().clone()
            out = torch.empty((T * B, self.H), **f32)
            for t in range(T):
                K.gemm(h, c['whh'], c['bhh'], gh, B)
                h_new = out[t * B:(t + 1) * B]
                K.gru_cell_step(gi[t * B:(t + 1) * B], gh, h, h_new, B)
                h = h_new
            h_n.append(h.clone())
            x = out
        return x.view(T, B, self.H).transpose(0, 1).contiguous(), torch.stack(h_n)

    def _run_prefix(self, z, x, extra=None):
        """Teacher-forced evaluation: logits [B,T,V] and the final state [n_layers,B,H]."""
        B, T = x.shape
        sess = GRUSession(self, z, 1, extra=extra)
        rows = K.RowState(B, max(T, 1), self.device, False, 0)
        rows.sent.copy_(torch.arange(B, dtype=torch.int32, device=self.device))
        out = torch.empty((B, T, self.V), dtype=torch.float32, device=self.device)
        x32 = x.to(device=self.device, dtype=torch.int32)
        for p in range(T):
            rows.tok.copy_(x32[:, p])
            out[:, p] = sess.step(rows, B, p)[:B, :self.V]
        return out, torch.stack([h[:B].clone() for h in sess.h])


class RNNDecoder(_GRUModel):
    def __init__(self, emb_size, hid_size, vocab, dropword_p=0.0, device='cuda'):
        super().__init__(emb_size, hid_size, vocab, 1, device)

    def __call__(self, z, x):
        return self._run_prefix(z, x)[0]


class RNNLM(_GRUModel):
    def __init__(self, size, vocab, n_layers=1, device='cuda'):
        super().__init__(size, size, vocab, n_layers, device)

    def __call__(self, z, x, return_z=False):
        logits, z_new = self._run_prefix(z, x)
        return (logits, z_new) if return_z else logits


class ConditionalLM(_GRUModel):
    """``ConditionalLM(size, vocab)`` as called at ``/root/reference/src/prod/dialog_model.py:52,99-100``:
    ``lm(z [2, B, size], x, style=s, return_z=True) -> (logits, z')``; under ``InferenceState`` the style comes in through
    ``kwargs`` (``:119``).  The reference's model source is absent; this build's choice: a GRU language model (two
    layers, as the caller's z has) whose token embeddings get a learned style embedding added.  ``style`` is one
    index for the whole batch or one per sentence."""
    session_takes_kwargs = True

    def __init__(self, size, vocab, n_layers=2, n_styles=2, device='cuda'):
        super().__init__(size, size, vocab, n_layers, device)
        self._add('style_embed.weight', (n_styles, size), std=1.0)
        order = ['embed.weight', 'style_embed.weight'] + [k for k in self._params if k not in ('embed.weight', 'style_embed.weight')]
        self._params = OrderedDict((k, self._params[k]) for k in order)

    def _style_rows(self, style, batch):
        styles = torch.as_tensor(0 if style is None else style, dtype=torch.long, device=self.device).reshape(-1)
        if styles.numel() not in (1, batch):
            raise ValueError('style must be one index or one per sentence (%d), got %d' % (batch, styles.numel()))
        return self._params['style_embed.weight'][styles.expand(batch)].contiguous()

    def session(self, z, enc_mask=None, rows_per_sentence=1, max_len=None, style=None, **ignored):
        batch = z.shape[1] if z.dim() == 3 else z.shape[0]
        return GRUSession(self, z, rows_per_sentence, extra=self._style_rows(style, batch))

    def __call__(self, z, x, style=None, return_z=False):
        logits, z_new = self._run_prefix(z, x, extra=self._style_rows(style, x.shape[0]))
        return (logits, z_new) if return_z else logits


# Largest table of per-token input pre-activations a model may keep (bytes; V * 3H floats: 1.5 MB for a character vocabulary,
# 393 MB for 32000 tokens at H = 1024); TST_NO_GI_TABLE=1 in the environment turns the tables off (measurements; same results)
GI_TABLE_MAX_BYTES = 0 if os.environ.get('TST_NO_GI_TABLE') else 2 << 30

# rows from which the one-kernel GRU cell is used: below, 128 x 64 tiles leave most compute units without work and the
# small-problem GEMM kernels of the three-call form are faster
FUSED_MIN_ROWS = int(os.environ.get('TST_GRU_FUSED_MIN_ROWS', '2048'))   # measured crossover at E = H = 1024: between 1024 and 2048 rows
# the same when layer 0 reads its input pre-activations from the per-token table (half the work in either form): the two forms
# are within 2 % of each other from 1 k rows on, the three-call form 3-5 % ahead between 2 k and 4 k rows (measured)
FUSED_MIN_ROWS_TAB = int(os.environ.get('TST_GRU_FUSED_MIN_ROWS', '4096'))


class GRUSession:
    """Carried hidden state [layer][row][H] (two buffers per layer: rows are re-gathered when the batch
    is compacted or beams are reordered) and work buffers.  ``extra`` [B, E]: a vector per sentence that is added
    to the embedding of each of its tokens."""
    needs_anc = False
    takes_soft = True
    incremental = True
    lag_ok = True     # rows beyond the true count may be stepped on stale state without harm (engine.DecodeLoop.lagged)

    def __init__(self, model, z, k, extra=None):
        self.m, self.V = model, model.V
        self.extra = extra
        dev = model.device
        z = z.to(device=dev, dtype=torch.float32)
        if z.dim() == 2:
            z = z.unsqueeze(0)
        if z.shape[0] != model.n_layers or z.shape[2] != model.H:
            raise ValueError('state must be [n_layers=%d, B, H=%d], got %s' % (model.n_layers, model.H, tuple(z.shape)))
        B = z.shape[1]
        R = B * k
        self.R = R
        f32 = dict(dtype=torch.float32, device=dev)
        self.h = [torch.empty((R, model.H), **f32) for _ in range(model.n_layers)]
        for l in range(model.n_layers):
            self.h[l][:B] = z[l]        # step 0: one row per sentence; the first reorder fans out to k rows
        self.h_alt = [torch.empty((R, model.H), **f32) for _ in range(model.n_layers)]
        self.x = torch.empty((R, model.E), **f32)
        self.gi = torch.empty((R, 3 * model.H), **f32)
        self.gh = torch.empty((R, 3 * model.H), **f32)
        self.logits = torch.empty((R, (model.V + 3) // 4 * 4), **f32)
        # wide vocabulary, many rows: the top cell also writes its state k-major and the output projection stages both
        # operands by LDS-DMA (tst_gru_step.ht)
        self.ht = None
        if R >= K.DMA_MIN_ROWS and model.V >= 2048 and model.H % 16 == 0 and K.gru_cell_fused_ok(model.E, model.H):
            self.ht = torch.empty((model.H, (R + 127) // 128 * 128), **f32)
        self.plan = None

    def _plan(self):
        if self.plan is None:
            m, W = self.m, self.m.prepared()
            # token inputs without a per-sentence vector added to them: layer 0 looks its input pre-activations up per token
            gi_tab = m.gi_table() if self.extra is None else None
            self.plan = K.GruPlan(E=m.E, H=m.H, V=m.V, rows=self.R, emb=W['emb'], extra=self.extra, cells=W['cells'], h=self.h,
                                  h_alt=self.h_alt, wout=W['wout'], bout=W['bout'], x=self.x, gi=self.gi, gh=self.gh, logits=self.logits,
                                  ht=self.ht, gi_tab=gi_tab)
        return self.plan

    def reorder(self, row_parent, n_rows, identity):
        if identity:
            return
        self._plan().reorder(row_parent, n_rows)      # one native call gathers every layer's state
        self.h, self.h_alt = self.h_alt, self.h

    def step(self, rows, n, t, soft=None):
        """One native call issues the launches of the step (embedding, cells, output projection)."""
        phases = K.STEP_LAYERS | K.STEP_VOCAB
        if soft is None:
            phases |= K.STEP_EMBED
        else:   # token distributions instead of ids: expected embedding
            K.soft_embed(soft, self.m.soft_embedding_matrix(), 1.0, None, self.x, n)
        # from FUSED_MIN_ROWS rows on every layer's cell is one kernel (gate pre-activations stay in the matrix-core accumulators);
        # it writes the new state to the other buffer of the pair
        plan = self._plan()
        tab = plan.c.gi_tab is not None and soft is None
        fused = n >= (FUSED_MIN_ROWS_TAB if tab else FUSED_MIN_ROWS) and K.gru_cell_fused_ok(self.m.E, self.m.H)
        if fused:
            phases |= K.STEP_FUSED_CELL
        plan.run(rows, n, phases)
        if fused:
            self.h, self.h_alt = self.h_alt, self.h
        return self.logits


class CharLMFromEmbs:
    """``CharLMFromEmbs(rnn_lm, style_embed, n_layers)`` of ``/root/reference/src/prod/throwaway_models.py:38-55``: a GRU
    language model whose initial state is a learned style embedding.  ``rnn_lm`` is an ``RNNLM`` of this package,
    ``style_embed`` the embedding matrix [n_styles, H] (a tensor, or anything with a ``.weight``)."""

    def __init__(self, rnn_lm, style_embed, n_layers):
        self.n_layers, self.lm = n_layers, rnn_lm
        w = style_embed.weight if hasattr(style_embed, 'weight') else style_embed
        self.style_embed = w.detach().to(device=rnn_lm.device, dtype=torch.float32).contiguous()
        self.device, self.V = rnn_lm.device, rnn_lm.V

    def __call__(self, z, x, return_z=False):
        return self.lm(z, x, return_z=return_z)

    def init_z(self, batch_size, style):
        """[n_layers, batch_size, H]: the embedding of ``style`` for every layer and sentence."""
        return self.style_embed[int(style)].expand(self.n_layers, batch_size, -1).contiguous()

    def session(self, z, enc_mask=None, rows_per_sentence=1, max_len=None):
        return self.lm.session(z, enc_mask, rows_per_sentence, max_len)


class WeightedLMEnsemble:
    """``WeightedLMEnsemble(models, weights)`` of ``/root/reference/src/prod/throwaway_models.py:9-36``: language models
    that are stepped side by side; the prediction is log(sum_i w_i * softmax(logits_i)).  ``zs`` stacks the members'
    states along a new first axis ([n_models, n_layers, B, H], batch axis 2, as at
    ``/root/reference/src/prod/dialog_model.py:103,117``).  Members are models of this package with a ``session``."""

    def __init__(self, models, weights):
        assert len(models) == len(weights)
        assert sum(weights) == 1
        self.models, self.weights = list(models), list(weights)
        self.device = self.models[0].device
        self.V = self.models[0].V
        if any(m.V != self.V for m in self.models):
            raise ValueError('ensemble members must share one vocabulary')
        self._w = torch.tensor(self.weights, dtype=torch.float32, device=self.device)

    def session(self, zs, enc_mask=None, rows_per_sentence=1, max_len=None):
        return EnsembleSession(self, zs, rows_per_sentence, max_len)

    def __call__(self, zs, x, return_z=False):
        """Teacher-forced evaluation: log-mixture [B, T, V] (and the stacked final states with ``return_z``)."""
        results = [m(z, x, True) for m, z in zip(self.models, zs)]
        B, T = x.shape
        stack = torch.stack([r[0].reshape(B * T, self.V) for r in results]).contiguous()
        out = torch.empty((B * T, self.V), dtype=torch.float32, device=self.device)
        K.mix_logprobs(stack, self._w, B * T, self.V, out)
        out = out.view(B, T, self.V)
        return (out, torch.stack([r[1] for r in results])) if return_z else out


class EnsembleSession:
    needs_anc = False
    incremental = True
    lag_ok = True

    def __init__(self, ens, zs, k, max_len):
        self.ens, self.V = ens, ens.V
        if zs.shape[0] != len(ens.models):
            raise ValueError('zs must stack one state per member along its first axis')
        self.members = [m.session(z, None, k, max_len) for m, z in zip(ens.models, zs)]
        R = self.members[0].R
        ldv = (self.V + 3) // 4 * 4
        # the members write their logits into one stacked buffer, which the mixing kernel reads
        self.stack = torch.empty((len(self.members), R, ldv), dtype=torch.float32, device=ens.device)
        for i, s in enumerate(self.members):
            s.logits = self.stack[i]
        self.logits = torch.empty((R, ldv), dtype=torch.float32, device=ens.device)

    def reorder(self, row_parent, n_rows, identity):
        for s in self.members:
            s.reorder(row_parent, n_rows, identity)

    def step(self, rows, n, t):
        for s in self.members:
            s.step(rows, n, t)
        K.mix_logprobs(self.stack, self.ens._w, n, self.V, self.logits)
        return self.logits


class RNNEncoder(HipModule):
    """``RNNEncoder(emb_size, hid_size, vocab, dropword_p, noise=...)``; ``encoder(x [B, T] ids) -> z [B, H]``, the
    final state of a one-layer GRU started at zero and run over all T positions (padding included, nothing is
    packed).  Inference only: ``dropword_p`` and ``noise`` act in training and are ignored.  The input pre-activations
    of a position are rows of the per-token table (``bias_ih + embedding @ W_ih``, made once per set of weights; see
    ``_GRUModel.gi_table``) looked up by the gate update kernel; without the table (``TST_NO_GI_TABLE``, or a table beyond
    ``GI_TABLE_MAX_BYTES``) the input products of all positions are one GEMM ([B * T, E] x [E, 3H]).  The loop over
    positions does the recurrent product and the gate update.  Same values either way."""

    def __init__(self, emb_size, hid_size, vocab, dropword_p=0.0, noise=0.0, device='cuda'):
        super().__init__(device)
        if emb_size % 4 or hid_size % 4:
            raise ValueError('embedding and hidden sizes must be multiples of 4')
        self.vocab, self.E, self.H, self.V = vocab, emb_size, hid_size, len(vocab)
        s = 1.0 / math.sqrt(hid_size)
        self._add('embed.weight', (self.V, emb_size), std=1.0)
        self._add('gru.weight_ih_l0', (3 * hid_size, emb_size), std=s)
        self._add('gru.weight_hh_l0', (3 * hid_size, hid_size), std=s)
        self._add('gru.bias_ih_l0', (3 * hid_size,), std=s)
        self._add('gru.bias_hh_l0', (3 * hid_size,), std=s)

    def prepared(self):
        if self._prepared is None:
            P = self._params
            self._prepared = dict(emb=P['embed.weight'].contiguous(), wih=_t(P['gru.weight_ih_l0']), bih=P['gru.bias_ih_l0'],
                                  whh=_t(P['gru.weight_hh_l0']), bhh=P['gru.bias_hh_l0'])
        return self._prepared

    def __call__(self, x):
        W = self.prepared()
        B, T = x.shape
        dev = self.device
        f32 = dict(dtype=torch.float32, device=dev)
        tok = x.to(device=dev, dtype=torch.int32).t().contiguous().view(-1)      # position-major: rows t * B + b
        if 'gi_tab' not in W:
            nbytes = self.V * 3 * self.H * 4
            W['gi_tab'] = K.gi_table(W['emb'], W['wih'], W['bih']) if 0 < nbytes <= GI_TABLE_MAX_BYTES else None
        tab = W['gi_tab']
        if tab is None:
            emb = torch.empty((T * B, self.E), **f32)
            K.embed_posadd(tok, W['emb'], None, 1.0, emb, T * B)
            gi = torch.empty((T * B, 3 * self.H), **f32)
            K.gemm(emb, W['wih'], W['bih'], gi, T * B)
        h = torch.zeros((B, self.H), **f32)
        gh = torch.empty((B, 3 * self.H), **f32)
        for t in range(T):
            K.gemm(h, W['whh'], W['bhh'], gh, B)
            if tab is None:
                K.gru_cell_step(gi[t * B:(t + 1) * B], gh, h, h, B)
            else:
                K.gru_cell_step_tab(tab, tok[t * B:(t + 1) * B], gh, h, h, B)
        return h


class Generator(HipModule):
    """``Generator(size, n_steps)``; ``gen(z [B, size]) -> [B, size]``: a one-layer GRU whose input is z at every one
    of the n_steps steps and whose initial state is z (``/root/reference/src/trainers/cyclegan_trainer.py:337-348``).
    The input product is therefore the same at every step and is computed once."""

    def __init__(self, size, n_steps, device='cuda'):
        super().__init__(device)
        if size % 4:
            raise ValueError('size must be a multiple of 4')
        self.size, self.n_steps = size, n_steps
        s = 1.0 / math.sqrt(size)
        self._add('gru.weight_ih_l0', (3 * size, size), std=s)
        self._add('gru.weight_hh_l0', (3 * size, size), std=s)
        self._add('gru.bias_ih_l0', (3 * size,), std=s)
        self._add('gru.bias_hh_l0', (3 * size,), std=s)

    def prepared(self):
        if self._prepared is None:
            P = self._params
            self._prepared = dict(wih=_t(P['gru.weight_ih_l0']), bih=P['gru.bias_ih_l0'], whh=_t(P['gru.weight_hh_l0']),
                                  bhh=P['gru.bias_hh_l0'])
        return self._prepared

    def __call__(self, z):
        W = self.prepared()
        z = z.to(device=self.device, dtype=torch.float32).contiguous()
        B = z.shape[0]
        f32 = dict(dtype=torch.float32, device=self.device)
        gi = torch.empty((B, 3 * self.size), **f32)
        gh = torch.empty((B, 3 * self.size), **f32)
        K.gemm(z, W['wih'], W['bih'], gi, B)
        h = z.clone()
        for _ in range(self.n_steps):
            K.gemm(h, W['whh'], W['bhh'], gh, B)
            K.gru_cell_step(gi, gh, h, h, B)
        return h


class FFN(HipModule):
    """``FFN(sizes, dropout=0)``: the stack of linear layers the reference's callers build as ``FFN([in, ..., out])``
    (``merge_z`` of ``/root/reference/src/prod/style_model.py:40,67``, applied to ``cat([z, morphs], dim=1)`` before the
    decode loop).  The reference's model source is absent; this build's choice is ReLU between layers and none after
    the last, so the one-layer form the pre-decode path uses is a plain linear map.  Inference only (``dropout`` is
    ignored).  Input widths that are not a multiple of 4 are zero-padded (input and weight alike), which leaves every
    output's sum of products unchanged."""

    def __init__(self, sizes, dropout=0.0, device='cuda'):
        super().__init__(device)
        sizes = list(sizes)
        if len(sizes) < 2:
            raise ValueError('FFN needs at least an input and an output size')
        self.sizes = sizes
        for i, (n_in, n_out) in enumerate(zip(sizes[:-1], sizes[1:])):
            self._add('layers.%d.weight' % i, (n_out, n_in), std=1.0 / math.sqrt(n_in))
            self._add('layers.%d.bias' % i, (n_out,), fill=0.0)

    def prepared(self):
        if self._prepared is None:
            P, layers = self._params, []
            for i in range(len(self.sizes) - 1):
                wt = _t(P['layers.%d.weight' % i])                 # [in, out]
                k_pad = (wt.shape[0] + 3) // 4 * 4
                if k_pad != wt.shape[0]:
                    full = torch.zeros((k_pad, wt.stride(0)), dtype=torch.float32, device=wt.device)
                    full[:wt.shape[0], :wt.shape[1]] = wt
                    wt = full[:, :wt.shape[1]]
                layers.append((wt, P['layers.%d.bias' % i]))
            self._prepared = layers
        return self._prepared

    def __call__(self, x):
        x = x.to(device=self.device, dtype=torch.float32)
        lead, B = x.shape[:-1], x.numel() // x.shape[-1]
        h = x.reshape(B, x.shape[-1])
        layers = self.prepared()
        for i, (wt, b) in enumerate(layers):
            k_pad = wt.shape[0]
            if h.shape[1] != k_pad or not h.is_contiguous():
                padded = torch.zeros((B, k_pad), dtype=torch.float32, device=self.device)
                padded[:, :h.shape[1]] = h
                h = padded
            n_out = wt.shape[1]
            out = torch.empty((B, (n_out + 3) // 4 * 4), dtype=torch.float32, device=self.device)
            K.gemm(h, wt, b, out, B, n_cols=n_out, act=0 if i == len(layers) - 1 else 1)
            h = out[:, :n_out]
        return h.reshape(*lead, h.shape[1]) if h.is_contiguous() else h.contiguous().reshape(*lead, h.shape[1])
