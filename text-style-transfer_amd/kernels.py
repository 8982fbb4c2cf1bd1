"""Thin wrappers that launch the HIP kernels of ``libtst_hip.so`` on torch device tensors.

torch is used for device memory and the current stream only; every function below is one C-ABI call.
Tensors must be contiguous fp32 / bf16 / int32 on the GPU; outputs are written into caller-provided tensors.
Where a kernel exists for fp32 and bf16 storage the wrapper picks the entry point by the tensors' dtype.
"""
import torch

from . import _lib
from ._lib import call, STEP_EMBED, STEP_LAYERS, STEP_VOCAB, STEP_FUSED_CELL  # noqa: F401


DMA_MIN_ROWS = 2048   # TST_DMA_MIN_ROWS of include/tst_hip.h
WREG_MIN_ROWS = 1024  # TST_WREG_MIN_ROWS

_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream():
    """hipStream_t of torch's current stream on the current device (the raw-handle query is ~10x cheaper than building a
    torch.cuda.Stream object, and this runs once per native call)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def _chk(t, dtype, name):
    if t is None:
        return
    if not (t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise ValueError('%s must be a contiguous %s tensor on the GPU (got %s, %s, cuda=%s)'
                         % (name, dtype, t.dtype, tuple(t.shape), t.is_cuda))


def embed_posadd(tok, emb, pe, scale, out, n_rows, pos=None):
    """out[:n_rows] = emb[tok[:n_rows]] * scale + (pe[pos[r]] if pos is given else the single row pe)"""
    _chk(tok, torch.int32, 'tok'); _chk(emb, torch.float32, 'emb'); _chk(pe, torch.float32, 'pe')
    _chk(out, torch.float32, 'out'); _chk(pos, torch.int32, 'pos')
    V, d = emb.shape
    assert out.shape[1] == d and out.shape[0] >= n_rows and tok.shape[0] >= n_rows
    call('tst_embed_posadd_f32', _p(tok), _p(emb), _p(pe), _p(pos), float(scale), _p(out), n_rows, d, V, _stream())


def embed_posadd_rows(tok, emb, pe, scale, z, zidx, out, n_rows, pos=None):
    """out[:n_rows] = emb[tok] * scale + pe + z[zidx]: ``embed_posadd`` and ``add_rows`` in one pass"""
    _chk(tok, torch.int32, 'tok'); _chk(emb, torch.float32, 'emb'); _chk(pe, torch.float32, 'pe'); _chk(z, torch.float32, 'z')
    _chk(out, torch.float32, 'out'); _chk(pos, torch.int32, 'pos'); _chk(zidx, torch.int32, 'zidx')
    V, d = emb.shape
    assert out.shape[1] == d and z.shape[1] == d and min(out.shape[0], tok.shape[0], zidx.shape[0]) >= n_rows
    call('tst_embed_posadd_rows_f32', _p(tok), _p(emb), _p(pe), _p(pos), float(scale), _p(z), _p(zidx), _p(out), n_rows, d, V, _stream())


def soft_embed(dist, emb_pad, scale, pe, out, n_rows):
    """out[:n_rows] = (dist[:n_rows] @ emb_pad) * scale + pe: the embedding of token distributions.  ``dist`` is
    [rows, Vp] with Vp = V rounded up to a multiple of 4 (zero columns beyond V), ``emb_pad`` [Vp, d] (zero rows beyond V)."""
    gemm(dist, emb_pad, None, out, n_rows)
    if scale != 1.0 or pe is not None:
        _chk(pe, torch.float32, 'pe')
        call('tst_scale_posadd_f32', _p(out), float(scale), _p(pe), n_rows, out.shape[1], _stream())


def gumbel_softmax(logits, n_rows, n_cols, u, temperature, out):
    """out[:n_rows, :n_cols] = softmax((softmax(logits) + gumbel(u)) / temperature); 2-D fp32 views, unit inner stride"""
    for t, nm in ((logits, 'logits'), (u, 'u'), (out, 'out')):
        if not (t.is_cuda and t.dtype == torch.float32 and t.stride(-1) == 1):
            raise ValueError('%s must be an fp32 GPU tensor with unit inner stride' % nm)
    assert min(logits.shape[0], u.shape[0], out.shape[0]) >= n_rows and min(logits.shape[1], u.shape[1], out.shape[1]) >= n_cols
    call('tst_gumbel_softmax_f32', _p(logits), logits.stride(0), n_rows, n_cols, _p(u), u.stride(0), float(temperature),
         _p(out), out.stride(0), _stream())


def mix_logprobs(stack, weights, n_rows, n_cols, out):
    """out[:n_rows, :n_cols] = log(sum_i weights[i] * softmax(stack[i, :n_rows, :n_cols])); ``stack`` [n_models, rows, ld]"""
    _chk(stack, torch.float32, 'stack'); _chk(weights, torch.float32, 'weights')
    if not (out.is_cuda and out.dtype == torch.float32 and out.stride(-1) == 1):
        raise ValueError('out must be an fp32 GPU tensor with unit inner stride')
    assert stack.dim() == 3 and weights.numel() == stack.shape[0] and stack.shape[1] >= n_rows and stack.shape[2] >= n_cols
    call('tst_mix_logprobs_f32', _p(stack), stack.stride(0), stack.stride(1), stack.shape[0], _p(weights), n_rows, n_cols,
         _p(out), out.stride(0), _stream())


def layernorm(x, gamma, beta, out, n_rows, eps=1e-5):
    """out = LayerNorm(x); ``out`` is fp32, or bf16 when it feeds a bf16 GEMM"""
    _chk(x, torch.float32, 'x'); _chk(gamma, torch.float32, 'gamma'); _chk(beta, torch.float32, 'beta')
    _chk(out, out.dtype if out.dtype == torch.bfloat16 else torch.float32, 'out')
    d = x.shape[1]
    assert gamma.numel() == d and beta.numel() == d and out.shape[1] == d
    name = 'tst_layernorm_f32_bf16' if out.dtype == torch.bfloat16 else 'tst_layernorm_f32'
    call(name, _p(x), _p(gamma), _p(beta), float(eps), _p(out), n_rows, d, _stream())


def ln_gemm(x, gamma, beta, h, wt, bias, out, n_rows, n_cols=None, act=0, eps=1e-5):
    """out[:n_rows, :N] = epilogue(LayerNorm(x[:n_rows]) @ wt[:, :N] + bias); ``h`` [rows, K] is scratch (it receives the
    normalised rows only when the problem is too large for the fused kernel)."""
    for t, nm in ((x, 'x'), (gamma, 'gamma'), (beta, 'beta'), (h, 'h')):
        _chk(t, torch.float32, nm)
    for t, nm in ((wt, 'wt'), (out, 'out'), (bias, 'bias')):
        if t is not None and not (t.is_cuda and t.dtype == torch.float32 and t.stride(-1) == 1):
            raise ValueError('%s must be an fp32 GPU tensor with unit inner stride' % nm)
    K = x.shape[1]
    N = wt.shape[1] if n_cols is None else n_cols
    assert wt.shape[0] == K and h.shape[1] == K and out.shape[1] >= N and min(x.shape[0], h.shape[0], out.shape[0]) >= n_rows
    assert gamma.numel() == K and beta.numel() == K and (bias is None or bias.numel() >= N)
    call('tst_ln_gemm_f32', _p(x), _p(gamma), _p(beta), float(eps), _p(h), _p(wt), wt.stride(0), _p(bias), _p(out), out.stride(0),
         n_rows, N, K, int(act), _stream())


def vocab_argmax_ok(n_rows, n_cols, d):
    """Whether ``ln_vocab_argmax`` (and the keys output of a transformer step) accepts this shape."""
    return bool(_lib._dll.tst_vocab_argmax_ok(n_rows, n_cols, d))


def ln_vocab_argmax(x, gamma, beta, wt, bias, keys, n_rows, n_cols=None, eps=1e-5):
    """keys[:n_rows] (int64 bit patterns [rows, ARGMAX_SLOTS], ZERO on entry) = arg-max keys of LayerNorm(x[:n_rows]) @ wt[:, :N]
    + bias: final LayerNorm, vocabulary projection and arg-max as one kernel; ``tokens_of_keys`` decodes them."""
    for t, nm in ((x, 'x'), (gamma, 'gamma'), (beta, 'beta')):
        _chk(t, torch.float32, nm)
    _chk(keys, torch.int64, 'keys')
    for t, nm in ((wt, 'wt'), (bias, 'bias')):
        if t is not None and not (t.is_cuda and t.dtype == torch.float32 and t.stride(-1) == 1):
            raise ValueError('%s must be an fp32 GPU tensor with unit inner stride' % nm)
    K = x.shape[1]
    N = wt.shape[1] if n_cols is None else n_cols
    assert wt.shape[0] == K and x.shape[0] >= n_rows and keys.numel() >= n_rows * ARGMAX_SLOTS
    assert gamma.numel() == K and beta.numel() == K and (bias is None or bias.numel() >= N)
    call('tst_ln_vocab_argmax_f32', _p(x), _p(gamma), _p(beta), float(eps), _p(wt), wt.stride(0), _p(bias), _p(keys), n_rows, N, K, _stream())


ARGMAX_SLOTS = 16   # TST_ARGMAX_SLOTS of include/tst_hip.h


def tokens_of_keys(keys):
    """Token ids (int32) of arg-max keys [rows, ARGMAX_SLOTS]: the largest key of a row as an UNSIGNED number decides; key 0 =
    no candidate = token 0, else the complement of the lower 32 bits."""
    top = -0x8000000000000000
    best = (keys.view(-1, ARGMAX_SLOTS) ^ top).max(dim=1).values ^ top   # unsigned order on signed storage
    low = best & 0xffffffff
    return torch.where(best == 0, torch.zeros_like(low), 0xffffffff - low).to(torch.int32)


def chain_ok(n_rows, d, k1, n2):
    return bool(_lib._dll.tst_chain_ok(n_rows, d, k1, n2))


def chain(a, w1t, b1, x, gamma, beta, n_rows, w2t=None, b2=None, out=None, n_cols=None, act2=0, out_t=None, eps=1e-5):
    """x[:n_rows] += a[:n_rows] @ w1t + b1; y = LayerNorm(x); then ``out[:n_rows, :N2] = act2(y @ w2t[:, :N2] + b2)``, or,
    without ``w2t``, ``out_t[k, r] = y[r, k]``: one kernel per tile of 64 rows (d_model 256)."""
    for t, nm in ((a, 'a'), (w1t, 'w1t'), (w2t, 'w2t'), (out, 'out'), (b1, 'b1'), (b2, 'b2'), (out_t, 'out_t')):
        if t is not None and not (t.is_cuda and t.dtype == torch.float32 and t.stride(-1) == 1):
            raise ValueError('%s must be an fp32 GPU tensor with unit inner stride' % nm)
    _chk(x, torch.float32, 'x'); _chk(gamma, torch.float32, 'gamma'); _chk(beta, torch.float32, 'beta')
    K1, d = w1t.shape[0], x.shape[1]
    assert a.shape[1] == K1 and w1t.shape[1] >= d and min(a.shape[0], x.shape[0]) >= n_rows and gamma.numel() == d and beta.numel() == d
    N2 = 0
    if w2t is not None:
        N2 = w2t.shape[1] if n_cols is None else n_cols
        assert w2t.shape[0] == d and out.shape[1] >= N2 and out.shape[0] >= n_rows and (b2 is None or b2.numel() >= N2)
    else:
        assert out_t is not None and out_t.shape[0] == d and out_t.shape[1] >= n_rows
    call('tst_chain_f32', _p(a), a.stride(0), K1, _p(w1t), w1t.stride(0), _p(b1), _p(x), _p(gamma), _p(beta), float(eps),
         _p(w2t), 0 if w2t is None else w2t.stride(0), _p(b2), int(act2), _p(out), 0 if out is None else out.stride(0), N2,
         _p(out_t), 0 if out_t is None else out_t.stride(0), n_rows, d, _stream())


def add_rows(x, z, idx, n_rows):
    """x[r] += z[idx[r]]"""
    _chk(x, torch.float32, 'x'); _chk(z, torch.float32, 'z'); _chk(idx, torch.int32, 'idx')
    assert x.shape[1] == z.shape[1] and x.shape[0] >= n_rows and idx.shape[0] >= n_rows
    call('tst_add_rows_f32', _p(x), _p(z), _p(idx), n_rows, x.shape[1], _stream())


def gemm_bf16(a, w, bias, out, n_rows, n_cols=None, res=None, act=0):
    """out[:n_rows, :N] = epilogue(a[:n_rows] @ w[:N].T + bias); ``a`` [rows, K] and ``w`` [N, K] are bf16 (``w`` is an
    nn.Linear weight as stored); ``out`` is bf16 or fp32, ``res`` fp32; 2-D views with unit inner stride."""
    for t, nm, dt in ((a, 'a', torch.bfloat16), (w, 'w', torch.bfloat16), (res, 'res', torch.float32), (bias, 'bias', torch.float32)):
        if t is not None and not (t.is_cuda and t.dtype == dt and t.stride(-1) == 1):
            raise ValueError('%s must be a %s GPU tensor with unit inner stride' % (nm, dt))
    if not (out.is_cuda and out.dtype in (torch.bfloat16, torch.float32) and out.stride(-1) == 1):
        raise ValueError('out must be a bf16 or fp32 GPU tensor with unit inner stride')
    K = a.shape[1]
    N = w.shape[0] if n_cols is None else n_cols
    assert w.shape[1] == K and w.shape[0] >= N and out.shape[1] >= N and a.shape[0] >= n_rows and out.shape[0] >= n_rows
    assert bias is None or bias.numel() >= N
    call('tst_gemm_mfma_bf16', _p(a), a.stride(0), _p(w), w.stride(0), _p(bias), _p(res), 0 if res is None else res.stride(0),
         _p(out), out.stride(0), n_rows, N, K, int(act), int(out.dtype == torch.bfloat16), _stream())


def gemm_res_ln_bf16(a, w, bias, x, gamma, beta, h, n_rows, eps=1e-5):
    """x[:n_rows] += a[:n_rows] @ w.T + bias (fp32 residual stream, 512 wide); h[:n_rows] = LayerNorm(x) in bf16: one kernel"""
    for t, nm, dt in ((a, 'a', torch.bfloat16), (w, 'w', torch.bfloat16), (h, 'h', torch.bfloat16), (bias, 'bias', torch.float32)):
        if t is not None and not (t.is_cuda and t.dtype == dt and t.stride(-1) == 1):
            raise ValueError('%s must be a %s GPU tensor with unit inner stride' % (nm, dt))
    _chk(x, torch.float32, 'x'); _chk(gamma, torch.float32, 'gamma'); _chk(beta, torch.float32, 'beta')
    K, N = a.shape[1], w.shape[0]
    assert w.shape[1] == K and x.shape[1] == N and h.shape[1] >= N and min(a.shape[0], x.shape[0], h.shape[0]) >= n_rows
    call('tst_gemm_res_ln_bf16', _p(a), a.stride(0), _p(w), w.stride(0), _p(bias), _p(x), _p(gamma), _p(beta), float(eps), _p(h),
         h.stride(0), n_rows, N, K, _stream())


def gemm(a, wt, bias, out, n_rows, n_cols=None, res=None, act=0, strip=None):
    """out[:n_rows, :N] = epilogue(a[:n_rows] @ wt[:, :N] + bias); ``wt`` is [K, ldw] with ldw >= N.
    ``a``, ``out`` and ``res`` may be row-strided 2-D views (unit stride in the last dimension)."""
    for t, nm in ((a, 'a'), (wt, 'wt'), (out, 'out'), (res, 'res'), (bias, 'bias')):
        if t is not None and not (t.is_cuda and t.dtype == torch.float32 and t.stride(-1) == 1):
            raise ValueError('%s must be an fp32 GPU tensor with unit inner stride' % nm)
    K, ldw = wt.shape[0], wt.stride(0)
    N = wt.shape[1] if n_cols is None else n_cols
    assert a.shape[1] == K and out.shape[1] >= N and a.shape[0] >= n_rows and out.shape[0] >= n_rows
    assert bias is None or bias.numel() >= N
    if strip is not None:   # the small-problem kernel, whatever the size (tests, tuning)
        call('tst_gemm_skinny_f32', _p(a), a.stride(0), _p(wt), ldw, _p(bias), _p(res), 0 if res is None else res.stride(0),
             _p(out), out.stride(0), n_rows, N, K, int(act), int(strip), _stream())
        return
    call('tst_gemm_mfma_f32', _p(a), a.stride(0), _p(wt), ldw, _p(bias), _p(res), 0 if res is None else res.stride(0),
         _p(out), out.stride(0), n_rows, N, K, int(act), _stream())


def gemm_dma(at, wt, bias, out, n_rows, n_cols=None, variant=0):
    """out[:n_rows, :N] = at[:, :n_rows].T @ wt[:, :N] + bias with ``at`` [K, ld] k-major (ld >= n_rows rounded up to 128)."""
    for t, nm in ((at, 'at'), (wt, 'wt'), (out, 'out'), (bias, 'bias')):
        if t is not None and not (t.is_cuda and t.dtype == torch.float32 and t.stride(-1) == 1):
            raise ValueError('%s must be an fp32 GPU tensor with unit inner stride' % nm)
    Kd = wt.shape[0]
    N = wt.shape[1] if n_cols is None else n_cols
    assert at.shape[0] == Kd and out.shape[1] >= N and out.shape[0] >= n_rows
    call('tst_gemm_dma_f32', _p(at), at.stride(0), _p(wt), wt.stride(0), _p(bias), _p(out), out.stride(0), n_rows, N, Kd, int(variant),
         _stream())


def layernorm_t(x, gamma, beta, out_t, n_rows, eps=1e-5):
    """out_t[k, r] = LayerNorm(x)[r, k] for r < n_rows; ``out_t`` [d, ld >= n_rows]"""
    _chk(x, torch.float32, 'x'); _chk(gamma, torch.float32, 'gamma'); _chk(beta, torch.float32, 'beta')
    if not (out_t.is_cuda and out_t.dtype == torch.float32 and out_t.stride(1) == 1):
        raise ValueError('out_t must be an fp32 GPU tensor with unit inner stride')
    d = x.shape[1]
    assert out_t.shape[0] == d and out_t.shape[1] >= n_rows
    call('tst_layernorm_t_f32', _p(x), _p(gamma), _p(beta), float(eps), _p(out_t), out_t.stride(0), n_rows, d, _stream())


def attn_self_step(cache, anc, pos, out, n_rows, d, heads):
    """cache [n_pos, slab_rows, 3d] fp32 or bf16 (``out`` of the same type); anc [rows, T] int32"""
    _chk(cache, cache.dtype if cache.dtype == torch.bfloat16 else torch.float32, 'cache'); _chk(anc, torch.int32, 'anc')
    _chk(out, cache.dtype, 'out')
    assert cache.shape[2] == 3 * d and pos < cache.shape[0] and out.shape[1] == d
    call('tst_attn_self_step_bf16' if cache.dtype == torch.bfloat16 else 'tst_attn_self_step_f32', _p(cache), cache.shape[1], _p(anc), anc.shape[1], pos, _p(out), n_rows, d, heads, _stream())


def attn_cross_step(q, kv, mask, sent, out, n_rows, d, heads, kvrow=None, group=None):
    """q [rows, d]; kv [n_sent, S, 2d]; mask [n_sent, S] uint8 or None; sent [rows] int32.
    With ``kvrow`` ([n_sent, S] int32) ``kv`` is packed, [n_packed, 2d]: the row of (sentence, position) is kvrow[s, p].
    ``group`` (fp32): adjacent rows in groups of that size are hypotheses of one sentence (keys / values loaded once)."""
    if group is not None:
        _chk(q, torch.float32, 'q'); _chk(kv, torch.float32, 'kv'); _chk(mask, torch.uint8, 'mask'); _chk(sent, torch.int32, 'sent')
        _chk(out, torch.float32, 'out'); _chk(kvrow, torch.int32, 'kvrow')
        S = mask.shape[1] if mask is not None else kv.shape[1]
        call('tst_attn_cross_group_f32', _p(q), _p(kv), _p(mask), _p(sent), _p(kvrow), S, _p(out), n_rows, d, heads, int(group), _stream())
        return
    dt = torch.bfloat16 if q.dtype == torch.bfloat16 else torch.float32
    _chk(q, dt, 'q'); _chk(kv, dt, 'kv'); _chk(mask, torch.uint8, 'mask'); _chk(sent, torch.int32, 'sent')
    _chk(out, dt, 'out')
    if kvrow is not None:
        _chk(kvrow, torch.int32, 'kvrow')
        assert dt == torch.float32 and mask is not None and tuple(kvrow.shape) == tuple(mask.shape) and kv.shape[-1] == 2 * d
        call('tst_attn_cross_packed_f32', _p(q), _p(kv), _p(mask), _p(sent), _p(kvrow), mask.shape[1], _p(out), n_rows, d, heads, _stream())
        return
    assert kv.shape[2] == 2 * d and (mask is None or tuple(mask.shape) == tuple(kv.shape[:2]))
    call('tst_attn_cross_step_bf16' if dt == torch.bfloat16 else 'tst_attn_cross_step_f32', _p(q), _p(kv), _p(mask), _p(sent), kv.shape[1], _p(out), n_rows, d, heads, _stream())


def attn_cross_mfma(q, kv, mask, sent, out, n_rows, d, heads, group, kvrow=None):
    """``attn_cross_step(..., group=)`` on the matrix cores (scores in chain order); fp32, S <= 64, group 2..8"""
    _chk(q, torch.float32, 'q'); _chk(kv, torch.float32, 'kv'); _chk(mask, torch.uint8, 'mask'); _chk(sent, torch.int32, 'sent')
    _chk(out, torch.float32, 'out'); _chk(kvrow, torch.int32, 'kvrow')
    S = mask.shape[1] if mask is not None else kv.shape[1]
    call('tst_attn_cross_mfma_f32', _p(q), _p(kv), _p(mask), _p(sent), _p(kvrow), S, _p(out), n_rows, d, heads, int(group), _stream())


def gru_cell_step(gi, gh, h, h_out, n_rows):
    for t, nm in ((gi, 'gi'), (gh, 'gh'), (h, 'h'), (h_out, 'h_out')):
        _chk(t, torch.float32, nm)
    H = h.shape[1]
    assert gi.shape[1] == 3 * H and gh.shape[1] == 3 * H
    call('tst_gru_cell_step_f32', _p(gi), _p(gh), _p(h), _p(h_out), n_rows, H, _stream())


def gru_cell_fused_ok(E, H):
    return bool(_lib._dll.tst_gru_cell_fused_ok(E, H))


def gru_cell_fused(x, h, wih, bih, whh, bhh, h_out, n_rows):
    """h_out[:n_rows] = GRU cell(x @ wih + bih, h @ whh + bhh, h) as one kernel; ``wih`` [E, 3H], ``whh`` [H, 3H] (transposed
    nn.GRU weights); ``h_out`` must not overlap ``h``."""
    for t, nm in ((x, 'x'), (h, 'h'), (wih, 'wih'), (whh, 'whh'), (h_out, 'h_out'), (bih, 'bih'), (bhh, 'bhh')):
        if t is not None and not (t.is_cuda and t.dtype == torch.float32 and t.stride(-1) == 1):
            raise ValueError('%s must be an fp32 GPU tensor with unit inner stride' % nm)
    E, H = wih.shape[0], whh.shape[0]
    assert x.shape[1] == E and h.shape[1] == H and h_out.shape[1] == H and min(x.shape[0], h.shape[0], h_out.shape[0]) >= n_rows
    call('tst_gru_cell_fused_f32', _p(x), x.stride(0), _p(h), h.stride(0), _p(wih), wih.stride(0), _p(bih), _p(whh), whh.stride(0),
         _p(bhh), _p(h_out), h_out.stride(0), n_rows, E, H, _stream())


def gru_cell_fused_t(x, h, wih, bih, whh, bhh, h_out, h_out_t, n_rows):
    """``gru_cell_fused`` that also writes the new state k-major: ``h_out_t`` [H, ld >= n_rows]"""
    for t, nm in ((x, 'x'), (h, 'h'), (wih, 'wih'), (whh, 'whh'), (h_out, 'h_out'), (bih, 'bih'), (bhh, 'bhh'), (h_out_t, 'h_out_t')):
        if t is not None and not (t.is_cuda and t.dtype == torch.float32 and t.stride(-1) == 1):
            raise ValueError('%s must be an fp32 GPU tensor with unit inner stride' % nm)
    E, H = wih.shape[0], whh.shape[0]
    assert x.shape[1] == E and h.shape[1] == H and h_out.shape[1] == H and min(x.shape[0], h.shape[0], h_out.shape[0]) >= n_rows
    assert h_out_t.shape[0] == H and h_out_t.shape[1] >= n_rows
    call('tst_gru_cell_fused_t_f32', _p(x), x.stride(0), _p(h), h.stride(0), _p(wih), wih.stride(0), _p(bih), _p(whh), whh.stride(0),
         _p(bhh), _p(h_out), h_out.stride(0), _p(h_out_t), h_out_t.stride(0), n_rows, E, H, _stream())


def gru_cell_fused_tab(gi_tab, tok, h, whh, bhh, h_out, n_rows, h_out_t=None):
    """``gru_cell_fused`` for inputs that are embedding lookups: the input pre-activations of row r are row ``tok[r]`` of
    ``gi_tab`` [V, 3H] (= bias_ih + embedding @ wih, see ``gi_table``) instead of a product computed here."""
    for t, nm in ((gi_tab, 'gi_tab'), (h, 'h'), (whh, 'whh'), (h_out, 'h_out'), (bhh, 'bhh'), (h_out_t, 'h_out_t')):
        if t is not None and not (t.is_cuda and t.dtype == torch.float32 and t.stride(-1) == 1):
            raise ValueError('%s must be an fp32 GPU tensor with unit inner stride' % nm)
    _chk(tok, torch.int32, 'tok')
    H = whh.shape[0]
    assert gi_tab.shape[1] >= 3 * H and h.shape[1] == H and h_out.shape[1] == H and min(tok.numel(), h.shape[0], h_out.shape[0]) >= n_rows
    assert h_out_t is None or (h_out_t.shape[0] == H and h_out_t.shape[1] >= n_rows)
    call('tst_gru_cell_fused_tab_f32', _p(gi_tab), gi_tab.stride(0), _p(tok), gi_tab.shape[0], _p(h), h.stride(0), _p(whh), whh.stride(0),
         _p(bhh), _p(h_out), h_out.stride(0), _p(h_out_t), 0 if h_out_t is None else h_out_t.stride(0), n_rows, H, _stream())


def gru_cell_step_tab(gi_tab, tok, gh, h, h_out, n_rows):
    """``gru_cell_step`` with the input pre-activations of row r taken from row ``tok[r]`` of ``gi_tab`` [V, 3H]."""
    if not (gi_tab.is_cuda and gi_tab.dtype == torch.float32 and gi_tab.stride(-1) == 1):
        raise ValueError('gi_tab must be an fp32 GPU tensor with unit inner stride')
    _chk(gh, torch.float32, 'gh'); _chk(h, torch.float32, 'h'); _chk(h_out, torch.float32, 'h_out')
    _chk(tok, torch.int32, 'tok')
    H = h.shape[1]
    assert gi_tab.shape[1] >= 3 * H and gh.shape[1] == 3 * H and h_out.shape[1] == H
    call('tst_gru_cell_step_tab_f32', _p(gi_tab), gi_tab.stride(0), _p(tok), gi_tab.shape[0], _p(gh), _p(h), _p(h_out), n_rows, H, _stream())


def gi_table(emb, wih, bih):
    """[V, 3H] = bias_ih + emb @ wih: the input pre-activations of a GRU layer for every vocabulary entry, by the same product
    kernel (hence the same fma chain per element) that a step would run on the looked-up embeddings."""
    V = emb.shape[0]
    out = torch.empty((V, wih.shape[1]), dtype=torch.float32, device=emb.device)
    gemm(emb, wih, bih, out, V)
    return out


def gather_rows(src, idx, dst, n_rows):
    _chk(src, torch.float32, 'src'); _chk(dst, torch.float32, 'dst'); _chk(idx, torch.int32, 'idx')
    assert src.shape[1] == dst.shape[1]
    call('tst_gather_rows_f32', _p(src), _p(idx), _p(dst), n_rows, src.shape[1], _stream())


def scatter_rows(src, idx, dst, n_rows):
    """dst[idx[r]] = src[r] for r < n_rows"""
    _chk(src, torch.float32, 'src'); _chk(dst, torch.float32, 'dst'); _chk(idx, torch.int32, 'idx')
    assert src.shape[1] == dst.shape[1]
    call('tst_scatter_rows_f32', _p(src), _p(idx), _p(dst), n_rows, src.shape[1], _stream())


class PackedSources:
    """The non-padding positions of a source batch [B, S] as packed rows (``tst_pack_sources``): ``mask`` [B, S] uint8,
    ``kvrow`` [B, S] int32, and per packed row ``tok``, ``pos``, ``sent``, ``padded_row`` (first ``n`` entries valid).  One
    device-to-host read (the number of tokens)."""

    def __init__(self, src, pad):
        if not (src.is_cuda and src.dtype == torch.int64 and src.dim() == 2 and src.is_contiguous()):
            raise ValueError('src must be a contiguous int64 [B, S] tensor on the GPU')
        B, S = src.shape
        dev = src.device
        self.B, self.S = B, S
        self.mask = torch.empty((B, S), dtype=torch.uint8, device=dev)
        shapes = [(B, S), (B * S,), (B * S,), (B * S,), (B * S,), (2 * B,), (1,)]
        arena = torch.empty(_arena_size(shapes), dtype=torch.int32, device=dev)
        self.kvrow, self.tok, self.pos, self.sent, self.padded_row, work, n_live = _carve(arena, shapes)
        call('tst_pack_sources', _p(src), int(pad), B, S, _p(self.mask), _p(self.kvrow), _p(self.tok), _p(self.pos), _p(self.sent),
             _p(self.padded_row), _p(work), _p(n_live), _stream())
        self.n = int(n_live.item())


def _logits_suffix(logits):
    if not (logits.is_cuda and logits.dtype in (torch.float32, torch.bfloat16) and logits.stride(-1) == 1):
        raise ValueError('logits must be an fp32 or bf16 GPU tensor with unit inner stride')
    return '_bf16' if logits.dtype == torch.bfloat16 else '_f32'


def argmax_rows(logits, n_rows, n_cols, out_tok):
    _chk(out_tok, torch.int32, 'out_tok')
    call('tst_argmax_rows' + _logits_suffix(logits), _p(logits), logits.stride(0), n_rows, n_cols, _p(out_tok), _stream())


def sample_rows(logits, n_rows, n_cols, temperature, seed, step, sent, out_tok):
    _chk(out_tok, torch.int32, 'out_tok'); _chk(sent, torch.int32, 'sent')
    call('tst_sample_rows' + _logits_suffix(logits), _p(logits), logits.stride(0), n_rows, n_cols, float(temperature), int(seed) & 0xffffffff,
         int(step), _p(sent), _p(out_tok), _stream())


def vocab_stats_ok(n_rows, n_cols, k_inner):
    return bool(_lib._dll.tst_vocab_stats_ok(n_rows, n_cols, k_inner))


def vocabproj_stats(a, w, bias, out, part, n_rows, n_cols=None):
    """out[:n_rows, :N] = a[:n_rows] @ w[:N].T + bias (all bf16 but the fp32 bias) and ``part`` [ceil(rows / 32), ceil(N / 64), 64]
    fp32: per row and slice of 64 columns the largest stored logit and the sum of exp(logit - largest)."""
    for t, nm, dt in ((a, 'a', torch.bfloat16), (w, 'w', torch.bfloat16), (out, 'out', torch.bfloat16), (bias, 'bias', torch.float32)):
        if not (t.is_cuda and t.dtype == dt and t.stride(-1) == 1):
            raise ValueError('%s must be a %s GPU tensor with unit inner stride' % (nm, dt))
    _chk(part, torch.float32, 'part')
    Kd = a.shape[1]
    N = w.shape[0] if n_cols is None else n_cols
    ns = (N + 63) // 64
    assert w.shape[1] == Kd and out.shape[1] >= N and a.shape[0] >= n_rows and out.shape[0] >= n_rows
    assert part.numel() >= (n_rows + 31) // 32 * ns * 64
    call('tst_vocabproj_stats_bf16', _p(a), a.stride(0), _p(w), w.stride(0), _p(bias), _p(out), out.stride(0), n_rows, N, Kd, _p(part), ns,
         _stream())


def topk_from_stats(logits, part, n_rows, n_cols, k, top_val, top_idx, lse):
    _chk(top_val, torch.float32, 'top_val'); _chk(top_idx, torch.int32, 'top_idx'); _chk(lse, torch.float32, 'lse')
    _chk(part, torch.float32, 'part')
    if _logits_suffix(logits) != '_bf16':
        raise ValueError('statistics exist for bf16 logits only')
    call('tst_topk_from_stats_bf16', _p(logits), logits.stride(0), _p(part), (n_cols + 63) // 64, n_rows, n_cols, k, _p(top_val), _p(top_idx),
         _p(lse), _stream())


def logsoftmax_topk(logits, n_rows, n_cols, k, top_val, top_idx, lse):
    _chk(top_val, torch.float32, 'top_val'); _chk(top_idx, torch.int32, 'top_idx'); _chk(lse, torch.float32, 'lse')
    call('tst_logsoftmax_topk' + _logits_suffix(logits), _p(logits), logits.stride(0), n_rows, n_cols, k, _p(top_val), _p(top_idx), _p(lse),
         _stream())


class StepPlan:
    """Host-side ``tst_tfm_step`` of one decode session: all weight, cache and work-buffer pointers of a transformer
    decoder stack, so that a whole model step is ONE native call (``tst_tfm_step_run``).  ``layers`` is a list of dicts
    with the keys of ``tst_tfm_layer`` holding tensors (weights with unit inner stride; their row stride is taken as
    the leading dimension); the plan keeps every tensor alive."""

    def __init__(self, *, d, heads, d_ff, V, bf16, n_pos, slab_rows, S, anc_ld, emb, pe, scale, z, mask, ln, wout, bout,
                 x, h, q, a, ff, logits, layers, eps=1e-5, kvrow=None, group=1, ht=None, part=None):
        self._keep = [kvrow, emb, pe, z, mask, ln, wout, bout, x, h, q, a, ff, logits, layers, ht, part]
        n = len(layers)
        self._layers = (_lib.TstTfmLayer * n)()
        for dst, L in zip(self._layers, layers):
            for name, _ in _lib.TstTfmLayer._fields_:
                if name.startswith('ld_'):
                    w = L.get(name[3:])
                    setattr(dst, name, 0 if w is None else w.stride(0))
                else:
                    setattr(dst, name, _p(L.get(name)))
        self.c = _lib.TstTfmStep(
            n_layers=n, d=d, heads=heads, d_ff=d_ff, V=V, bf16=int(bool(bf16)), group=int(group), n_pos=n_pos, slab_rows=slab_rows, S=S, anc_ld=anc_ld,
            emb=_p(emb), pe=_p(pe), scale=float(scale), eps=float(eps), z=_p(z), mask=_p(mask), kvrow=_p(kvrow), ln_g=_p(ln[0]), ln_b=_p(ln[1]),
            wout=_p(wout), bout=_p(bout), ld_wout=wout.stride(0), x=_p(x), h=_p(h), q=_p(q), a=_p(a), ff=_p(ff), logits=_p(logits),
            ld_logits=logits.stride(0), layers=self._layers, ht=_p(ht), ld_ht=0 if ht is None else ht.stride(0),
            part=_p(part), n_slices=(V + 63) // 64, argmax_keys=None)
        self._keys = None

    def set_keys(self, keys):
        """``keys`` (int64 [>= slab_rows, ARGMAX_SLOTS], or None): the coming calls of ``run`` end in the arg-max keys of the logits instead of the
        logits (``tst_tfm_step.argmax_keys``: fp32, all phases, a shape ``vocab_argmax_ok`` accepts)."""
        if keys is not None:
            _chk(keys, torch.int64, 'keys')
        self._keys = keys
        self.c.argmax_keys = _p(keys)

    def run(self, rows, n_rows, t, phases):
        call('tst_tfm_step_run', self.c, _p(rows.tok), _p(rows.anc), _p(rows.sent), n_rows, t, phases, _stream())


class GruPlan:
    """Host-side ``tst_gru_step`` of one GRU decode session (see ``StepPlan``).  ``cells`` is a list of dicts with
    ``wih, bih, whh, bhh``; ``h`` / ``h_alt`` are the session's lists of state buffers, which ``swap()`` exchanges
    after a reorder."""

    def __init__(self, *, E, H, V, rows, emb, extra, cells, h, h_alt, wout, bout, x, gi, gh, logits, ht=None, gi_tab=None):
        self._keep = [emb, extra, cells, h, h_alt, wout, bout, x, gi, gh, logits, ht, gi_tab]
        n = len(cells)
        self._layers = (_lib.TstGruLayer * n)()
        for dst, c, a, b in zip(self._layers, cells, h, h_alt):
            dst.wih, dst.bih, dst.ld_wih = _p(c['wih']), _p(c['bih']), c['wih'].stride(0)
            dst.whh, dst.bhh, dst.ld_whh = _p(c['whh']), _p(c['bhh']), c['whh'].stride(0)
            dst.h, dst.h_alt = _p(a), _p(b)
        self.c = _lib.TstGruStep(n_layers=n, E=E, H=H, V=V, rows=rows, emb=_p(emb), extra=_p(extra), wout=_p(wout), bout=_p(bout),
                                 ld_wout=wout.stride(0), x=_p(x), gi=_p(gi), gh=_p(gh), logits=_p(logits),
                                 ld_logits=logits.stride(0), layers=self._layers, ht=_p(ht), ld_ht=0 if ht is None else ht.stride(0),
                                 gi_tab=_p(gi_tab), ld_gi_tab=0 if gi_tab is None else gi_tab.stride(0))

    def run(self, rows, n_rows, phases):
        """With ``STEP_FUSED_CELL`` in ``phases`` (and the layers run) the new state of every layer lands in ``h_alt``: the
        pointers are swapped here, the caller swaps its tensors."""
        call('tst_gru_step_run', self.c, _p(rows.tok), _p(rows.sent), n_rows, phases, _stream())
        if phases & STEP_FUSED_CELL and phases & STEP_LAYERS:
            for L in self._layers:
                L.h, L.h_alt = L.h_alt, L.h

    def reorder(self, row_parent, n_rows):
        call('tst_gru_reorder_run', self.c, _p(row_parent), n_rows, _stream())
        for L in self._layers:
            L.h, L.h_alt = L.h_alt, L.h


class SelectPlan:
    """Host-side ``tst_select_step``: selection + retirement + compaction of one decode step as ONE native call."""

    def __init__(self, *, k, eos, V, logits, am, top_val, top_idx, lse, sel, lp, T, top_ends, sample, out_tok, out_len, out_score,
                 scratch, row_parent, n_alive, stats=None):
        # stats: None, or (part, K) of a session whose vocabulary projection leaves per-slice statistics beside its bf16 logits
        self._keep = [logits, am, top_val, top_idx, lse, sel, lp, out_tok, out_len, out_score, scratch, row_parent, n_alive, stats]
        _logits_suffix(logits)   # fp32 or bf16, unit inner stride
        sp = sel.ptrs()
        self.c = _lib.TstSelectStep(
            k=k, eos=eos, V=V, bf16_logits=int(logits.dtype == torch.bfloat16), logits=_p(logits), ld_logits=logits.stride(0),
            am=_p(am), top_val=_p(top_val), top_idx=_p(top_idx), lse=_p(lse), sel_parent=sp[0], sel_tok=sp[1], sel_score=sp[2],
            sel_fin=sp[3], sel_nemit=sp[4], lp=_p(lp), T=T, top_ends=int(top_ends), sample=int(sample is not None),
            temperature=float(sample[0]) if sample is not None else 1.0, seed=(int(sample[1]) & 0xffffffff) if sample is not None else 0,
            out_tok=_p(out_tok), out_len=_p(out_len), out_score=_p(out_score), scratch_alive_pos=_p(scratch),
            row_parent=_p(row_parent), n_alive=_p(n_alive), n_sent_dev=None,
            part=None if stats is None else _p(stats[0]), n_slices=(V + 63) // 64, stats_k=0 if stats is None else int(stats[1]),
            am_keys=None, n_alive_host=None)
        self.logits_ptr = _p(logits)
        self._keys = None

    def set_logits(self, logits):
        """A model that returns a new logits tensor every step (any foreign callable)."""
        _logits_suffix(logits)
        self._keep[0] = logits
        self.c.part = None           # statistics belong to the buffer the plan was made for
        self.c.logits, self.c.ld_logits, self.c.bf16_logits = _p(logits), logits.stride(0), int(logits.dtype == torch.bfloat16)
        self.logits_ptr = _p(logits)

    def set_counts(self, n_alive, n_sent_dev):
        """Where the step writes the number of surviving sentences, and (or None) where the true number of sentences
        entering the step is read from when ``n_sent`` of ``run`` is an upper bound."""
        self._keep_counts = (n_alive, n_sent_dev)
        self.c.n_alive, self.c.n_sent_dev = _p(n_alive), _p(n_sent_dev)

    def set_keys(self, keys):
        """``keys`` (int64, or None): the coming steps decode their tokens from these arg-max keys (greedy selection only)
        instead of reading the logits."""
        self._keys = keys
        self.c.am_keys = _p(keys)

    def set_count_host(self, addr):
        """``addr`` (int, or None): device-visible address of 4 bytes of page-locked host memory that also receives the number
        of surviving sentences of the coming steps."""
        self.c.n_alive_host = addr

    def run(self, cur, nxt, n_sent, k_in, step, sample_step, last_step):
        call('tst_select_step_run', self.c, cur.struct(), nxt.struct(), n_sent, k_in, step, sample_step, int(last_step), _stream())


def _carve(arena, shapes):
    """Views of the given shapes into a 1-D arena, each starting at a multiple of 64 elements (256 bytes)."""
    out, at = [], 0
    for shape in shapes:
        n = 1
        for x in shape:
            n *= x
        out.append(arena[at:at + n].view(shape))
        at += (n + 63) // 64 * 64
    return out


def _arena_size(shapes):
    total = 0
    for shape in shapes:
        n = 1
        for x in shape:
            n *= x
        total += (n + 63) // 64 * 64
    return total


class RowState:
    """Per-row decode state on the device (``tst_rows``): token history, ancestor table, scores, flags.  The int32
    arrays are views into one allocation (a decode call makes two of these objects; at small batches the time to
    allocate and clear many little tensors is a visible part of the call)."""

    def __init__(self, rows, T, device, with_anc, fill_token):
        shapes = [(rows, T + 1), (rows, T) if with_anc else (0,), (rows,), (rows,), (rows,), (rows,)]
        arena = torch.zeros(_arena_size(shapes), dtype=torch.int32, device=device)
        self.hist, anc, self.sent, self.tok, self.fin, self.nemit = _carve(arena, shapes)
        self.anc = anc if with_anc else None
        if fill_token != 0:
            self.hist.fill_(fill_token)
        self.score = torch.zeros(rows, dtype=torch.float32, device=device)
        self._struct = None

    def struct(self):
        if self._struct is None:   # the buffers live as long as this object
            self._struct = _lib.TstRows(_p(self.hist), _p(self.anc), _p(self.sent), _p(self.tok), _p(self.score), _p(self.fin),
                                        _p(self.nemit))
        return self._struct


class Selection:
    """Output arrays of the select kernels (one entry per new row slot)."""

    def __init__(self, rows, device):
        shapes = [(rows,)] * 4
        arena = torch.zeros(_arena_size(shapes), dtype=torch.int32, device=device)
        self.parent, self.tok, self.fin, self.nemit = _carve(arena, shapes)
        self.score = torch.zeros(rows, dtype=torch.float32, device=device)

    def ptrs(self):
        return _p(self.parent), _p(self.tok), _p(self.score), _p(self.fin), _p(self.nemit)


def greedy_select(argmax_tok, n_rows, eos, sel, cur):
    st = cur.struct()
    call('tst_greedy_select', _p(argmax_tok), n_rows, eos, *sel.ptrs(), st, _stream())


def beam_merge(cur, top_val, top_idx, lse, n_sent, k, k_in, eos, sel):
    st = cur.struct()
    call('tst_beam_merge', st, _p(top_val), _p(top_idx), _p(lse), n_sent, k, k_in, eos, *sel.ptrs(), _stream())


def retire_compact(cur, nxt, sel, lp, n_sent, k, step, T, eos, last_step, top_ends, out_tok, out_len, out_score, scratch,
                   row_parent, n_alive):
    a, b = cur.struct(), nxt.struct()
    call('tst_retire_compact', a, b, *sel.ptrs(), _p(lp), n_sent, k, step, T, eos, int(last_step), int(top_ends), _p(out_tok), _p(out_len),
         _p(out_score), _p(scratch), _p(row_parent), _p(n_alive), _stream())
