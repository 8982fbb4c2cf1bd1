"""GPU: every kernel, called through the C ABI, against the canonical oracle — bit for bit
(``np.array_equal``; the canonical arithmetic fixes every fp32 operation and its order)."""
import numpy as np
import pytest
import torch

from oracle import canon as C

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def K():
    import tst_amd
    from tst_amd import kernels
    assert torch.cuda.is_available()
    return kernels


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def rs(seed):
    return np.random.RandomState(seed)


GEMM_SHAPES = [
    (1, 4, 4, False, 0, False), (5, 50, 48, True, 0, False), (130, 200, 36, True, 1, False),
    (300, 1000, 256, True, 0, True), (257, 128, 1024, True, 1, True), (64, 3072, 1024, True, 0, False),
    (1000, 768, 256, True, 0, False), (129, 129, 20, False, 1, True), (37, 90, 132, True, 0, True)]


def _gemm_case(K, M, N, K_, bias, act, res, strip=None, odd_ldc=False):
    r = rs(M * 7 + N)
    A = r.standard_normal((M, K_)).astype(np.float32)
    W = r.standard_normal((K_, N)).astype(np.float32)
    b = r.standard_normal(N).astype(np.float32) if bias else None
    R0 = r.standard_normal((M, N)).astype(np.float32) if res else None
    ldw = (N + 3) // 4 * 4
    Wd = torch.zeros((K_, ldw), dtype=torch.float32, device='cuda')
    Wd[:, :N] = dev(W)
    ldc = ldw + (3 if odd_ldc else 0)      # an odd row stride of C: no vector stores possible
    out = torch.full((M, ldc), 7.0, dtype=torch.float32, device='cuda')
    K.gemm(dev(A), Wd[:, :N], None if b is None else dev(b), out, M, n_cols=N, res=None if R0 is None else dev(R0), act=act, strip=strip)
    want = C.gemm(A, W, b, R0, act)
    got = out.cpu().numpy()
    assert np.array_equal(got[:, :N], want)
    assert (got[:, N:] == 7.0).all()  # nothing written beyond column N


@pytest.mark.parametrize('M,N,K_,bias,act,res', GEMM_SHAPES + [
    (2100, 520, 132, True, 1, True),      # beyond the small-problem kernel: 64 x 64 x 16 tiles (K % 32 != 0)
    (2100, 520, 128, True, 1, True),      # 64 x 64 x 32 tiles: relu + residual, partial tiles in both directions
    (3000, 300, 256, False, 0, False),    # 64 x 64 x 32 tiles, no bias
    (4100, 2000, 64, True, 0, True),      # 64 x 64 x 32 tiles, two k-tiles only
    (11000, 1100, 48, True, 0, True)])    # 128 x 128 tiles
def test_gemm_bit_exact(K, M, N, K_, bias, act, res):
    _gemm_case(K, M, N, K_, bias, act, res)


@pytest.mark.parametrize('strip', [1, 2, 4])
@pytest.mark.parametrize('M,N,K_,bias,act,res', GEMM_SHAPES)
def test_gemm_small_problem_kernel_bit_exact(K, M, N, K_, bias, act, res, strip):
    """The one-wave-per-strip kernel at every strip width, whatever size the dispatcher would give it."""
    _gemm_case(K, M, N, K_, bias, act, res, strip=strip)
    if M == 300 or M == 37:
        _gemm_case(K, M, N, K_, bias, act, res, strip=strip, odd_ldc=True)


@pytest.mark.parametrize('M,N,K_,act', [(1, 256, 256, 0), (37, 768, 256, 0), (256, 1024, 256, 1), (300, 90, 256, 1), (640, 768, 256, 0),
                                        (1000, 768, 256, 0), (130, 1536, 512, 0), (70, 2048, 512, 1), (3000, 1024, 256, 1),
                                        (50, 200, 64, 0)])
def test_layernorm_gemm_bit_exact(K, M, N, K_, act):
    """LayerNorm + product as one entry point: the fused small-problem kernel (row lengths 256 and 512, all strip widths),
    and the two-kernel route for everything else, against the canonical LayerNorm followed by the canonical product."""
    r = rs(M + N + K_)
    x = (r.standard_normal((M, K_)) * 2 + 0.5).astype(np.float32)
    g, b = r.standard_normal(K_).astype(np.float32), r.standard_normal(K_).astype(np.float32)
    W = r.standard_normal((K_, N)).astype(np.float32)
    bias = r.standard_normal(N).astype(np.float32)
    ldw = (N + 3) // 4 * 4
    Wd = torch.zeros((K_, ldw), dtype=torch.float32, device='cuda')
    Wd[:, :N] = dev(W)
    out = torch.full((M, ldw), 7.0, dtype=torch.float32, device='cuda')
    h = torch.empty((M, K_), dtype=torch.float32, device='cuda')
    K.ln_gemm(dev(x), dev(g), dev(b), h, Wd[:, :N], dev(bias), out, M, n_cols=N, act=act)
    want = C.gemm(C.layernorm(x, g, b), W, bias, None, act)
    got = out.cpu().numpy()
    assert np.array_equal(got[:, :N], want)
    assert (got[:, N:] == 7.0).all()


def test_gemm_in_place_residual_and_row_prefix(K):
    r = rs(3)
    A = r.standard_normal((200, 64)).astype(np.float32)
    W = r.standard_normal((64, 64)).astype(np.float32)
    X = r.standard_normal((200, 64)).astype(np.float32)
    x = dev(X)
    K.gemm(dev(A), dev(W), None, x, 150, res=x)
    got = x.cpu().numpy()
    assert np.array_equal(got[:150], C.gemm(A[:150], W, None, X[:150]))
    assert np.array_equal(got[150:], X[150:])


@pytest.mark.parametrize('d', [32, 256, 512, 1000])
def test_layernorm_bit_exact(K, d):
    r = rs(d)
    x = (r.standard_normal((37, d)) * 3 + 1).astype(np.float32)
    g, b = r.standard_normal(d).astype(np.float32), r.standard_normal(d).astype(np.float32)
    y = torch.empty((37, d), dtype=torch.float32, device='cuda')
    K.layernorm(dev(x), dev(g), dev(b), y, 37)
    assert np.array_equal(y.cpu().numpy(), C.layernorm(x, g, b))


def test_embed_posadd_bit_exact(K):
    r = rs(5)
    emb = r.standard_normal((50, 64)).astype(np.float32)
    pe = r.standard_normal((9, 64)).astype(np.float32)
    tok = r.randint(0, 50, size=33).astype(np.int32)
    out = torch.empty((33, 64), dtype=torch.float32, device='cuda')
    K.embed_posadd(dev(tok), dev(emb), dev(pe[3]), 8.0, out, 33)
    assert np.array_equal(out.cpu().numpy(), C.embed_posadd(tok, emb, pe[3], 8.0))
    pos = r.randint(0, 9, size=33).astype(np.int32)
    K.embed_posadd(dev(tok), dev(emb), dev(pe), 8.0, out, 33, pos=dev(pos))
    want = np.stack([C.embed_posadd(tok[i:i + 1], emb, pe[pos[i]], 8.0)[0] for i in range(33)])
    assert np.array_equal(out.cpu().numpy(), want)
    K.embed_posadd(dev(tok), dev(emb), None, 1.0, out, 33)
    assert np.array_equal(out.cpu().numpy(), emb[tok])
    # with the conditioning vector of each row's sentence added in the same pass: == embedding, then the row add
    z = r.standard_normal((7, 64)).astype(np.float32)
    zidx = r.randint(0, 7, size=33).astype(np.int32)
    K.embed_posadd_rows(dev(tok), dev(emb), dev(pe[3]), 8.0, dev(z), dev(zidx), out, 33)
    assert np.array_equal(out.cpu().numpy(), C.embed_posadd(tok, emb, pe[3], 8.0) + z[zidx])
    ref = torch.empty((33, 64), dtype=torch.float32, device='cuda')
    K.embed_posadd(dev(tok), dev(emb), dev(pe[3]), 8.0, ref, 33)
    K.add_rows(ref, dev(z), dev(zidx), 33)
    assert torch.equal(out, ref)


@pytest.mark.parametrize('d,heads,pos', [(256, 2, 0), (256, 2, 7), (256, 2, 49), (512, 8, 20), (64, 2, 5), (96, 1, 70),
                                         (512, 8, 0), (512, 8, 130), (256, 4, 63), (768, 12, 200)])
def test_attn_self_step_bit_exact(K, d, heads, pos):
    r = rs(d + pos)
    R, slab, T = 21, 32, pos + 3
    cache = r.standard_normal((T, slab, 3 * d)).astype(np.float32)
    anc = r.randint(0, slab, size=(R, T)).astype(np.int32)
    out = torch.empty((R, d), dtype=torch.float32, device='cuda')
    K.attn_self_step(dev(cache), dev(anc), pos, out, R, d, heads)
    # oracle: gather each row's logical K/V sequence
    slots = anc[:, :pos + 1].copy()
    slots[:, pos] = np.arange(R)
    Kc = np.stack([np.stack([cache[p, slots[i, p], d:2 * d] for p in range(pos + 1)]) for i in range(R)])
    Vc = np.stack([np.stack([cache[p, slots[i, p], 2 * d:] for p in range(pos + 1)]) for i in range(R)])
    q = np.stack([cache[pos, i, :d] for i in range(R)])
    assert np.array_equal(out.cpu().numpy(), C.attention(q, Kc, Vc, None, None, pos + 1, heads))


@pytest.mark.parametrize('d,heads,S,masked', [(256, 2, 12, True), (256, 2, 42, True), (512, 8, 9, False), (128, 1, 100, True),
                                            (256, 4, 70, True), (512, 8, 150, True)])
def test_attn_cross_step_bit_exact(K, d, heads, S, masked):
    r = rs(d + S)
    B, R = 7, 26
    q = r.standard_normal((R, d)).astype(np.float32)
    kv = r.standard_normal((B, S, 2 * d)).astype(np.float32)
    sent = r.randint(0, B, size=R).astype(np.int32)
    lens = r.randint(1, S + 1, size=B)
    mask = (np.arange(S)[None] < lens[:, None]).astype(np.uint8) if masked else None
    out = torch.empty((R, d), dtype=torch.float32, device='cuda')
    K.attn_cross_step(dev(q), dev(kv), None if mask is None else dev(mask), dev(sent), out, R, d, heads)
    want = C.attention(q, kv[:, :, :d], kv[:, :, d:], sent, mask, S, heads)
    assert np.array_equal(out.cpu().numpy(), want)


@pytest.mark.parametrize('group', [2, 4, 8, 3])
@pytest.mark.parametrize('d,heads,S', [(256, 2, 42), (128, 4, 17), (64, 1, 70), (512, 8, 9)])
def test_attn_cross_group_bit_exact(K, d, heads, S, group):
    """Beams of a sentence as adjacent rows: keys / values loaded once per group.  Same bits as the oracle for groups that
    share a sentence, for groups that do not (computed row by row), with a packed k|v, and for the sizes the grouped
    kernel is not built for (group 3, 64-element heads: the row-per-wave kernel runs).  The matrix-core variant of the
    kernel (keys staged in LDS, scores and outputs as MFMA products) equals the oracle with chain-order scores."""
    r = rs(d + S + group)
    B = 9
    n_groups = 11
    R = n_groups * group
    q = r.standard_normal((R, d)).astype(np.float32)
    kv = r.standard_normal((B, S, 2 * d)).astype(np.float32)
    sent = np.repeat(r.randint(0, B, size=n_groups), group).astype(np.int32)
    sent[group:2 * group] = r.randint(0, B, size=group)          # one group of unrelated rows
    sent[-1] = (sent[-1] + 1) % B                                # and one with a single stray row
    lens = r.randint(2, S + 1, size=B)                            # every sentence keeps a position after the hole below
    mask = (np.arange(S)[None] < lens[:, None]).astype(np.uint8)
    mask[3, 0] = 0                                                # a hole, not only trailing padding
    want = C.attention(q, kv[:, :, :d], kv[:, :, d:], sent, mask, S, heads)
    out = torch.full((R, d), 7.0, dtype=torch.float32, device='cuda')
    K.attn_cross_step(dev(q), dev(kv), dev(mask), dev(sent), out, R, d, heads, group=group)
    assert np.array_equal(out.cpu().numpy(), want)
    if S <= 64:   # the matrix-core kernel: scores as single fma chains, otherwise the same arithmetic
        want_chain = C.attention(q, kv[:, :, :d], kv[:, :, d:], sent, mask, S, heads, chain=True)
        assert not np.array_equal(want_chain, want)                       # the two orders differ in the last bits
        np.testing.assert_allclose(want_chain, want, rtol=2e-5, atol=2e-6)
        outm = torch.full((R, d), 7.0, dtype=torch.float32, device='cuda')
        K.attn_cross_mfma(dev(q), dev(kv), dev(mask), dev(sent), outm, R, d, heads, group)
        assert np.array_equal(outm.cpu().numpy(), want_chain)
    # packed k|v: only the unmasked positions exist
    live = mask.reshape(-1) != 0
    kvrow = (np.cumsum(live) - 1).astype(np.int32).reshape(B, S)
    packed = kv.reshape(B * S, 2 * d)[live]
    out2 = torch.full((R, d), 7.0, dtype=torch.float32, device='cuda')
    K.attn_cross_step(dev(q), dev(packed), dev(mask), dev(sent), out2, R, d, heads, kvrow=dev(kvrow), group=group)
    assert np.array_equal(out2.cpu().numpy(), want)
    # a row count that is no multiple of the group
    out3 = torch.full((R, d), 7.0, dtype=torch.float32, device='cuda')
    K.attn_cross_step(dev(q), dev(kv), dev(mask), dev(sent), out3, R - 1, d, heads, group=group)
    assert np.array_equal(out3.cpu().numpy()[:R - 1], want[:R - 1]) and (out3.cpu().numpy()[R - 1] == 7.0).all()


def test_gru_cell_bit_exact(K):
    r = rs(11)
    R, H = 19, 64
    gi, gh = (r.standard_normal((R, 3 * H)) * 2).astype(np.float32), (r.standard_normal((R, 3 * H)) * 2).astype(np.float32)
    h = r.standard_normal((R, H)).astype(np.float32)
    hd = dev(h)
    K.gru_cell_step(dev(gi), dev(gh), hd, hd, R)
    assert np.array_equal(hd.cpu().numpy(), C.gru_cell(gi, gh, h))


def test_gather_rows(K):
    r = rs(12)
    src = r.standard_normal((40, 32)).astype(np.float32)
    idx = r.randint(0, 40, size=25).astype(np.int32)
    dst = torch.zeros((25, 32), dtype=torch.float32, device='cuda')
    K.gather_rows(dev(src), dev(idx), dst, 25)
    assert np.array_equal(dst.cpu().numpy(), src[idx])


@pytest.mark.parametrize('V', [5, 50, 1000, 32000, 40001])
def test_argmax_rows_first_max(K, V):
    r = rs(V)
    x = r.standard_normal((13, V)).astype(np.float32)
    x[1, :] = 0.5                      # all equal: index 0
    x[2, [V - 1, V // 2]] = 9.0        # tie: the earlier index
    x[3, V - 1] = 9.0                  # max in the last (possibly partial) chunk
    ld = (V + 3) // 4 * 4
    xd = torch.zeros((13, ld), dtype=torch.float32, device='cuda')
    xd[:, :V] = dev(x)
    tok = torch.empty(13, dtype=torch.int32, device='cuda')
    K.argmax_rows(xd, 13, V, tok)
    assert np.array_equal(tok.cpu().numpy(), C.argmax_rows(x))
    assert np.array_equal(tok.cpu().numpy(), x.argmax(1))


@pytest.mark.parametrize('V,k', [(8, 8), (50, 3), (1000, 4), (32000, 4), (32000, 8), (32000, 1), (40004, 5), (33001, 2)])
def test_logsoftmax_topk_bit_exact(K, V, k):
    r = rs(V + k)
    R = 9
    x = (r.standard_normal((R, V)) * 2).astype(np.float32)
    x[1, :] = -1.25                                  # all equal: candidate list overflows -> whole-row rounds
    x[2, r.randint(0, V, size=min(V, 3000))] = 4.0   # many ties at the top
    x[3, V - 1] = 50.0
    ld = (V + 3) // 4 * 4
    xd = torch.zeros((R, ld), dtype=torch.float32, device='cuda')
    xd[:, :V] = dev(x)
    tv = torch.empty((R, k), dtype=torch.float32, device='cuda')
    ti = torch.empty((R, k), dtype=torch.int32, device='cuda')
    lse = torch.empty(R, dtype=torch.float32, device='cuda')
    K.logsoftmax_topk(xd, R, V, k, tv, ti, lse)
    wv, wi, wl = C.logsoftmax_topk(x, k)
    assert np.array_equal(ti.cpu().numpy(), wi)
    assert np.array_equal(tv.cpu().numpy(), wv)
    assert np.array_equal(lse.cpu().numpy(), wl)
    ref = torch.logsumexp(torch.from_numpy(x).double(), -1).numpy()
    assert np.allclose(wl, ref, rtol=1e-6, atol=1e-6)


def test_logsoftmax_topk_unaligned_rows(K):
    r = rs(77)
    V, R, k = 1001, 6, 4
    x = r.standard_normal((R, V)).astype(np.float32)
    xd = dev(x)  # ld = 1001: rows are not 16-byte aligned
    tv = torch.empty((R, k), dtype=torch.float32, device='cuda')
    ti = torch.empty((R, k), dtype=torch.int32, device='cuda')
    lse = torch.empty(R, dtype=torch.float32, device='cuda')
    K.logsoftmax_topk(xd, R, V, k, tv, ti, lse)
    wv, wi, wl = C.logsoftmax_topk(x, k)
    assert np.array_equal(ti.cpu().numpy(), wi) and np.array_equal(tv.cpu().numpy(), wv) and np.array_equal(lse.cpu().numpy(), wl)
    tok = torch.empty(R, dtype=torch.int32, device='cuda')
    K.argmax_rows(xd, R, V, tok)
    assert np.array_equal(tok.cpu().numpy(), x.argmax(1))


@pytest.mark.parametrize('V,temp', [(50, 1.0), (1000, 0.7), (32000, 1.3)])
def test_sample_rows_bit_exact(K, V, temp):
    r = rs(V)
    R = 40
    x = (r.standard_normal((R, V)) * 2).astype(np.float32)
    sent = r.permutation(1000)[:R].astype(np.int32)
    tok = torch.empty(R, dtype=torch.int32, device='cuda')
    K.sample_rows(dev(x), R, V, temp, 1234, 7, dev(sent), tok)
    assert np.array_equal(tok.cpu().numpy(), C.sample_rows(x, temp, 1234, 7, sent))


@pytest.mark.parametrize('R,E,H', [(300, 64, 128), (129, 1024, 1024), (1, 32, 64), (2048, 256, 192)])
def test_gru_cell_fused_equals_three_call_form_and_oracle(K, R, E, H):
    """One kernel (gate pre-activations in the accumulators) against x Wih, h Whh, gru_cell as three calls and against the
    canonical oracle: bit for bit; rows beyond R untouched; strided x / h / h_out views."""
    rs = np.random.RandomState(R + E + H)
    x = rs.standard_normal((R, E)).astype(np.float32)
    h = rs.standard_normal((R, H)).astype(np.float32)
    wih = (rs.standard_normal((E, 3 * H)) / np.sqrt(E)).astype(np.float32)
    whh = (rs.standard_normal((H, 3 * H)) / np.sqrt(H)).astype(np.float32)
    bih, bhh = (0.1 * rs.standard_normal(3 * H)).astype(np.float32), (0.1 * rs.standard_normal(3 * H)).astype(np.float32)
    want = C.gru_cell(C.gemm(x, wih, bih), C.gemm(h, whh, bhh), h)
    xd, hd, wi, wh, bi, bh = (torch.from_numpy(a).cuda() for a in (x, h, wih, whh, bih, bhh))
    out = torch.full((R + 2, H), 7.0, device='cuda')
    K.gru_cell_fused(xd, hd, wi, bi, wh, bh, out, R)
    assert np.array_equal(out[:R].cpu().numpy(), want) and (out[R:] == 7.0).all()
    gi, gh = torch.empty((R, 3 * H), device='cuda'), torch.empty((R, 3 * H), device='cuda')
    K.gemm(xd, wi, bi, gi, R)
    K.gemm(hd, wh, bh, gh, R)
    three = torch.empty((R, H), device='cuda')
    K.gru_cell_step(gi, gh, hd, three, R)
    assert torch.equal(out[:R], three)
    # strided operands, no biases
    xs = torch.zeros((R, E + 8), device='cuda'); xs[:, :E] = xd
    hs = torch.zeros((R, H + 4), device='cuda'); hs[:, :H] = hd
    outs = torch.zeros((R, H + 12), device='cuda')
    K.gru_cell_fused(xs[:, :E], hs[:, :H], wi, None, wh, None, outs[:, :H], R)
    assert np.array_equal(outs[:, :H].cpu().numpy(), C.gru_cell(C.gemm(x, wih), C.gemm(h, whh), h)) and (outs[:, H:] == 0).all()
    # the form that also writes the new state k-major (the operand layout of the LDS-DMA output projection): same bits
    out2 = torch.full((R + 2, H), 7.0, device='cuda')
    ld = (R + 127) // 128 * 128
    out_t = torch.full((H, ld), 7.0, device='cuda')
    K.gru_cell_fused_t(xd, hd, wi, bi, wh, bh, out2, out_t, R)
    assert torch.equal(out2, out) and torch.equal(out_t[:, :R], out[:R].t()) and (out_t[:, R:] == 7.0).all()
    K.gru_cell_fused_t(xs[:, :E], hs[:, :H], wi, None, wh, None, out2, out_t, R)
    assert torch.equal(out2[:R], outs[:, :H]) and torch.equal(out_t[:, :R], outs[:, :H].t())
    from tst_amd import _lib
    with pytest.raises(_lib.TstError, match='must not overlap'):
        K.gru_cell_fused(xd, hd, wi, bi, wh, bh, hd, R)
    assert not K.gru_cell_fused_ok(1024, 1000) and not K.gru_cell_fused_ok(1028, 1024) and K.gru_cell_fused_ok(1024, 1024)


@pytest.mark.parametrize('M,N,Kd', [(300, 700, 64), (2049, 1000, 256), (128, 256, 16), (5, 32000, 256), (1000, 515, 512), (260, 520, 1024)])
@pytest.mark.parametrize('variant', [0, 2, 3, 4, 5])
def test_gemm_dma_k_major_activations_bit_exact(K, M, N, Kd, variant):
    """Both operands staged by LDS-DMA (activations given k-major): the same fma chain per element as the register-staged
    kernel and the oracle, so ``==``; columns of the k-major operand beyond M hold garbage on purpose."""
    r = rs(M + N + Kd)
    a = r.standard_normal((M, Kd)).astype(np.float32)
    w = (r.standard_normal((Kd, N)) / np.sqrt(Kd)).astype(np.float32)
    b = r.standard_normal(N).astype(np.float32)
    ldw = (N + 3) // 4 * 4
    wd = torch.zeros((Kd, ldw), device='cuda'); wd[:, :N] = dev(w)
    ld = (M + 255) // 256 * 256 if variant == 5 else (M + 127) // 128 * 128
    at = torch.full((Kd, ld), 1e30, device='cuda')
    at[:, :M] = dev(a).t()
    out = torch.full((M + 1, ldw + 4), 7.0, device='cuda')
    K.gemm_dma(at, wd[:, :N], dev(b), out, M, variant=variant)
    assert np.array_equal(out[:M, :N].cpu().numpy(), C.gemm(a, w, b))
    assert (out[M:] == 7.0).all() and (out[:, N:] == 7.0).all()


@pytest.mark.parametrize('M,N,Kd', [(2049, 32000, 256), (15676, 32000, 256), (700, 31999 + 1, 1024), (4100, 5000, 64)])
def test_gemm_dma_persistent_form_walks_many_tiles(K, M, N, Kd):
    """The persistent variant when every workgroup walks several tiles (full and partial ones, the ring of stages running on
    across them, a tile's stores in flight under the next tile): bit for bit the one-tile-per-workgroup kernel, which is
    pinned to the oracle above; the bytes around C stay untouched."""
    g = torch.Generator(device='cuda').manual_seed(M + N)
    a = torch.randn((M, Kd), device='cuda', generator=g)
    w = torch.randn((Kd, N), device='cuda', generator=g) / Kd ** 0.5
    b = torch.randn(N, device='cuda', generator=g)
    ld = (M + 127) // 128 * 128
    at = torch.full((Kd, ld), 1e30, device='cuda')
    at[:, :M] = a.t()
    ref = torch.full((M + 1, N + 4), 7.0, device='cuda')
    out = torch.full((M + 1, N + 4), 7.0, device='cuda')
    K.gemm_dma(at, w, b, ref, M, n_cols=N, variant=3)
    for _ in range(2):
        K.gemm_dma(at, w, b, out, M, n_cols=N, variant=4)
    assert torch.equal(out, ref)
    K.gemm_dma(at, w, None, ref, M, n_cols=N, variant=3)
    K.gemm_dma(at, w, None, out, M, n_cols=N, variant=4)
    assert torch.equal(out, ref)


@pytest.mark.parametrize('R,d', [(130, 256), (64, 512), (1, 32), (1000, 64)])
def test_layernorm_k_major_output_equals_layernorm(K, R, d):
    r = rs(R + d)
    x = (r.standard_normal((R, d)) * 3).astype(np.float32)
    g, b = r.standard_normal(d).astype(np.float32), r.standard_normal(d).astype(np.float32)
    y = torch.empty((R, d), device='cuda')
    K.layernorm(dev(x), dev(g), dev(b), y, R)
    yt = torch.full((d, R + 9), 7.0, device='cuda')
    K.layernorm_t(dev(x), dev(g), dev(b), yt, R)
    assert torch.equal(yt[:, :R], y.t()) and (yt[:, R:] == 7.0).all()
    assert np.array_equal(y.cpu().numpy(), C.layernorm(x, g, b))


@pytest.mark.parametrize('B,S', [(7, 12), (300, 42), (1, 70), (1030, 5)])
def test_pack_sources_and_scatter_rows(K, B, S):
    r = rs(B + S)
    src = r.randint(2, 50, size=(B, S)).astype(np.int64)
    lens = r.randint(0, S + 1, size=B)
    src[np.arange(S)[None] >= lens[:, None]] = 1                     # right padding with id 1, some sentences empty
    if B > 3:
        src[3, 0] = 1                                                 # and a hole in front
    p = K.PackedSources(dev(src), 1)
    live = src != 1
    idx = np.flatnonzero(live.reshape(-1))
    assert p.n == len(idx) and np.array_equal(p.mask.cpu().numpy(), live.astype(np.uint8))
    assert np.array_equal(p.padded_row.cpu().numpy()[:p.n], idx)
    assert np.array_equal(p.tok.cpu().numpy()[:p.n], src.reshape(-1)[idx]) and np.array_equal(p.pos.cpu().numpy()[:p.n], idx % S)
    assert np.array_equal(p.sent.cpu().numpy()[:p.n], idx // S)
    kvrow = p.kvrow.cpu().numpy()
    assert np.array_equal(kvrow[live], np.arange(p.n)) and (kvrow[~live] == -1).all()
    rows = torch.randn((max(p.n, 1), 8), device='cuda')
    out = torch.zeros((B * S, 8), device='cuda')
    K.scatter_rows(rows, p.padded_row, out, p.n)
    want = np.zeros((B * S, 8), np.float32)
    want[idx] = rows.cpu().numpy()[:p.n]
    assert np.array_equal(out.cpu().numpy(), want)
