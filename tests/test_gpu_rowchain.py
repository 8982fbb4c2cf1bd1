"""GPU: the row-tile chain kernel (residual product -> LayerNorm -> product in one launch) through the C ABI, bit for bit
against the canonical oracle's three separate operations and against the three separate entry points."""
import numpy as np
import pytest
import torch

from oracle import canon as C

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def K():
    import tst_amd  # noqa: F401
    from tst_amd import kernels
    assert torch.cuda.is_available()
    return kernels


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _inputs(R, K1, N2, seed):
    r = np.random.RandomState(seed)
    d = 256
    A = r.standard_normal((R, K1)).astype(np.float32)
    W1 = (r.standard_normal((K1, d)) / np.sqrt(K1)).astype(np.float32)
    b1 = r.standard_normal(d).astype(np.float32)
    X = (r.standard_normal((R, d)) * 2 + 0.3).astype(np.float32)
    g, b = r.standard_normal(d).astype(np.float32), r.standard_normal(d).astype(np.float32)
    W2 = (r.standard_normal((d, max(N2, 4))) / 16).astype(np.float32)
    b2 = r.standard_normal(max(N2, 4)).astype(np.float32)
    return A, W1, b1, X, g, b, W2, b2


@pytest.mark.parametrize('R,K1,N2,act2', [
    (1, 256, 256, 0), (63, 256, 1024, 1), (64, 1024, 768, 0), (65, 256, 256, 1), (200, 1024, 0, 0), (1000, 256, 1024, 1),
    (1000, 1024, 768, 0), (777, 256, 0, 0), (5000, 256, 256, 0), (4133, 1024, 1024, 1), (2049, 512, 768, 0)])
def test_chain_bit_exact_against_oracle(K, R, K1, N2, act2):
    A, W1, b1, X, g, b, W2, b2 = _inputs(R, K1, N2, R + K1 + N2)
    x = dev(X)
    guard = 7.0
    if N2:
        ldc = N2 + 4                        # a row stride beyond the columns: nothing may be written there
        out = torch.full((R + 3, ldc), guard, dtype=torch.float32, device='cuda')
        K.chain(dev(A), dev(W1), dev(b1), x, dev(g), dev(b), R, w2t=dev(W2), b2=dev(b2), out=out, n_cols=N2, act2=act2)
    else:
        ldt = (R + 127) // 128 * 128
        out_t = torch.full((256, ldt), guard, dtype=torch.float32, device='cuda')
        K.chain(dev(A), dev(W1), dev(b1), x, dev(g), dev(b), R, out_t=out_t)
    want_x = C.gemm(A, W1, b1, X, 0)
    assert np.array_equal(x.cpu().numpy(), want_x)
    want_y = C.layernorm(want_x, g, b)
    if N2:
        want = C.gemm(want_y, W2, b2, None, act2)
        got = out.cpu().numpy()
        assert np.array_equal(got[:R, :N2], want)
        assert (got[:R, N2:] == guard).all() and (got[R:] == guard).all()
    else:
        got = out_t.cpu().numpy()
        assert np.array_equal(got[:, :R], want_y.T)
        assert (got[:, R:] == guard).all()


@pytest.mark.parametrize('R,K1,N2,act2', [(9000, 256, 256, 0), (20000, 256, 1024, 1), (16384, 1024, 768, 0), (12345, 1024, 0, 0)])
def test_chain_equals_the_three_entry_points_at_decode_sizes(K, R, K1, N2, act2):
    """Thousands of rows (several workgroups per XCD, a partial last tile): the chain against tst_gemm_mfma_f32 with a
    residual, tst_layernorm_f32 / tst_layernorm_t_f32 and tst_gemm_mfma_f32, which are themselves pinned to the oracle."""
    A, W1, b1, X, g, b, W2, b2 = _inputs(R, K1, N2, 11)
    a, w1, bb1, gg, be, w2, bb2 = dev(A), dev(W1), dev(b1), dev(g), dev(b), dev(W2), dev(b2)
    x_ref = dev(X)
    K.gemm(a, w1, bb1, x_ref, R, res=x_ref)
    x = dev(X)
    if N2:
        h = torch.empty((R, 256), dtype=torch.float32, device='cuda')
        ref = torch.empty((R, N2), dtype=torch.float32, device='cuda')
        K.layernorm(x_ref, gg, be, h, R)
        K.gemm(h, w2, bb2, ref, R, n_cols=N2, act=act2)
        out = torch.empty((R, N2), dtype=torch.float32, device='cuda')
        K.chain(a, w1, bb1, x, gg, be, R, w2t=w2, b2=bb2, out=out, n_cols=N2, act2=act2)
    else:
        ldt = (R + 127) // 128 * 128
        ref = torch.zeros((256, ldt), dtype=torch.float32, device='cuda')
        out = torch.zeros((256, ldt), dtype=torch.float32, device='cuda')
        K.layernorm_t(x_ref, gg, be, ref, R)
        K.chain(a, w1, bb1, x, gg, be, R, out_t=out)
    assert torch.equal(x, x_ref)
    assert torch.equal(out, ref)


def test_chain_rejects_other_shapes(K):
    from tst_amd import _lib
    assert not K.chain_ok(100, 512, 512, 256) and not K.chain_ok(100, 256, 300, 256) and not K.chain_ok(100, 256, 256, 512)
    x = torch.zeros((8, 512), dtype=torch.float32, device='cuda')
    with pytest.raises(_lib.TstError, match='tst_chain_f32: needs d == 256'):
        K.chain(x, torch.zeros((512, 512), device='cuda'), None, x.clone(), torch.ones(512, device='cuda'), torch.zeros(512, device='cuda'), 8,
                w2t=torch.zeros((512, 256), device='cuda'), out=torch.zeros((8, 256), device='cuda'))
