"""GPU: final LayerNorm + vocabulary projection + arg-max as one kernel (``tst_ln_vocab_argmax_f32``) and the greedy decode
loop that uses it, through the C ABI.  The tokens must be those of the canonical oracle's LayerNorm -> product -> first
arg-max, and those of the three separate entry points (``==``): the kernel forms the same fma chains."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import canon as C

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def K():
    import tst_amd
    from tst_amd import kernels
    assert torch.cuda.is_available()
    return kernels


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _padded(W):
    Kd, N = W.shape
    ld = (N + 3) // 4 * 4
    Wd = torch.zeros((Kd, ld), dtype=torch.float32, device='cuda')
    Wd[:, :N] = dev(W)
    return Wd[:, :N]


def _separate_calls(K, x, g, b, Wd, bias, R, N):
    """tokens of tst_layernorm_f32 -> tst_gemm_mfma_f32 -> tst_argmax_rows_f32"""
    d = x.shape[1]
    h = torch.empty((R, d), dtype=torch.float32, device='cuda')
    K.layernorm(x, g, b, h, R)
    ld = (N + 3) // 4 * 4
    logits = torch.zeros((R, ld), dtype=torch.float32, device='cuda')
    K.gemm(h, Wd, bias, logits, R, n_cols=N)
    tok = torch.empty(R, dtype=torch.int32, device='cuda')
    K.argmax_rows(logits, R, N, tok)
    return tok, logits


@pytest.mark.parametrize('R,V,d,with_bias', [(256, 32000, 256, True), (1, 100, 256, True), (37, 1000, 512, True), (130, 4099, 256, False),
                                             (65, 63, 512, True), (200, 20000, 512, True), (64, 64, 256, True)])
def test_fused_kernel_equals_oracle_and_separate_calls(K, R, V, d, with_bias):
    assert K.vocab_argmax_ok(R, V, d)
    r = np.random.RandomState(R + V + d)
    x = (r.standard_normal((R, d)) * 2 + 0.5).astype(np.float32)
    g, b = r.standard_normal(d).astype(np.float32), r.standard_normal(d).astype(np.float32)
    W = (r.standard_normal((d, V)) / 16).astype(np.float32)
    if V >= 100:
        W[:, V - 1] = W[:, 3]                   # equal logits in the first and the last tile: the earlier column wins
        W[:, V // 2 + 1] = W[:, V // 2]         # and next to each other
    bias = r.standard_normal(V).astype(np.float32) if with_bias else None
    if with_bias and V >= 100:
        bias[V - 1] = bias[3]
        bias[V // 2 + 1] = bias[V // 2]
    xd, gd, bd, Wd = dev(x), dev(g), dev(b), _padded(W)
    biasd = None if bias is None else dev(bias)
    keys = torch.zeros((R + 3, K.ARGMAX_SLOTS), dtype=torch.int64, device='cuda')
    keys[R:] = 77                                # rows beyond R are not touched
    K.ln_vocab_argmax(xd, gd, bd, Wd, biasd, keys, R, n_cols=V)
    got = K.tokens_of_keys(keys[:R]).cpu().numpy()
    assert (keys[R:] == 77).all()
    want = C.argmax_rows(C.gemm(C.layernorm(x, g, b), W, bias, None, 0))
    assert np.array_equal(got, want)
    sep, _ = _separate_calls(K, xd, gd, bd, Wd, biasd, R, V)
    assert np.array_equal(got, sep.cpu().numpy())


def test_special_values_follow_argmax_rows(K):
    """Equal largest values: the first index wins; NaN and -inf are never selected; a row without any candidate gives token 0;
    +inf is selected.  Made with LayerNorm parameters that turn every normalised row into zeros, so that logits == bias."""
    R, V, d = 5, 300, 256
    x = np.random.RandomState(3).standard_normal((R, d)).astype(np.float32)
    g = np.zeros(d, np.float32)
    b = np.zeros(d, np.float32)                 # LayerNorm(x) == 0 everywhere: logits == bias (0 * w + bias, fma chains of zeros)
    W = np.ones((d, V), np.float32)
    bias = np.full(V, -1.0, np.float32)
    bias[10], bias[20] = -0.0, 0.0              # all rows: the zeros at 10 and 20 are the largest and equal -> 10
    xd, gd, bd, Wd = dev(x), dev(g), dev(b), _padded(W)
    keys = torch.zeros((R, K.ARGMAX_SLOTS), dtype=torch.int64, device='cuda')
    K.ln_vocab_argmax(xd, gd, bd, Wd, dev(bias), keys, R, n_cols=V)
    sep, logits = _separate_calls(K, xd, gd, bd, Wd, dev(bias), R, V)
    assert np.array_equal(K.tokens_of_keys(keys).cpu().numpy(), sep.cpu().numpy())
    assert (sep.cpu().numpy() == 10).all()
    for special in (np.nan, -np.inf):
        bias2 = np.full(V, special, np.float32)
        keys.zero_()
        K.ln_vocab_argmax(xd, gd, bd, Wd, dev(bias2), keys, R, n_cols=V)
        sep, _ = _separate_calls(K, xd, gd, bd, Wd, dev(bias2), R, V)
        assert (keys == 0).all()
        assert np.array_equal(K.tokens_of_keys(keys).cpu().numpy(), sep.cpu().numpy())
        bias2[150] = -3.0e38                    # one finite value among them
        bias2[7] = np.nan
        keys.zero_()
        K.ln_vocab_argmax(xd, gd, bd, Wd, dev(bias2), keys, R, n_cols=V)
        assert (K.tokens_of_keys(keys).cpu().numpy() == 150).all()
    bias3 = np.full(V, -1.0, np.float32)
    bias3[200] = np.inf
    keys.zero_()
    K.ln_vocab_argmax(xd, gd, bd, Wd, dev(bias3), keys, R, n_cols=V)
    assert (K.tokens_of_keys(keys).cpu().numpy() == 200).all()


def test_shapes_outside_the_kernel_are_refused(K):
    import tst_amd
    assert not K.vocab_argmax_ok(257, 32000, 256)
    assert not K.vocab_argmax_ok(16, 100, 128)
    x = torch.zeros((300, 256), dtype=torch.float32, device='cuda')
    g = torch.ones(256, dtype=torch.float32, device='cuda')
    W = torch.zeros((256, 32000), dtype=torch.float32, device='cuda')
    keys = torch.zeros((300, K.ARGMAX_SLOTS), dtype=torch.int64, device='cuda')
    with pytest.raises(tst_amd.TstError):
        K.ln_vocab_argmax(x, g, g, W, None, keys, 300)


CHILD = r'''
import hashlib, sys, torch
sys.path.insert(0, %r)
import tst_amd
from tst_amd import synthetic
n_sent = int(sys.argv[1])
hp = dict(d_model=256, dropout=0.1, n_layers=2, pe_max_len=60, n_heads=2, d_ff=1024)
V = 4000
vocab = tst_amd.Vocab.synthetic(V)
model = tst_amd.Transformer(tst_amd.HP(hp), vocab, vocab)
model.load_state_dict(synthetic.synthetic_state_dict(model, 3, 4.0, 2.0, seed=7))
src = synthetic.multi30k_like_sources(n_sent, V, seed=99).cuda()
encs, mask = model.encoder(src)
st = tst_amd.InferenceState({'model': model.decoder, 'inputs': encs, 'enc_mask': mask, 'vocab': vocab, 'max_len': 30})
tok, lens, score = st.decode_packed()
print('RESULT %%s %%d' %% (hashlib.sha256(tok.cpu().numpy().tobytes() + lens.cpu().numpy().tobytes()).hexdigest(), int(lens.sum())))
''' % ROOT


def _decode(n_sent, env=None):
    """digest of the tokens of a greedy decode in a child process (the switches are read when the package is imported)"""
    e = dict(os.environ)
    for k in ('TST_NO_ARGMAX_KEYS', 'TST_NO_HOST_COUNT', 'TST_NO_LAG'):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run([sys.executable, '-c', CHILD, str(n_sent)], capture_output=True, text=True, env=e, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    return [l for l in p.stdout.splitlines() if l.startswith('RESULT ')][-1]


@pytest.mark.parametrize('n_sent', [7, 200])
def test_greedy_decode_is_the_same_with_and_without_the_fused_tail(n_sent):
    """The decode loop with arg-max keys and the host-visible survivor count against the loop with logits, a separate
    arg-max and a queued copy of the count: identical token lists."""
    a = _decode(n_sent)
    b = _decode(n_sent, {'TST_NO_ARGMAX_KEYS': '1', 'TST_NO_HOST_COUNT': '1'})
    c = _decode(n_sent, {'TST_NO_LAG': '1'})
    assert a == b
    assert a == c
    assert int(a.split()[-1]) > 2 * n_sent
