"""GPU: the GRU cell step with layer 0's input pre-activations looked up per token in a table (bias_ih + embedding @ W_ih for
every vocabulary entry) instead of computed from the looked-up embeddings, through the C ABI.  The table is made by the same
product kernel, so every result must equal (``==``) the kernels that multiply, the three-call form and the canonical oracle."""
import hashlib
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


@pytest.mark.parametrize('R,E,H,V', [(300, 64, 128, 50), (129, 1024, 1024, 128), (1, 32, 64, 7), (2048, 256, 192, 1000), (4100, 128, 256, 3000)])
def test_cell_from_table_equals_cell_from_embeddings_and_oracle(K, R, E, H, V):
    rs = np.random.RandomState(R + E + H + V)
    emb = rs.standard_normal((V, E)).astype(np.float32)
    tok = rs.randint(0, V, size=R).astype(np.int32)
    h = rs.standard_normal((R, H)).astype(np.float32)
    wih = (rs.standard_normal((E, 3 * H)) / np.sqrt(E)).astype(np.float32)
    whh = (rs.standard_normal((H, 3 * H)) / np.sqrt(H)).astype(np.float32)
    bih, bhh = (0.1 * rs.standard_normal(3 * H)).astype(np.float32), (0.1 * rs.standard_normal(3 * H)).astype(np.float32)
    x = emb[tok]
    want = C.gru_cell(C.gemm(x, wih, bih), C.gemm(h, whh, bhh), h)
    embd, tokd, hd, wi, wh, bi, bh = (torch.from_numpy(a).cuda() for a in (emb, tok, h, wih, whh, bih, bhh))
    tab = K.gi_table(embd, wi, bi)
    assert np.array_equal(tab.cpu().numpy(), C.gemm(emb, wih, bih))            # the table itself is the canonical product
    # one kernel: recurrent product + update, input pre-activations from the table
    out = torch.full((R + 2, H), 7.0, device='cuda')
    K.gru_cell_fused_tab(tab, tokd, hd, wh, bh, out, R)
    assert np.array_equal(out[:R].cpu().numpy(), want) and (out[R:] == 7.0).all()
    ref = torch.empty((R, H), device='cuda')
    K.gru_cell_fused(torch.from_numpy(x).cuda(), hd, wi, bi, wh, bh, ref, R)
    assert torch.equal(out[:R], ref)
    # with the k-major copy of the new state
    ld = (R + 127) // 128 * 128
    out2, out_t = torch.full((R + 2, H), 7.0, device='cuda'), torch.full((H, ld), 7.0, device='cuda')
    K.gru_cell_fused_tab(tab, tokd, hd, wh, bh, out2, R, h_out_t=out_t)
    assert torch.equal(out2, out) and torch.equal(out_t[:, :R], out[:R].t()) and (out_t[:, R:] == 7.0).all()
    # the update alone (three-call form: the recurrent product is a call of its own), in place
    gh = torch.empty((R, 3 * H), device='cuda')
    K.gemm(hd, wh, bh, gh, R)
    h_inplace = hd.clone()
    K.gru_cell_step_tab(tab, tokd, gh, h_inplace, h_inplace, R)
    assert torch.equal(h_inplace, out[:R])
    # a padded table (row stride > 3 H) and no recurrent bias
    tabp = torch.zeros((V, 3 * H + 8), device='cuda')
    tabp[:, :3 * H] = tab
    out3 = torch.empty((R, H), device='cuda')
    K.gru_cell_fused_tab(tabp[:, :3 * H], tokd, hd, wh, None, out3, R)
    assert np.array_equal(out3.cpu().numpy(), C.gru_cell(C.gemm(x, wih, bih), C.gemm(h, whh), h))
    h2 = hd.clone()
    K.gemm(hd, wh, None, gh, R)
    K.gru_cell_step_tab(tabp[:, :3 * H], tokd, gh, h2, h2, R)
    assert torch.equal(h2, out3)


def test_ids_outside_the_vocabulary_are_clamped_like_the_embedding_lookup(K):
    R, E, H, V = 70, 32, 64, 11
    rs = np.random.RandomState(5)
    emb = torch.from_numpy(rs.standard_normal((V, E)).astype(np.float32)).cuda()
    wi = torch.from_numpy((rs.standard_normal((E, 3 * H)) / 6).astype(np.float32)).cuda()
    wh = torch.from_numpy((rs.standard_normal((H, 3 * H)) / 8).astype(np.float32)).cuda()
    h = torch.from_numpy(rs.standard_normal((R, H)).astype(np.float32)).cuda()
    tok = torch.from_numpy(rs.randint(-5, V + 5, size=R).astype(np.int32)).cuda()
    tab = K.gi_table(emb, wi, None)
    x = torch.empty((R, E), device='cuda')
    K.embed_posadd(tok, emb, None, 1.0, x, R)
    ref, out = torch.empty((R, H), device='cuda'), torch.empty((R, H), device='cuda')
    K.gru_cell_fused(x, h, wi, None, wh, None, ref, R)
    K.gru_cell_fused_tab(tab, tok, h, wh, None, out, R)
    assert torch.equal(out, ref)
    from tst_amd import _lib
    with pytest.raises(_lib.TstError, match='must not overlap'):
        K.gru_cell_fused_tab(tab, tok, h, wh, None, h, R)


def test_encoder_with_the_table_equals_embedding_product_and_update(K):
    """RNNEncoder looks the input pre-activations of every position up in its table; the same pass spelled out with the
    embedding kernel, one input product over all positions and the plain update kernel gives the same final states."""
    import tst_amd
    from tst_amd import synthetic
    vocab = tst_amd.Vocab.synthetic(90)
    enc = tst_amd.RNNEncoder(64, 128, vocab)
    enc.load_state_dict(synthetic.synthetic_state_dict(enc, seed=4))
    B, T = 37, 9
    x = torch.randint(0, 90, (B, T), generator=torch.Generator().manual_seed(8)).cuda()
    z = enc(x)
    W = enc.prepared()
    assert W['gi_tab'] is not None
    tok = x.to(torch.int32).t().contiguous().view(-1)
    emb = torch.empty((T * B, 64), device='cuda')
    K.embed_posadd(tok, W['emb'], None, 1.0, emb, T * B)
    gi = torch.empty((T * B, 3 * 128), device='cuda')
    K.gemm(emb, W['wih'], W['bih'], gi, T * B)
    h = torch.zeros((B, 128), device='cuda')
    gh = torch.empty((B, 3 * 128), device='cuda')
    for t in range(T):
        K.gemm(h, W['whh'], W['bhh'], gh, B)
        K.gru_cell_step(gi[t * B:(t + 1) * B], gh, h, h, B)
    assert torch.equal(z, h)


CHILD = r'''
import hashlib, sys, torch
sys.path.insert(0, %r)
import tst_amd
from tst_amd import synthetic
n_sent, beam, layers = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
vocab = tst_amd.Vocab.synthetic(300, extra=('|',))
if layers == 1:
    model = tst_amd.RNNDecoder(128, 256, vocab)
    z = torch.randn((n_sent, 256), generator=torch.Generator().manual_seed(3)).cuda()
else:
    model = tst_amd.RNNLM(128, vocab, n_layers=layers)
    z = torch.randn((layers, n_sent, 128), generator=torch.Generator().manual_seed(3)).cuda()
model.load_state_dict(synthetic.synthetic_state_dict(model, vocab.stoi['|'], 1.0, 2.0, seed=9))
cfg = {'model': model, 'inputs': z, 'vocab': vocab, 'max_len': 20, 'eos_token': '|', 'beam_size': beam}
if layers > 1:
    cfg['inputs_batch_dim'] = 1
st = tst_amd.InferenceState(cfg)
tok, lens, score = st.decode_packed()
print('RESULT %%s %%d' %% (hashlib.sha256(tok.cpu().numpy().tobytes() + lens.cpu().numpy().tobytes() + score.cpu().numpy().tobytes()).hexdigest(),
                          int(lens.sum())))
''' % ROOT


def _decode(n_sent, beam, layers, env=None):
    e = dict(os.environ)
    e.pop('TST_NO_GI_TABLE', None)
    e.update(env or {})
    p = subprocess.run([sys.executable, '-c', CHILD, str(n_sent), str(beam), str(layers)], capture_output=True, text=True, env=e, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    return [l for l in p.stdout.splitlines() if l.startswith('RESULT ')][-1]


@pytest.mark.parametrize('n_sent,beam,layers', [(100, 1, 1), (3000, 1, 1), (700, 4, 1), (150, 1, 2), (2500, 1, 2)])
def test_decode_with_and_without_the_table_gives_the_same_output(n_sent, beam, layers):
    """Greedy and beam decoding of one- and two-layer GRU models, small batches (three-call cell) and large ones (one-kernel
    cell), with the per-token table and with TST_NO_GI_TABLE=1: identical tokens, lengths and scores."""
    a = _decode(n_sent, beam, layers)
    b = _decode(n_sent, beam, layers, {'TST_NO_GI_TABLE': '1'})
    assert a == b
    assert int(a.split()[-1]) > n_sent
