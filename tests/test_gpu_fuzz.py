"""Seeded random configurations of the decode paths (transformer from cached encoder outputs and from source ids, GRU decoder,
transformer LM, stateful GRU LMs; model sizes, head counts, vocabulary, batch, source lengths, beam, length penalty, max_len,
sampling temperature) against the canonical oracle, bit for bit: tokens, hypothesis scores and output lengths.
The cases are drawn to reach what the fixed-size tests do not: one-sentence and one-position batches, heads of 8 to 128
elements, beams wider than the vocabulary's live mass, max_len of 1, sentences that never end, batches that all end at once,
row counts on both sides of the loop's small-batch switches."""
import os

import numpy as np
import pytest
import torch

from oracle import canon as C
from oracle import torch_ref as R

pytestmark = pytest.mark.gpu

# TST_FUZZ_FIRST / TST_FUZZ_SCALE move and widen the seed ranges for a longer sweep than the suite's (same draws per seed)
FIRST, SCALE = int(os.environ.get('TST_FUZZ_FIRST', '0')), int(os.environ.get('TST_FUZZ_SCALE', '1'))


def seeds(n):
    return range(FIRST, FIRST + n * SCALE)


@pytest.fixture(scope='module')
def T():
    import tst_amd
    assert torch.cuda.is_available()
    return tst_amd


def draw(seed):
    r = np.random.RandomState(1000 + seed)
    heads = int(r.choice([1, 2, 4, 8]))
    dh = int(r.choice([8, 16, 32, 64, 128]))
    while heads * dh > 512:
        dh //= 2
    d = heads * dh
    return dict(hp=R.mt_hp(d_model=d, n_heads=heads, n_layers=int(r.choice([1, 2, 3])), d_ff=int(r.choice([16, 64, 256, 1024]))),
                V=int(r.choice([5, 17, 64, 300, 1000, 5000])), B=int(r.choice([1, 2, 7, 33, 100, 300])), S=int(r.choice([1, 2, 9, 23])),
                max_len=int(r.choice([1, 2, 7, 19, 40])), beam=int(r.choice([1, 1, 2, 3, 4, 8])), alpha=float(r.choice([0.0, 0.0, 0.6, 1.0])),
                eos_bias=float(r.choice([-2.0, 0.5, 2.0, 4.0, 9.0])))


@pytest.mark.parametrize('seed', seeds(28))
def test_random_configuration_bit_exact(T, seed):
    c = draw(seed)
    hp, V, B, S, max_len, beam, alpha = c['hp'], c['V'], c['B'], c['S'], c['max_len'], c['beam'], c['alpha']
    tv = R.synthetic_vocab(V)
    tmodel = R.synthetic_init(R.Transformer(R.HP(hp), tv, tv), 3, c['eos_bias'], 2.0, seed=seed)
    vocab = T.Vocab.synthetic(V)
    model = T.Transformer(T.HP(hp), vocab, vocab)
    model.load_state_dict(tmodel.state_dict())
    r = np.random.RandomState(seed)
    lens = r.randint(1, S + 1, size=B)
    encs = r.standard_normal((B, S, hp['d_model'])).astype(np.float32)
    mask = (np.arange(S)[None] < lens[:, None])[:, None, :]
    st = T.InferenceState({'model': model.decoder, 'inputs': torch.from_numpy(encs).cuda(), 'enc_mask': torch.from_numpy(mask).cuda(),
                           'vocab': vocab, 'max_len': max_len, 'beam_size': beam, 'length_penalty': alpha})
    if beam > V:                                    # a beam cannot hold more distinct first tokens than the vocabulary has
        with pytest.raises(ValueError, match='larger than the vocabulary'):
            st.inference()
        return
    out = [[int(t) for t in s.tolist()] for s in st.inference()]
    sess = C.CanonTransformerDecoder(tmodel.decoder, hp['n_heads']).session(encs, mask, beam, max_len)
    if beam == 1:
        want, _, _ = C.greedy_decode(sess, B, max_len, 2, 3, 0)
    else:
        want, scores, _, _ = C.beam_decode(sess, B, beam, max_len, 2, 3, alpha, 0)
        assert np.array_equal(st.scores.cpu().numpy(), scores), c
    assert out == want, c
    assert all(1 <= len(s) <= 1 + max_len and s[0] == 2 for s in out)


@pytest.mark.parametrize('seed', seeds(10))
def test_random_whole_path_from_source_ids(T, seed):
    """Source ids with ragged padding -> encoder pass -> decode, as transformer_mt_trainer.py:91-98 calls it: encoder output at
    every real position, tokens and scores, bit for bit."""
    c = draw(100 + seed)
    hp, V, B, S, max_len, beam, alpha = c['hp'], max(c['V'], 17), c['B'], max(c['S'], 2), max(c['max_len'], 2), c['beam'], c['alpha']
    tv = R.synthetic_vocab(V)
    tmodel = R.synthetic_init(R.Transformer(R.HP(hp), tv, tv), 3, c['eos_bias'], 2.0, seed=seed)
    vocab = T.Vocab.synthetic(V)
    model = T.Transformer(T.HP(hp), vocab, vocab)
    model.load_state_dict(tmodel.state_dict())
    r = np.random.RandomState(seed)
    lens = r.randint(1, S + 1, size=B)
    lens[r.randint(B)] = S                                     # one sentence fills the padded width
    src = np.where(np.arange(S)[None] < lens[:, None], r.randint(4, V, size=(B, S)), 1).astype(np.int64)
    encs, enc_mask = model.encoder(torch.from_numpy(src).cuda())
    want_encs, want_mask = C.CanonTransformerEncoder(tmodel.encoder, hp['n_heads'])(src)
    real = want_mask[:, 0, :]
    assert np.array_equal(encs.cpu().numpy()[real], want_encs[real])
    st = T.InferenceState({'model': model.decoder, 'inputs': encs, 'enc_mask': enc_mask, 'vocab': vocab, 'max_len': max_len,
                           'beam_size': beam, 'length_penalty': alpha})
    out = [[int(t) for t in s.tolist()] for s in st.inference()]
    sess = C.CanonTransformerDecoder(tmodel.decoder, hp['n_heads']).session(want_encs, want_mask, beam, max_len)
    if beam == 1:
        want, _, _ = C.greedy_decode(sess, B, max_len, 2, 3, 0)
    else:
        want, scores, _, _ = C.beam_decode(sess, B, beam, max_len, 2, 3, alpha, 0)
        assert np.array_equal(st.scores.cpu().numpy(), scores), c
    assert out == want, c


@pytest.mark.parametrize('seed', seeds(10))
def test_random_gru_decoder(T, seed):
    """GRU decoder (RNNDecoder(emb, hid, vocab), cyclegan_trainer.py:64) at random sizes, greedy and beam."""
    r = np.random.RandomState(2000 + seed)
    E, H = int(r.choice([16, 64, 96, 256])), int(r.choice([32, 64, 192, 512]))
    V, B = int(r.choice([9, 40, 128, 700])), int(r.choice([1, 3, 40, 200, 2100]))
    max_len, beam = int(r.choice([1, 5, 30])), int(r.choice([1, 1, 2, 4]))
    tv = R.synthetic_vocab(V)
    tmodel = R.synthetic_init(R.RNNDecoder(E, H, tv), 3, float(r.choice([0.0, 1.5, 3.0])), 2.0, seed=seed)
    vocab = T.Vocab.synthetic(V)
    model = T.RNNDecoder(E, H, vocab)
    model.load_state_dict(tmodel.state_dict())
    z = r.standard_normal((B, H)).astype(np.float32)
    st = T.InferenceState({'model': model, 'inputs': torch.from_numpy(z).cuda(), 'vocab': vocab, 'max_len': max_len, 'beam_size': beam})
    out = [[int(t) for t in s.tolist()] for s in st.inference()]
    sess = C.CanonGRUDecoder(tmodel).session(z, beam)
    if beam == 1:
        want, _, _ = C.greedy_decode(sess, B, max_len, 2, 3, 0)
    else:
        want, scores, _, _ = C.beam_decode(sess, B, beam, max_len, 2, 3, 0.0, 0)
        assert np.array_equal(st.scores.cpu().numpy(), scores)
    assert out == want, (E, H, V, B, max_len, beam)


@pytest.mark.parametrize('seed', seeds(24))
def test_random_gemm_shapes_bit_exact(seed):
    """tst_gemm_mfma_f32 at random shapes across every kernel the dispatcher can pick (strip kernels, 64 x 64 x 16, 64 x 64 x 32,
    128 x 128, 128 x 256 tiles), with and without bias / relu / residual: equal to the k-ordered fma chain of the oracle."""
    from tst_amd import kernels as K
    r = np.random.RandomState(3000 + seed)
    M = int(r.choice([1, 3, 17, 100, 257, 1000, 2100, 4100, 9000]))
    N = int(r.choice([1, 8, 33, 90, 256, 520, 1000, 2304]))
    Kd = int(r.choice([4, 20, 36, 64, 132, 256, 512, 1024]))
    while M * N * Kd > 5e8:
        M = max(1, M // 2)
    bias, act, res = bool(r.randint(2)), int(r.randint(2)), bool(r.randint(2))
    A = r.standard_normal((M, Kd)).astype(np.float32)
    W = r.standard_normal((Kd, N)).astype(np.float32)
    b = r.standard_normal(N).astype(np.float32) if bias else None
    R0 = r.standard_normal((M, N)).astype(np.float32) if res else None
    ldw = (N + 3) // 4 * 4
    Wd = torch.zeros((Kd, ldw), dtype=torch.float32, device='cuda')
    Wd[:, :N] = torch.from_numpy(W).cuda()
    out = torch.full((M, ldw), 7.0, dtype=torch.float32, device='cuda')
    K.gemm(torch.from_numpy(A).cuda(), Wd[:, :N], None if b is None else torch.from_numpy(b).cuda(), out, M, n_cols=N,
           res=None if R0 is None else torch.from_numpy(R0).cuda(), act=act)
    got = out.cpu().numpy()
    assert np.array_equal(got[:, :N], C.gemm(A, W, b, R0, act)), (M, N, Kd, bias, act, res)
    assert (got[:, N:] == 7.0).all()


@pytest.mark.parametrize('seed', seeds(10))
def test_random_transformer_lm(T, seed):
    """z-conditioned TransformerLM (lm_trainer.py:72-74,86) in fp32 at random sizes, greedy and beam with and without a
    length penalty: tokens and scores bit for bit."""
    r = np.random.RandomState(4000 + seed)
    heads = int(r.choice([1, 2, 4, 8]))
    d = heads * int(r.choice([8, 16, 32, 64]))
    max_len = int(r.choice([1, 3, 12, 30]))
    hp = dict(model_size=d, n_layers=int(r.choice([1, 2, 4])), n_heads=heads, d_ff=int(r.choice([32, 128, 512])), pe_max_len=max_len + 2)
    V, B = int(r.choice([12, 90, 700, 3000])), int(r.choice([1, 4, 50, 260]))
    beam, alpha = int(r.choice([1, 1, 3, 8])), float(r.choice([0.0, 0.7]))
    tv = R.synthetic_vocab(V)
    tmodel = R.synthetic_init(R.TransformerLM(R.HP(hp), tv), 3, float(r.choice([0.0, 2.0, 5.0])), 2.0, seed=seed)
    vocab = T.Vocab.synthetic(V)
    lm = T.TransformerLM(T.HP(hp), vocab, dtype='fp32')
    lm.load_state_dict(tmodel.state_dict())
    z = (0.3 * r.standard_normal((B, d))).astype(np.float32)
    st = T.InferenceState({'model': lm, 'inputs': torch.from_numpy(z).cuda(), 'vocab': vocab, 'max_len': max_len,
                           'beam_size': beam, 'length_penalty': alpha})
    out = [s.cpu().tolist() for s in st.inference()]
    sess = C.CanonTransformerLM(tmodel).session(z, beam, max_len)
    if beam == 1:
        want = C.greedy_decode(sess, B, max_len, 2, 3)[0]
    else:
        want, scores, _, _ = C.beam_decode(sess, B, beam, max_len, 2, 3, alpha)
        assert np.array_equal(st.scores.cpu().numpy(), scores), (hp, V, B, beam, alpha)
    assert out == want, (hp, V, B, beam, alpha)


@pytest.mark.parametrize('seed', seeds(12))
def test_random_stateful_gru_lm(T, seed):
    """RNNLM / ConditionalLM of 1 to 3 layers in the stateful mode of prod/dialog_model.py:108-120 (bos = eos = '|', the carried
    state substituted for the inputs) at random sizes: arg-max, temperature sampling (draw for draw) and beam search; a
    ConditionalLM with one style per sentence."""
    r = np.random.RandomState(5000 + seed)
    H, layers = int(r.choice([32, 64, 128, 320])), int(r.choice([1, 2, 3]))
    V, B = int(r.choice([10, 50, 130, 900])), int(r.choice([1, 5, 70, 300, 2200]))
    max_len = int(r.choice([1, 6, 25]))
    mode = str(r.choice(['max', 'sample', 'beam']))
    conditional = bool(r.randint(2)) and layers == 2
    tv = R.synthetic_vocab(V, extra=('|',))
    bias = float(r.choice([0.0, 1.0, 2.5]))
    tmodel = R.synthetic_init(R.ConditionalLM(H, tv) if conditional else R.RNNLM(H, tv, layers), tv.stoi['|'], bias, 2.0, seed=seed)
    vocab = T.Vocab.synthetic(V, extra=('|',))
    lm = T.ConditionalLM(H, vocab) if conditional else T.RNNLM(H, vocab, layers)
    lm.load_state_dict(tmodel.state_dict())
    z = r.standard_normal((layers, B, H)).astype(np.float32)
    kwargs = {'return_z': True}
    styles = None
    if conditional:
        styles = r.randint(0, 2, size=B)
        kwargs['style'] = torch.from_numpy(styles)
    d = {'model': lm, 'inputs': torch.from_numpy(z).cuda(), 'vocab': vocab, 'max_len': max_len, 'bos_token': '|', 'eos_token': '|',
         'substitute_inputs': True, 'inputs_batch_dim': 1, 'kwargs': kwargs}
    bar = vocab.stoi['|']
    dec = C.CanonGRUDecoder(tmodel)
    what = (H, layers, V, B, max_len, mode, conditional)
    if mode == 'beam':
        k = int(r.choice([2, 4, 7]))
        d['beam_size'] = k
        st = T.InferenceState(d)
        out = [s.cpu().tolist() for s in st.inference()]
        want, scores, _, _ = C.beam_decode(dec.session(z, k, style=styles), B, k, max_len, bar, bar, 0.0)
        assert np.array_equal(st.scores.cpu().numpy(), scores), what
    elif mode == 'sample':
        temp, sd = float(r.choice([0.5, 1.0, 1.6])), int(r.randint(1000))
        d.update(sample_type='sample', temperature=temp, seed=sd)
        out = [s.cpu().tolist() for s in T.InferenceState(d).inference()]
        want = C.greedy_decode(dec.session(z, style=styles), B, max_len, bar, bar, sample=(temp, sd))[0]
    else:
        out = [s.cpu().tolist() for s in T.InferenceState(d).inference()]
        want = C.greedy_decode(dec.session(z, style=styles), B, max_len, bar, bar)[0]
    assert out == want, what
