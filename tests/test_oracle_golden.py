"""CPU: pins the oracle.  (1) The torch restatement of the reference decode loop and (2) the
canonical-arithmetic C oracle must both reproduce the golden fixtures, which were produced by the real
``/root/reference/src/inference.py`` (see tests/golden/make_golden.py): tokens exactly, per-step logits
within the fp32 tolerance of this build (|a-b| <= 1e-4 |b| + 1e-5)."""
import numpy as np
import pytest
import torch

from helpers import load_fixture, logits_close, replay_rand, torch_model_for
from oracle import canon as C
from oracle import torch_ref as R

GREEDY = ['tfm_v64', 'tfm_v1000', 'gru_h32', 'gru_h1024', 'rnnlm_stateful', 'tlm_d64']
KEEP = 6


def run_torch_restatement(fx, model, vocab):
    cfg = fx['config']
    if cfg['kind'] == 'transformer':
        st = R.RefInferenceState({'model': model.decoder, 'inputs': torch.from_numpy(fx['encs']),
                                  'enc_mask': torch.from_numpy(fx['enc_mask']), 'vocab': vocab, 'max_len': cfg['max_len']})
    elif cfg['kind'] == 'gru':
        st = R.RefInferenceState({'model': model, 'inputs': torch.from_numpy(fx['z']), 'vocab': vocab,
                                  'max_len': cfg['max_len'], 'eos_token': cfg['eos_token']})
    elif cfg['kind'] == 'tlm':
        st = R.RefInferenceState({'model': model, 'inputs': torch.from_numpy(fx['z']), 'vocab': vocab, 'max_len': cfg['max_len']})
    else:
        st = R.RefInferenceState({'model': model, 'inputs': torch.from_numpy(fx['z']), 'vocab': vocab,
                                  'max_len': cfg['max_len'], 'bos_token': '|', 'eos_token': '|', 'substitute_inputs': True,
                                  'inputs_batch_dim': 1, 'kwargs': {'return_z': True}})
    seqs = [s.tolist() for s in st.inference(keep_logits=KEEP)]
    return seqs, [l.numpy() for l in st.step_logits], [r.numpy() for r in st.step_rows]


def run_canon(fx, model, vocab):
    cfg = fx['config']
    if cfg['kind'] == 'transformer':
        sess = C.CanonTransformerDecoder(model.decoder, cfg['hp']['n_heads']).session(fx['encs'], fx['enc_mask'], 1, cfg['max_len'])
        return C.greedy_decode(sess, fx['encs'].shape[0], cfg['max_len'], vocab.stoi['<bos>'], vocab.stoi['<eos>'], KEEP)
    if cfg['kind'] == 'tlm':
        sess = C.CanonTransformerLM(model).session(fx['z'], 1, cfg['max_len'])
        return C.greedy_decode(sess, fx['z'].shape[0], cfg['max_len'], vocab.stoi['<bos>'], vocab.stoi['<eos>'], KEEP)
    sess = C.CanonGRUDecoder(model).session(fx['z'])
    if cfg['kind'] == 'gru':
        return C.greedy_decode(sess, fx['z'].shape[0], cfg['max_len'], vocab.stoi['<bos>'], vocab.stoi[cfg['eos_token']], KEEP)
    return C.greedy_decode(sess, fx['z'].shape[1], cfg['max_len'], vocab.stoi['|'], vocab.stoi['|'], KEEP)


@pytest.mark.parametrize('name', GREEDY)
@pytest.mark.parametrize('which', ['torch_restatement', 'canonical_oracle'])
def test_greedy_matches_reference_fixture(name, which):
    fx = load_fixture(name)
    model, vocab = torch_model_for(fx['config'])
    seqs, logits, rows = (run_torch_restatement if which == 'torch_restatement' else run_canon)(fx, model, vocab)
    assert seqs == fx['seqs']
    n_kept = sum(1 for k in fx if k.startswith('logits_s'))
    assert n_kept > 0 and len(logits) >= n_kept
    for i in range(n_kept):
        assert np.array_equal(rows[i], fx['rows_s%d' % i])
        ok, err = logits_close(logits[i], fx['logits_s%d' % i])
        assert ok, (name, which, i, err)


@pytest.mark.parametrize('name', ['beam_tfm_v64_k4', 'beam_tfm_v64_k3a'])
def test_canonical_beam_matches_torch_beam_fixture(name):
    """Beam search is build-defined (the reference has none); the fixture comes from the torch statement
    of the specification, the canonical oracle must agree with it."""
    fx = load_fixture(name)
    cfg = fx['config']
    model, vocab = torch_model_for(cfg)
    dec = C.CanonTransformerDecoder(model.decoder, cfg['hp']['n_heads'])
    B = fx['encs'].shape[0]
    seqs, scores, _, _ = C.beam_decode(dec.session(fx['encs'], fx['enc_mask'], cfg['beam'], cfg['max_len']), B, cfg['beam'],
                                       cfg['max_len'], vocab.stoi['<bos>'], vocab.stoi['<eos>'], cfg['alpha'])
    assert seqs == fx['seqs']
    assert np.allclose(scores, fx['scores'], rtol=1e-5, atol=1e-4)


def test_beam_of_one_is_greedy():
    fx = load_fixture('tfm_v64')
    cfg = fx['config']
    model, vocab = torch_model_for(cfg)
    dec = C.CanonTransformerDecoder(model.decoder, cfg['hp']['n_heads'])
    B = fx['encs'].shape[0]
    seqs, _, _, _ = C.beam_decode(dec.session(fx['encs'], fx['enc_mask'], 1, cfg['max_len']), B, 1, cfg['max_len'],
                                  vocab.stoi['<bos>'], vocab.stoi['<eos>'], 0.0)
    assert seqs == fx['seqs']


def test_canonical_functions_close_to_libm():
    x = np.linspace(-30, 30, 20001).astype(np.float32)
    assert np.allclose(C.fmap(0, x), np.exp(x.astype(np.float64)), rtol=2e-7)
    assert np.allclose(C.fmap(2, x), 1 / (1 + np.exp(-x.astype(np.float64))), rtol=4e-7, atol=1e-38)
    assert np.allclose(C.fmap(3, x), np.tanh(x.astype(np.float64)), rtol=0, atol=2e-7)
    y = np.exp(np.linspace(-20, 20, 20001)).astype(np.float32)
    assert np.allclose(C.fmap(1, y), np.log(y.astype(np.float64)), rtol=3e-7, atol=2e-7)


def test_canonical_encoder_close_to_torch():
    fx = load_fixture('tfm_v1000')
    model, vocab = torch_model_for(fx['config'])
    encs, mask = C.CanonTransformerEncoder(model.encoder, fx['config']['hp']['n_heads'])(fx['src'])
    assert np.array_equal(mask, fx['enc_mask'])
    valid = fx['enc_mask'][:, 0, :]
    ok, err = logits_close(encs[valid], fx['encs'][valid])
    assert ok, err


def test_sampler_follows_the_categorical_distribution():
    """``sample_type='sample'`` (/root/reference/src/inference.py:147) has no fixed output; the oracle's
    Gumbel-max sampler is pinned to it in distribution: chi-square of 200k draws against softmax(x / T)."""
    rs = np.random.RandomState(3)
    V, n = 12, 200000
    x = (rs.standard_normal(V) * 1.5).astype(np.float32)
    for temp in (1.0, 0.5):
        p = np.exp(x.astype(np.float64) / temp)
        p /= p.sum()
        draws = C.sample_rows(np.tile(x, (n, 1)), temp, seed=11, step=4, sent=np.arange(n))
        counts = np.bincount(draws, minlength=V)
        chi2 = ((counts - n * p) ** 2 / (n * p)).sum()
        assert chi2 < 45.0, (temp, chi2)      # 11 degrees of freedom: P(chi2 > 45) < 1e-5
    a = C.sample_rows(np.tile(x, (50, 1)), 1.0, seed=1, step=0, sent=np.arange(50))
    b = C.sample_rows(np.tile(x, (50, 1))[::-1].copy(), 1.0, seed=1, step=0, sent=np.arange(50)[::-1].copy())
    assert np.array_equal(a, b[::-1])         # a sentence's draw does not depend on its row in the batch
    assert not np.array_equal(a, C.sample_rows(np.tile(x, (50, 1)), 1.0, seed=2, step=0, sent=np.arange(50)))


def test_gumbel_op_restatement_matches_reference_function():
    """``oracle.torch_ref.ref_gumbel_softmax`` against outputs of the reference's own ``gumbel_softmax_sample``."""
    fx = load_fixture('gumbel_op')
    for i in range(3):
        y = R.ref_gumbel_softmax(torch.softmax(torch.from_numpy(fx['logits_%d' % i]), dim=1), torch.from_numpy(fx['u_%d' % i]),
                                 float(fx['temp_%d' % i]))
        assert np.array_equal(y.numpy(), fx['y_%d' % i])


@pytest.mark.parametrize('name', ['gumbel_gru', 'gumbel_tfm'])
def test_gumbel_restatement_matches_reference_fixture(name):
    """The gumbel branch of the restated loop against the real loop's outputs (same uniform draws)."""
    fx = load_fixture(name)
    cfg = fx['config']
    model, vocab = torch_model_for(cfg)
    V = cfg['V']
    prefix = torch.zeros(fx['dists'].shape[0], 1, V)
    prefix[:, 0, vocab.stoi['<bos>']] = 1.0
    state = {'vocab': vocab, 'max_len': cfg['max_len'], 'gumbel': True, 'temperature': cfg['temperature'], 'active_seqs': prefix}
    if cfg['kind'] == 'gumbel_gru':
        state.update(model=R.SoftInput(model, V), inputs=torch.from_numpy(fx['z']))
    else:
        state.update(model=R.SoftInput(model.decoder, V), inputs=torch.from_numpy(fx['encs']),
                     enc_mask=torch.from_numpy(fx['enc_mask']))
    with replay_rand([fx['u_s%d' % i] for i in range(cfg['n_steps'])]):
        out = R.RefInferenceState(state).inference()
    assert [o.shape[1] // V for o in out] == fx['lens'].tolist()
    for i, o in enumerate(out):
        assert tuple(o.shape) == (1, fx['lens'][i] * V)
        assert np.array_equal(o.reshape(-1, V).numpy(), fx['dists'][i, :fx['lens'][i]])


def test_canonical_rnn_encoder_and_generator_close_to_torch():
    """The pre-decode pass of the GRU path (/root/reference/src/trainers/cyclegan_trainer.py:321-327,337-348): the
    canonical-arithmetic oracle against stock PyTorch on the specification modules."""
    vocab = R.synthetic_vocab(50)
    enc = R.synthetic_init(R.RNNEncoder(32, 64, vocab), seed=5)
    gen = R.synthetic_init(R.Generator(64, 3), seed=6)
    x = torch.from_numpy(np.random.RandomState(2).randint(0, 50, size=(7, 11)))
    with torch.no_grad():
        z = enc(x)
        z2 = gen(z)
    cz = C.rnn_encoder(enc, x.numpy())
    ok, err = logits_close(cz, z.numpy())
    assert ok, err
    ok, err = logits_close(C.generator(gen, z.numpy()), z2.numpy())
    assert ok, err


def ensemble_torch_model(cfg):
    vocab = R.synthetic_vocab(cfg['V'], extra=('|',))
    members = [R.synthetic_init(R.RNNLM(cfg['H'], vocab, cfg['n_layers']), vocab.stoi['|'], cfg['eos_bias'], cfg['out_scale'], seed=sd)
               for sd in cfg['member_seeds']]
    return R.WeightedLMEnsemble(members, cfg['weights']).eval(), members, vocab


def test_conditional_lm_restatements_match_reference_fixture():
    """Style-conditioned GRU language model: the torch restatement of the loop and the canonical oracle against the
    fixture made by the real loop (style passed through ``kwargs``), for the fixture's style and for the other one."""
    fx = load_fixture('conditional_lm')
    cfg = fx['config']
    model, vocab = torch_model_for(cfg)
    B = fx['z'].shape[1]
    for style, want in ((cfg['style'], fx['seqs']), (0, cfg['seqs_style0'])):
        with torch.no_grad():
            got = R.RefInferenceState({'model': model, 'inputs': torch.from_numpy(fx['z']), 'vocab': vocab, 'max_len': cfg['max_len'],
                                       'bos_token': '|', 'eos_token': '|', 'substitute_inputs': True, 'inputs_batch_dim': 1,
                                       'kwargs': {'return_z': True, 'style': style}}).inference()
        assert [s.tolist() for s in got] == want
        sess = C.CanonGRUDecoder(model).session(fx['z'], style=style)
        seqs, logits, rows = C.greedy_decode(sess, B, cfg['max_len'], vocab.stoi['|'], vocab.stoi['|'], KEEP)
        assert seqs == want
        if style == cfg['style']:
            for i in range(sum(1 for k in fx if k.startswith('logits_s'))):
                ok, err = logits_close(logits[i], fx['logits_s%d' % i])
                assert ok, (i, err)
    assert fx['seqs'] != cfg['seqs_style0']          # the style matters in this fixture


def test_canonical_ffn_close_to_torch():
    torch.manual_seed(2)
    m = R.FFN([87, 40, 64])
    x = np.random.RandomState(1).standard_normal((9, 87)).astype(np.float32)
    with torch.no_grad():
        want = m(torch.from_numpy(x)).numpy()
    ok, err = logits_close(C.ffn(m, x), want)
    assert ok, err


def test_ensemble_restatement_matches_reference_fixture():
    """``oracle.torch_ref.WeightedLMEnsemble`` under the restated loop against the real class under the real loop."""
    fx = load_fixture('ensemble_rnnlm')
    cfg = fx['config']
    ens, _, vocab = ensemble_torch_model(cfg)
    st = R.RefInferenceState({'model': ens, 'inputs': torch.from_numpy(fx['zs']), 'vocab': vocab, 'max_len': cfg['max_len'],
                              'bos_token': '|', 'eos_token': '|', 'substitute_inputs': True, 'inputs_batch_dim': 2,
                              'kwargs': {'return_z': True}})
    seqs = [s.tolist() for s in st.inference(keep_logits=KEEP)]
    assert seqs == fx['seqs']
    for i in range(sum(1 for k in fx if k.startswith('logits_s'))):
        assert np.array_equal(st.step_rows[i].numpy(), fx['rows_s%d' % i])
        assert np.array_equal(st.step_logits[i].numpy(), fx['logits_s%d' % i])


@pytest.mark.parametrize('name', ['prefix_tfm', 'prefix_rnnlm'])
def test_prefix_restatement_matches_reference_fixture(name):
    """A caller-provided ``active_seqs`` of token ids (/root/reference/src/inference.py:37) under the restated loop."""
    fx = load_fixture(name)
    cfg = fx['config']
    model, vocab = torch_model_for(cfg)
    prefix = torch.from_numpy(fx['prefix']).long()
    if cfg['kind'] == 'transformer':
        state = {'model': model.decoder, 'inputs': torch.from_numpy(fx['encs']), 'enc_mask': torch.from_numpy(fx['enc_mask'])}
    else:
        state = {'model': model, 'inputs': torch.from_numpy(fx['z']), 'bos_token': '|', 'eos_token': '|',
                 'substitute_inputs': True, 'inputs_batch_dim': 1, 'kwargs': {'return_z': True}}
    state.update(vocab=vocab, max_len=cfg['max_len'], active_seqs=prefix)
    st = R.RefInferenceState(state)
    assert [s.tolist() for s in st.inference(keep_logits=4)] == fx['seqs']
    for i in range(sum(1 for k in fx if k.startswith('logits_s'))):
        assert np.array_equal(st.step_rows[i].numpy(), fx['rows_s%d' % i])
        assert np.array_equal(st.step_logits[i].numpy(), fx['logits_s%d' % i])
