"""GPU: soft ("gumbel") decoding, ``gumbel=True`` of /root/reference/src/inference.py:83-88,104-105,136-138.

Floating-point path, so the bar is a tolerance: distributions within |a-b| <= 1e-4 |b| + 1e-6 of the outputs
of the real reference loop (fixtures ``gumbel_*``, made with ``oracle.torch_ref.SoftInput`` models and the same
uniform draws), sequence lengths (which token chose EOS, at which step) identical.
"""
import numpy as np
import pytest
import torch

from helpers import load_fixture, replay_rand, torch_model_for
from oracle import torch_ref as R

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-4, 1e-6


def close(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    err = np.abs(a - b) - (RTOL * np.abs(b) + ATOL)
    return bool((err <= 0).all()), float(np.abs(a - b).max())


@pytest.fixture(scope='module')
def T():
    import tst_amd
    assert torch.cuda.is_available()
    return tst_amd


def test_gumbel_softmax_kernel_matches_reference_function(T):
    """tst_gumbel_softmax_f32 against outputs of the reference's own ``gumbel_softmax_sample`` (odd and large V)."""
    from tst_amd import kernels as K
    fx = load_fixture('gumbel_op')
    for i in range(3):
        logits = torch.from_numpy(fx['logits_%d' % i]).cuda()
        u = torch.from_numpy(fx['u_%d' % i]).cuda()
        rows, V = logits.shape
        out = torch.full((rows, V + 3), -7.0, device='cuda')          # a strided output with untouched padding
        K.gumbel_softmax(logits, rows, V, u, float(fx['temp_%d' % i]), out)
        ok, err = close(out[:, :V].cpu().numpy(), fx['y_%d' % i])
        assert ok, (i, err)
        assert bool((out[:, V:] == -7.0).all())
        assert np.allclose(out[:, :V].sum(1).cpu().numpy(), 1.0, atol=1e-5)


def test_gumbel_softmax_kernel_extreme_draws(T):
    """u = 0 (the eps inside the logs keeps it finite) and u next to 1; a one-hot-like logit row."""
    from tst_amd import kernels as K
    V = 50
    logits = torch.zeros((2, V))
    logits[1, 7] = 60.0
    u = torch.rand((2, V), generator=torch.Generator().manual_seed(0))
    u[0, 0], u[0, 1], u[1, 3] = 0.0, 1.0 - 2.0 ** -24, 0.0
    want = R.ref_gumbel_softmax(torch.softmax(logits, dim=1), u, 0.3).numpy()
    out = torch.empty((2, V), device='cuda')
    K.gumbel_softmax(logits.cuda(), 2, V, u.cuda(), 0.3, out)
    ok, err = close(out.cpu().numpy(), want)
    assert ok, err
    assert np.isfinite(out.cpu().numpy()).all()


def native_model(T, cfg):
    tmodel, _ = torch_model_for(cfg)
    if cfg['kind'] == 'gumbel_gru':
        vocab = T.Vocab.synthetic(cfg['V'], extra=('|',))
        model = T.RNNDecoder(cfg['E'], cfg['H'], vocab)
    else:
        vocab = T.Vocab.synthetic(cfg['V'])
        model = T.Transformer(T.HP(cfg['hp']), vocab, vocab)
    model.load_state_dict(tmodel.state_dict())
    return tmodel, model, vocab


def state_dict_for(fx, model, vocab, **extra):
    cfg = fx['config']
    d = {'vocab': vocab, 'max_len': cfg['max_len'], 'gumbel': True, 'temperature': cfg['temperature']}
    if cfg['kind'] == 'gumbel_gru':
        d.update(model=model, inputs=torch.from_numpy(fx['z']).cuda())
    else:
        d.update(model=model.decoder if hasattr(model, 'decoder') else model, inputs=torch.from_numpy(fx['encs']).cuda(),
                 enc_mask=torch.from_numpy(fx['enc_mask']).cuda())
    d.update(extra)
    return d


def check_against_fixture(out, fx):
    V = fx['config']['V']
    assert [o.shape[1] // V for o in out] == fx['lens'].tolist()
    worst = 0.0
    for i, o in enumerate(out):
        assert o.is_cuda and o.dtype == torch.float32 and tuple(o.shape) == (1, fx['lens'][i] * V)
        ok, err = close(o.reshape(-1, V).cpu().numpy(), fx['dists'][i, :fx['lens'][i]])
        assert ok, (i, err)
        worst = max(worst, err)
    return worst


@pytest.mark.parametrize('name', ['gumbel_gru', 'gumbel_tfm'])
@pytest.mark.parametrize('prefix', ['one_hot_given', 'default'])
def test_gumbel_decode_matches_reference_fixture(T, name, prefix):
    fx = load_fixture(name)
    cfg = fx['config']
    _, model, vocab = native_model(T, cfg)
    extra = {}
    if prefix == 'one_hot_given':     # the form in which the reference loop runs
        p = torch.zeros(fx['dists'].shape[0], 1, cfg['V'])
        p[:, 0, vocab.stoi['<bos>']] = 1.0
        extra['active_seqs'] = p
    with replay_rand([fx['u_s%d' % i] for i in range(cfg['n_steps'])]):
        st = T.InferenceState(state_dict_for(fx, model, vocab, **extra))
        out = st.inference(keep_logits=4)
    check_against_fixture(out, fx)
    for i in range(sum(1 for k in fx if k.startswith('logits_s'))):
        lg = st.loop.kept[i][0].cpu().numpy()
        assert np.allclose(lg, fx['logits_s%d' % i], rtol=1e-4, atol=1e-5), i


def test_gumbel_foreign_model_gets_the_reference_layout(T):
    """Any other callable is fed ``[B, 1, t * V]`` exactly as the reference loop builds it."""
    fx = load_fixture('gumbel_gru')
    cfg = fx['config']
    tmodel, _, vocab = native_model(T, cfg)
    seen = []
    soft = R.SoftInput(tmodel.cuda(), cfg['V'])

    def model(z, active):
        seen.append(tuple(active.shape))
        return soft(z, active)

    with replay_rand([fx['u_s%d' % i] for i in range(cfg['n_steps'])]):
        out = T.InferenceState(state_dict_for(fx, model, vocab)).inference()
    check_against_fixture(out, fx)
    assert seen[0] == (5, 1, cfg['V']) and all(s[1] == 1 and s[2] == (i + 1) * cfg['V'] for i, s in enumerate(seen))


def test_gumbel_device_noise_and_stacking(T):
    fx = load_fixture('gumbel_tfm')
    cfg = fx['config']
    _, model, vocab = native_model(T, cfg)
    torch.manual_seed(5)
    out = T.InferenceState(state_dict_for(fx, model, vocab, noise='device')).inference()
    V = cfg['V']
    for o in out:
        d = o.reshape(-1, V)
        assert 1 <= d.shape[0] <= cfg['max_len'] + 1
        assert np.allclose(d.sum(1).cpu().numpy(), 1.0, atol=1e-5) and bool((d >= 0).all())
        assert float(d[0, vocab.stoi['<bos>']]) == 1.0
    # should_stack_finished: the reference stacks [1, t * V] rows without padding (len(seq) is 1 for all of them),
    # which only works when every sentence ran the same number of steps
    st = T.InferenceState(state_dict_for(fx, model, vocab, noise='device', should_stack_finished=True, eos_token='w9',
                                         max_len=3))
    try:
        stacked = st.inference()
    except RuntimeError:
        assert len(set(st.loop.out_len.cpu().tolist())) > 1
    else:
        assert len(set(st.loop.out_len.cpu().tolist())) == 1
        assert stacked.dim() == 3 and stacked.shape[0] == fx['dists'].shape[0] and stacked.shape[1] == 1


def test_gumbel_rejects_what_the_reference_cannot_do(T):
    fx = load_fixture('gumbel_gru')
    _, model, vocab = native_model(T, fx['config'])
    with pytest.raises(ValueError):
        T.InferenceState(state_dict_for(fx, model, vocab, beam_size=2)).inference()
    with pytest.raises(NotImplementedError):
        T.InferenceState(state_dict_for(fx, model, vocab, substitute_inputs=True)).inference()
    with pytest.raises(NotImplementedError):
        T.InferenceState(state_dict_for(fx, model, vocab, active_seqs=torch.zeros(5, 2, 40))).inference()


def test_gumbel_transformer_lm_against_restated_loop(T):
    """The z-conditioned transformer LM in soft mode against the restated reference loop on the torch specification
    (which the gumbel fixtures pin to the real loop), with the same uniform draws."""
    hp = dict(model_size=64, n_layers=2, n_heads=4, d_ff=128, pe_max_len=32)
    V, B, max_len = 90, 5, 7
    tv = R.synthetic_vocab(V)
    tmodel = R.synthetic_init(R.TransformerLM(R.HP(hp), tv), 3, 2.0, 2.0, seed=9)
    vocab = T.Vocab.synthetic(V)
    lm = T.TransformerLM(T.HP(hp), vocab)
    lm.load_state_dict(tmodel.state_dict())
    r = np.random.RandomState(12)
    z = (0.3 * r.standard_normal((B, 64))).astype(np.float32)
    draws = [r.random_sample((B, V)).astype(np.float32) for _ in range(max_len)]
    prefix = torch.zeros(B, 1, V)
    prefix[:, 0, 2] = 1.0
    # reference side: rows leave the batch when they end, so the draws of step i are the first n_active rows
    class Shrinking(replay_rand):
        def __enter__(self):
            outer = self
            super().__enter__()
            fake = torch.rand

            def shaped(*size, **kw):
                shape = tuple(size[0]) if len(size) == 1 and not isinstance(size[0], int) else tuple(size)
                outer.draws[outer.i] = outer.draws[outer.i][:shape[0]]
                return fake(*size, **kw)
            torch.rand = shaped
            return self
    with Shrinking(list(draws)):
        want = R.RefInferenceState({'model': R.SoftInput(tmodel, V), 'inputs': torch.from_numpy(z), 'vocab': tv, 'max_len': max_len,
                                    'gumbel': True, 'temperature': 0.8, 'active_seqs': prefix}).inference()
    with Shrinking(list(draws)):
        got = T.InferenceState({'model': lm, 'inputs': torch.from_numpy(z).cuda(), 'vocab': vocab, 'max_len': max_len,
                                'gumbel': True, 'temperature': 0.8, 'active_seqs': prefix}).inference()
    assert [tuple(g.shape) for g in got] == [tuple(w.shape) for w in want]
    for g, w in zip(got, want):
        ok, err = close(g.cpu().numpy(), w.numpy())
        assert ok, err
    lm16 = T.TransformerLM(T.HP(hp), vocab, dtype='bf16')
    lm16.load_state_dict(tmodel.state_dict())
    with pytest.raises(ValueError):      # the soft step needs fp32 logits
        T.InferenceState({'model': lm16, 'inputs': torch.from_numpy(z).cuda(), 'vocab': vocab, 'max_len': 3, 'gumbel': True,
                          'noise': 'device'}).inference()
