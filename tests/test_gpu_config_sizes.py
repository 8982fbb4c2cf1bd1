"""GPU: the decode path at the full sizes of the benchmark configurations.

The oracle cannot decode 8192 sentences over a 32000-word vocabulary in seconds, so the full-size runs are
checked (a) against the oracle on a random sample of the batch's sentences — a sentence's output does not
depend on its batch, so the sample decoded alone by the oracle must equal the same sentences inside the
full GPU batch, bit for bit — and (b) through size-independent properties: beam of one equals greedy, the
same batch in a different order gives the same sentences, output format invariants.
"""
import numpy as np
import pytest
import torch

from oracle import canon as C
from oracle import torch_ref as R

pytestmark = pytest.mark.gpu

HP = dict(d_model=256, dropout=0.1, n_layers=2, pe_max_len=60, n_heads=2, d_ff=1024)
V, MAX_LEN, EOS_BIAS, OUT_SCALE = 32000, 50, 5.5, 2.0


@pytest.fixture(scope='module')
def T():
    import tst_amd
    assert torch.cuda.is_available()
    return tst_amd


@pytest.fixture(scope='module')
def mt(T):
    """The benchmark's model (transformer-mt sizes, V = 32000, seed 42), its torch specification twin and
    the 8192-sentence synthetic batch."""
    from tst_amd import synthetic
    vocab = T.Vocab.synthetic(V)
    model = T.Transformer(T.HP(HP), vocab, vocab)
    model.load_state_dict(synthetic.synthetic_state_dict(model, 3, EOS_BIAS, OUT_SCALE, seed=42))
    tv = R.synthetic_vocab(V)
    tmodel = R.synthetic_init(R.Transformer(R.HP(HP), tv, tv), 3, EOS_BIAS, OUT_SCALE, seed=42)
    src = synthetic.multi30k_like_sources(8192, V, seed=1236)
    return model, tmodel, vocab, src


def gpu_decode(T, model, vocab, src, beam, alpha=0.0):
    encs, enc_mask = model.encoder(src.cuda())
    st = T.InferenceState({'model': model.decoder, 'inputs': encs, 'enc_mask': enc_mask, 'vocab': vocab,
                           'max_len': MAX_LEN, 'beam_size': beam, 'length_penalty': alpha})
    tok, lens, score = st.decode_packed()
    return tok.cpu().numpy(), lens.cpu().numpy(), score.cpu().numpy()


def oracle_decode(tmodel, src, beam, alpha=0.0):
    encs, mask = C.CanonTransformerEncoder(tmodel.encoder, HP['n_heads'])(src.numpy())
    sess = C.CanonTransformerDecoder(tmodel.decoder, HP['n_heads']).session(encs, mask, beam, MAX_LEN)
    if beam == 1:
        return C.greedy_decode(sess, src.shape[0], MAX_LEN, 2, 3)[0], None
    seqs, scores, _, _ = C.beam_decode(sess, src.shape[0], beam, MAX_LEN, 2, 3, alpha)
    return seqs, scores


def rows_as_lists(tok, lens, idx):
    return [tok[i, :lens[i]].tolist() for i in idx]


def test_config2_beam4_full_batch_token_identical_on_sample(T, mt):
    """BASELINE config 2: 8192 sentences, beam 4, fp32, V = 32000.  64 sampled sentences decoded by the
    oracle equal the GPU's output for those sentences inside the full batch: tokens and scores bit for bit."""
    model, tmodel, vocab, src = mt
    tok, lens, score = gpu_decode(T, model, vocab, src, 4)
    pick = np.sort(np.random.RandomState(5).choice(8192, size=64, replace=False))
    want, wscore = oracle_decode(tmodel, src[pick], 4)
    assert rows_as_lists(tok, lens, pick) == want
    assert np.array_equal(score[pick], wscore)
    # format invariants over the whole batch: BOS first, no EOS inside, at most 1 + max_len tokens, pad after
    assert (tok[:, 0] == 2).all() and lens.min() >= 1 and lens.max() <= MAX_LEN + 1
    col = np.arange(MAX_LEN + 1)[None, :]
    inside = col < lens[:, None]
    assert not (tok[inside] == 3).any() and (tok[~inside] == 1).all()
    assert len(set(lens.tolist())) > 10                                  # ragged termination at scale


def test_config2_length_penalty_sample(T, mt):
    """The same with a length penalty (no early termination on a finished top beam), 2048 sentences."""
    model, tmodel, vocab, src = mt
    tok, lens, score = gpu_decode(T, model, vocab, src[:2048], 4, alpha=0.6)
    pick = np.sort(np.random.RandomState(6).choice(2048, size=32, replace=False))
    want, wscore = oracle_decode(tmodel, src[pick], 4, alpha=0.6)
    assert rows_as_lists(tok, lens, pick) == want
    assert np.array_equal(score[pick], wscore)


def test_config1_greedy_256_sentences_vs_oracle_sample_and_properties(T, mt):
    """BASELINE config 1 (256 sentences, greedy) on the GPU: a 32-sentence sample equals the oracle; the beam
    machinery with one beam and a shuffled batch give the same sentences."""
    from tst_amd import synthetic
    model, tmodel, vocab, _ = mt
    src = synthetic.multi30k_like_sources(256, V, seed=1235)
    tok, lens, _ = gpu_decode(T, model, vocab, src, 1)
    pick = np.sort(np.random.RandomState(7).choice(256, size=32, replace=False))
    want, _ = oracle_decode(tmodel, src[pick], 1)
    assert rows_as_lists(tok, lens, pick) == want
    perm = np.random.RandomState(8).permutation(256)
    tok2, lens2, _ = gpu_decode(T, model, vocab, src[perm], 1)
    assert rows_as_lists(tok2, lens2, range(256)) == rows_as_lists(tok, lens, perm)


def test_config2_batch_order_independence_full_size(T, mt):
    """8192 sentences, beam 4: decoding the batch in another order permutes the outputs and nothing else
    (compaction, beam reordering and the ancestor-indexed KV cache do not leak between sentences)."""
    model, _, vocab, src = mt
    tok, lens, score = gpu_decode(T, model, vocab, src, 4)
    perm = np.random.RandomState(9).permutation(8192)
    tok2, lens2, score2 = gpu_decode(T, model, vocab, src[perm], 4)
    assert np.array_equal(lens2, lens[perm]) and np.array_equal(tok2, tok[perm]) and np.array_equal(score2, score[perm])


def test_config3_gru_full_batch_sample_vs_oracle(T):
    """BASELINE config 3: GRU decoder E = H = 1024 (cyclegan sizes), character vocabulary, EOS '|', 8192
    sentences, greedy and beam 4: 32 sampled sentences equal the oracle (tokens, beam scores) bit for bit."""
    from tst_amd import synthetic
    Vc = 128
    vocab = T.Vocab.synthetic(Vc, extra=('|',))
    model = T.RNNDecoder(1024, 1024, vocab)
    model.load_state_dict(synthetic.synthetic_state_dict(model, vocab.stoi['|'], 0.5, 2.0, seed=42))
    tv = R.synthetic_vocab(Vc, extra=('|',))
    tmodel = R.synthetic_init(R.RNNDecoder(1024, 1024, tv), tv.stoi['|'], 0.5, 2.0, seed=42)
    z = torch.randn((8192, 1024), generator=torch.Generator().manual_seed(1237))
    pick = np.sort(np.random.RandomState(10).choice(8192, size=32, replace=False))
    eos = vocab.stoi['|']
    ML = 100                                                       # simple_inference's default (src/inference.py:170)
    for beam in (1, 4):
        st = T.InferenceState({'model': model, 'inputs': z.cuda(), 'vocab': vocab, 'max_len': ML, 'eos_token': '|',
                               'beam_size': beam})
        tok, lens, score = (x.cpu().numpy() for x in st.decode_packed())
        sess = C.CanonGRUDecoder(tmodel).session(z[pick].numpy(), beam)
        if beam == 1:
            want = C.greedy_decode(sess, 32, ML, 2, eos)[0]
        else:
            want, wscore, _, _ = C.beam_decode(sess, 32, beam, ML, 2, eos)
            assert np.array_equal(score[pick], wscore)
        assert rows_as_lists(tok, lens, pick) == want
        assert len(set(lens.tolist())) > 5 and lens.max() <= ML + 1


@pytest.mark.parametrize('eos_bias', [1.0, 0.1])
def test_config3_gru_v32000_full_batch_sample_vs_oracle(T, eos_bias):
    """BASELINE config 3, the 32000-word variant (the K = 1024 vocabulary projection dominates it): 8192 sentences, greedy and
    beam 4, max_len 100; 48 sampled sentences equal the oracle bit for bit (tokens and beam scores).  Two EOS biases: the one
    the throughput figures are quoted on (short sentences) and one that keeps sentences alive for tens of steps."""
    from tst_amd import synthetic
    vocab = T.Vocab.synthetic(V, extra=('|',))
    model = T.RNNDecoder(1024, 1024, vocab)
    model.load_state_dict(synthetic.synthetic_state_dict(model, vocab.stoi['|'], eos_bias, 2.0, seed=42))
    tv = R.synthetic_vocab(V, extra=('|',))
    tmodel = R.synthetic_init(R.RNNDecoder(1024, 1024, tv), tv.stoi['|'], eos_bias, 2.0, seed=42)
    z = torch.randn((8192, 1024), generator=torch.Generator().manual_seed(1237))
    n_pick = 48 if eos_bias > 0.5 else 12
    pick = np.sort(np.random.RandomState(13).choice(8192, size=n_pick, replace=False))
    eos = vocab.stoi['|']
    ML = 100
    for beam in (1, 4):
        st = T.InferenceState({'model': model, 'inputs': z.cuda(), 'vocab': vocab, 'max_len': ML, 'eos_token': '|',
                               'beam_size': beam})
        tok, lens, score = (x.cpu().numpy() for x in st.decode_packed())
        del st
        torch.cuda.empty_cache()
        sess = C.CanonGRUDecoder(tmodel).session(z[pick].numpy(), beam)
        if beam == 1:
            want = C.greedy_decode(sess, n_pick, ML, 2, eos)[0]
        else:
            want, wscore, _, _ = C.beam_decode(sess, n_pick, beam, ML, 2, eos)
            assert np.array_equal(score[pick], wscore)
        assert rows_as_lists(tok, lens, pick) == want
        assert len(set(lens.tolist())) > 2 and lens.max() <= ML + 1


def test_config4_lm_sizes_fp32_sample_and_bf16_properties(T):
    """BASELINE config 4 sizes (model_size 512, 6 layers, 8 heads, d_ff 2048, V = 32000, beam 8, max_len 50).  fp32: 6
    sampled sentences of a 2048-sentence batch equal the oracle bit for bit (tokens and scores).  bf16 (tolerance-checked
    position by position at small sizes in test_gpu_lm_bf16.py): at the FULL size, 8192 sentences x beam 8 x 50 steps (a 60 GB
    q|k|v cache), batch-order independence (beam 8 and greedy) and the output format must hold."""
    hp = dict(model_size=512, n_layers=6, n_heads=8, d_ff=2048, pe_max_len=64)
    vocab = T.Vocab.synthetic(V)
    tv = R.synthetic_vocab(V)
    tmodel = R.synthetic_init(R.TransformerLM(R.HP(hp), tv), 3, 5.0, 2.0, seed=42)
    z = 0.3 * torch.randn((8192, 512), generator=torch.Generator().manual_seed(1238))
    T_ = 50
    lms = {}

    def run(dtype, zz, beam=8):
        if dtype not in lms:
            lms[dtype] = T.TransformerLM(T.HP(hp), vocab, dtype=dtype)
            lms[dtype].load_state_dict(tmodel.state_dict())
        st = T.InferenceState({'model': lms[dtype], 'inputs': zz.cuda(), 'vocab': vocab, 'max_len': T_, 'beam_size': beam})
        out = tuple(x.cpu().numpy() for x in st.decode_packed())
        del st
        torch.cuda.empty_cache()
        return out

    tok, lens, score = run('fp32', z[:2048])
    pick = np.sort(np.random.RandomState(11).choice(2048, size=6, replace=False))
    want, wscore, _, _ = C.beam_decode(C.CanonTransformerLM(tmodel).session(z[pick].numpy(), 8, T_), 6, 8, T_, 2, 3)
    assert rows_as_lists(tok, lens, pick) == want
    assert np.array_equal(score[pick], wscore)
    assert len(set(lens.tolist())) > 10
    tok16, lens16, score16 = run('bf16', z)
    perm = np.random.RandomState(12).permutation(8192)
    tok16p, lens16p, score16p = run('bf16', z[perm])
    assert np.array_equal(tok16p, tok16[perm]) and np.array_equal(lens16p, lens16[perm]) and np.array_equal(score16p, score16[perm])
    assert (tok16[:, 0] == 2).all() and lens16.max() <= T_ + 1 and lens16.min() >= 1
    col = np.arange(T_ + 1)[None, :]
    inside = col < lens16[:, None]
    assert not (tok16[inside] == 3).any() and (tok16[~inside] == 1).all()
    g1 = run('bf16', z[:512], beam=1)
    g1b = run('bf16', z[perm[:300]], beam=1)
    where = {int(s): i for i, s in enumerate(perm[:300]) if s < 512}
    assert len(where) > 5 and all(np.array_equal(g1[0][s], g1b[0][i]) for s, i in where.items())
    agree = float(np.mean([a == b for a, b in zip(rows_as_lists(tok16, lens16, range(2048)), rows_as_lists(tok, lens, range(2048)))]))
    print('\n[config 4] bf16 vs fp32 GPU path: %.0f %% of 2048 beam-8 outputs identical' % (100 * agree))


@pytest.mark.parametrize('kind,n_sent,beam,chunk', [('tfm', 45000, 4, 9000), ('gru', 70000, 4, 14000), ('lm_bf16', 14000, 8, 3500)])
def test_batches_beyond_2_31_elements_equal_chunked_decodes(kind, n_sent, beam, chunk):
    """Batches whose logits (rows x 32000) and, for the transformers, whose cache of one layer (rows x 50 x width) hold more
    than 2^31 elements (180 000 / 280 000 / 112 000 rows): every offset in the kernels has to be 64-bit.  A sentence's output
    does not depend on its batch, so the one large batch must give exactly what the same sentences give in chunks (tokens,
    lengths, scores).  Runs tools/probe/large_batch.py in a process of its own (it needs 50-115 GB of device memory)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert n_sent * beam * V > 2 ** 31
    torch.cuda.empty_cache()
    p = subprocess.run([sys.executable, os.path.join(root, 'tools', 'probe', 'large_batch.py'), str(n_sent), str(beam), str(chunk), kind],
                       capture_output=True, text=True, timeout=400)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-1500:])
    assert 'RESULT identical' in p.stdout
