"""GPU: edge cases of the decode path — smallest and degenerate batches, immediate and never-occurring EOS,
missing masks, odd vocabulary sizes, a foreign model under beam search — each against the canonical oracle."""
import numpy as np
import pytest
import torch

from oracle import canon as C
from oracle import torch_ref as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def T():
    import tst_amd
    assert torch.cuda.is_available()
    return tst_amd


def mt_case(T, V, B, S, eos_bias, seed, lens=None):
    hp = R.mt_hp()
    tv = R.synthetic_vocab(V)
    tmodel = R.synthetic_init(R.Transformer(R.HP(hp), tv, tv), 3, eos_bias, 2.0, seed=seed)
    vocab = T.Vocab.synthetic(V)
    model = T.Transformer(T.HP(hp), vocab, vocab)
    model.load_state_dict(tmodel.state_dict())
    r = np.random.RandomState(seed)
    lens = r.randint(1, S + 1, size=B) if lens is None else np.asarray(lens)
    encs = r.standard_normal((B, S, hp['d_model'])).astype(np.float32)
    mask = (np.arange(S)[None] < lens[:, None])[:, None, :]
    return tmodel, model, vocab, encs, mask


def decode(T, model, vocab, encs, mask, max_len, beam, **kw):
    d = {'model': model.decoder, 'inputs': torch.from_numpy(encs).cuda(), 'vocab': vocab, 'max_len': max_len, 'beam_size': beam}
    if mask is not None:
        d['enc_mask'] = torch.from_numpy(mask).cuda()
    d.update(kw)
    st = T.InferenceState(d)
    return [s.cpu().tolist() for s in st.inference()], st


def oracle(tmodel, encs, mask, max_len, beam, alpha=0.0):
    sess = C.CanonTransformerDecoder(tmodel.decoder, 2).session(encs, mask, beam, max_len)
    if beam == 1:
        return C.greedy_decode(sess, encs.shape[0], max_len, 2, 3)[0], None
    seqs, scores, _, _ = C.beam_decode(sess, encs.shape[0], beam, max_len, 2, 3, alpha)
    return seqs, scores


@pytest.mark.parametrize('B,beam,max_len', [(1, 1, 1), (1, 8, 1), (1, 8, 9), (3, 4, 1), (2, 2, 30), (257, 3, 5)])
def test_smallest_batches_and_single_steps(T, B, beam, max_len):
    tmodel, model, vocab, encs, mask = mt_case(T, 300, B, 7, 2.0, 21)
    got, st = decode(T, model, vocab, encs, mask, max_len, beam)
    want, scores = oracle(tmodel, encs, mask, max_len, beam)
    assert got == want
    if scores is not None:
        assert np.array_equal(st.scores.cpu().numpy(), scores)
    assert all(s[0] == 2 and len(s) <= 1 + max_len for s in got)


@pytest.mark.parametrize('beam', [1, 4])
def test_every_sentence_ends_at_once_and_none_ever_ends(T, beam):
    tmodel, model, vocab, encs, mask = mt_case(T, 200, 9, 6, 40.0, 22)          # EOS wins every arg-max
    got, st = decode(T, model, vocab, encs, mask, 12, beam)
    assert got == [[2]] * 9 and st.num_steps_done == 0
    assert got == oracle(tmodel, encs, mask, 12, beam)[0]
    tmodel, model, vocab, encs, mask = mt_case(T, 200, 9, 6, -40.0, 23)         # EOS never wins
    got, st = decode(T, model, vocab, encs, mask, 12, beam)
    assert all(len(s) == 13 and 3 not in s for s in got) and st.num_steps_done == 11
    assert got == oracle(tmodel, encs, mask, 12, beam)[0]


def test_no_mask_one_position_sources_and_max_positions(T):
    tmodel, model, vocab, encs, mask = mt_case(T, 150, 5, 4, 1.5, 24)
    got, _ = decode(T, model, vocab, encs, None, 10, 4)                          # enc_mask omitted: attend to everything
    assert got == oracle(tmodel, encs, None, 10, 4)[0]
    tmodel, model, vocab, encs, mask = mt_case(T, 150, 5, 4, 1.5, 25, lens=[1, 1, 4, 1, 2])
    got, _ = decode(T, model, vocab, encs, mask, 10, 4)                          # a single unmasked source position
    assert got == oracle(tmodel, encs, mask, 10, 4)[0]
    tmodel, model, vocab, encs, mask = mt_case(T, 150, 3, 60, -30.0, 26)         # pe_max_len = 60: longest source, longest decode
    got, _ = decode(T, model, vocab, encs, mask, 60, 2)
    assert got == oracle(tmodel, encs, mask, 60, 2)[0] and all(len(s) == 61 for s in got)
    with pytest.raises(ValueError, match='pe_max_len'):
        decode(T, model, vocab, encs, mask, 61, 2)


@pytest.mark.parametrize('V', [5, 37, 130, 1001])
def test_odd_vocabulary_sizes_gru_greedy_beam_and_sampling(T, V):
    """Vocabulary sizes that are not multiples of 4 / 16 (ragged last chunk of every kernel over V); beam up to V."""
    tv = R.synthetic_vocab(V)
    tmodel = R.synthetic_init(R.RNNDecoder(32, 64, tv), 3, 1.0, 2.0, seed=V)
    vocab = T.Vocab.synthetic(V)
    model = T.RNNDecoder(32, 64, vocab)
    model.load_state_dict(tmodel.state_dict())
    z = np.random.RandomState(V).standard_normal((11, 64)).astype(np.float32)
    for beam in (1, min(5, V)):
        st = T.InferenceState({'model': model, 'inputs': torch.from_numpy(z).cuda(), 'vocab': vocab, 'max_len': 9, 'beam_size': beam})
        got = [s.cpu().tolist() for s in st.inference()]
        sess = C.CanonGRUDecoder(tmodel).session(z, beam)
        want = C.greedy_decode(sess, 11, 9, 2, 3)[0] if beam == 1 else C.beam_decode(sess, 11, beam, 9, 2, 3)[0]
        assert got == want
    st = T.InferenceState({'model': model, 'inputs': torch.from_numpy(z).cuda(), 'vocab': vocab, 'max_len': 9,
                           'sample_type': 'sample', 'temperature': 1.7, 'seed': 4})
    got = [s.cpu().tolist() for s in st.inference()]
    assert got == C.greedy_decode(C.CanonGRUDecoder(tmodel).session(z), 11, 9, 2, 3, sample=(1.7, 4))[0]
    with pytest.raises(ValueError):
        T.InferenceState({'model': model, 'inputs': torch.from_numpy(z).cuda(), 'vocab': vocab, 'beam_size': 8 if V < 8 else 9}).inference()


def test_foreign_model_under_beam_search(T):
    """A model that is not from this package (the torch specification on the GPU, whole-prefix calls): beam search
    reorders its inputs and mask through index_select; same sequences as the native model."""
    tmodel, model, vocab, encs, mask = mt_case(T, 120, 6, 8, 1.0, 27)
    dec = tmodel.decoder.cuda()
    st = T.InferenceState({'model': dec, 'inputs': torch.from_numpy(encs).cuda(), 'enc_mask': torch.from_numpy(mask).cuda(),
                           'vocab': vocab, 'max_len': 10, 'beam_size': 3})
    got = [s.cpu().tolist() for s in st.inference()]
    native, _ = decode(T, model, vocab, encs, mask, 10, 3)
    assert got == native


def test_foreign_model_with_nan_logits_and_out_of_range_prefix_are_refused(T):
    """Non-finite logits have no rank in the selection kernels (torch.max would return a NaN's index): a foreign callable
    that produces them is refused instead of decoded; a prefix with ids outside the vocabulary raises like nn.Embedding."""
    vocab = T.Vocab.synthetic(50)

    def bad_model(inputs, ys):
        out = torch.zeros((ys.shape[0], ys.shape[1], 50), device='cuda')
        out[0, -1, 7] = float('nan')
        return out

    with pytest.raises(FloatingPointError):
        T.InferenceState({'model': bad_model, 'inputs': torch.zeros(3, 4).cuda(), 'vocab': vocab, 'max_len': 4}).inference()

    def minus_inf_model(inputs, ys):
        out = torch.full((ys.shape[0], ys.shape[1], 50), float('-inf'), device='cuda')
        out[:, -1, 9] = 0.0
        return out

    out = T.InferenceState({'model': minus_inf_model, 'inputs': torch.zeros(3, 4).cuda(), 'vocab': vocab, 'max_len': 3}).inference()
    assert [s.tolist() for s in out] == [[2, 9, 9, 9]] * 3            # -inf is an ordinary smallest value
    gru = T.RNNDecoder(32, 32, vocab)
    with pytest.raises(IndexError):
        T.InferenceState({'model': gru, 'inputs': torch.zeros(2, 32).cuda(), 'vocab': vocab, 'max_len': 4,
                          'active_seqs': torch.tensor([[2, 50], [2, 5]])}).inference()


@pytest.mark.parametrize('beam', [1, 4])
def test_loop_without_per_step_count_read_equals_exact_loop(T, beam):
    """Small batches launch a step before the previous step's survivor count has reached the host (the kernels take the true
    count from device memory, the host's is an upper bound): outputs, scores and step count must equal the loop that reads
    the count after every step; also for a GRU model, whose state rows are gathered on every step."""
    from tst_amd import engine
    tmodel, model, vocab, encs, mask = mt_case(T, 300, 37, 9, 2.0, 31)
    old = engine.LAG_MAX_ROWS
    outs = []
    try:
        for limit in (0, 1 << 20):
            engine.LAG_MAX_ROWS = limit
            got, st = decode(T, model, vocab, encs, mask, 14, beam)
            assert st.loop.lagged == (limit > 0)
            outs.append((got, st.scores.cpu().numpy().copy(), st.num_steps_done))
        assert outs[0][0] == outs[1][0] and np.array_equal(outs[0][1], outs[1][1]) and outs[0][2] == outs[1][2]
        assert outs[0][0] == oracle(tmodel, encs, mask, 14, beam)[0]
        vg = T.Vocab.synthetic(60, extra=('|',))
        gru = T.RNNLM(64, vg, 2)
        z = torch.randn((2, 33, 64), generator=torch.Generator().manual_seed(5)).cuda()
        res = []
        for limit in (0, 1 << 20):
            engine.LAG_MAX_ROWS = limit
            st = T.InferenceState({'model': gru, 'inputs': z, 'vocab': vg, 'max_len': 12, 'bos_token': '|', 'eos_token': '|',
                                   'substitute_inputs': True, 'inputs_batch_dim': 1, 'kwargs': {'return_z': True}, 'beam_size': beam})
            res.append([s.tolist() for s in st.inference()])
        assert res[0] == res[1]
    finally:
        engine.LAG_MAX_ROWS = old


def test_greedy_tail_switches_prefixes_and_kept_logits(T):
    """Greedy steps that end in arg-max keys (no logits) and the survivor count written to host memory by the tail kernel,
    against the loop with logits, a separate arg-max and a queued copy, in every combination; with a prefix of several
    tokens (fed step by step before the loop) and with the logits of the first steps kept (those steps must write logits):
    always the oracle's sequences."""
    from tst_amd import engine
    tmodel, model, vocab, encs, mask = mt_case(T, 300, 41, 9, 2.0, 77)
    want = oracle(tmodel, encs, mask, 14, 1)[0]
    old = (engine.KEYS_ON, engine.HOST_COUNT_ON, engine.LAG_MAX_ROWS)
    try:
        for keys_on in (True, False):
            for host_on in (True, False):
                for lag in (0, 1 << 20):
                    engine.KEYS_ON, engine.HOST_COUNT_ON, engine.LAG_MAX_ROWS = keys_on, host_on, lag
                    got, st = decode(T, model, vocab, encs, mask, 14, 1)
                    assert got == want
                    assert (st.loop.keys is not None) == keys_on
        engine.KEYS_ON, engine.HOST_COUNT_ON, engine.LAG_MAX_ROWS = True, True, 1 << 20
        st = T.InferenceState({'model': model.decoder, 'inputs': torch.from_numpy(encs).cuda(), 'enc_mask': torch.from_numpy(mask).cuda(),
                               'vocab': vocab, 'max_len': 14})
        kept = [s.cpu().tolist() for s in st.inference(keep_logits=3)]
        assert kept == want and len(st.loop.kept) == 3 and st.loop.keys is not None
        # a prefix: the first three tokens of every oracle sequence (padded with the sequence's own continuation)
        outs = []
        for keys_on in (True, False):
            engine.KEYS_ON = keys_on
            prefix = torch.tensor([[2, 5 + (i % 7), 9 + (i % 5)] for i in range(41)])
            st = T.InferenceState({'model': model.decoder, 'inputs': torch.from_numpy(encs).cuda(), 'enc_mask': torch.from_numpy(mask).cuda(),
                                   'vocab': vocab, 'max_len': 10, 'active_seqs': prefix})
            outs.append([s.cpu().tolist() for s in st.inference()])
        assert outs[0] == outs[1] and all(o[:3] == [2, 5 + (i % 7), 9 + (i % 5)] for i, o in enumerate(outs[0]))
    finally:
        engine.KEYS_ON, engine.HOST_COUNT_ON, engine.LAG_MAX_ROWS = old


@pytest.mark.parametrize('beam', [1, 4])
def test_concurrent_batches_on_streams_of_their_own_equal_sequential_decodes(T, beam):
    """Independent batches decoded at the same time, one host thread and one HIP stream each over ONE model object, give the
    outputs of the same batches decoded one after the other (the library keeps no state between calls and launches on the
    stream it is given)."""
    import threading
    tmodel, model, vocab, _, _ = mt_case(T, 200, 4, 9, 2.0, 5)
    cases = []
    for i in range(4):
        r = np.random.RandomState(100 + i)
        B, S = 40 + 13 * i, 9
        lens = r.randint(1, S + 1, size=B)
        cases.append((r.standard_normal((B, S, 256)).astype(np.float32), (np.arange(S)[None] < lens[:, None])[:, None, :]))
    want = [decode(T, model, vocab, e, m, 20, beam)[0] for e, m in cases]
    got, errors = [None] * len(cases), []

    def work(i):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                for _ in range(3):                 # several rounds, so that the threads really overlap
                    got[i] = decode(T, model, vocab, cases[i][0], cases[i][1], 20, beam)[0]
        except Exception as e:                     # noqa: BLE001 - reported below, in the test's thread
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(cases))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors
    assert got == want


def test_run_on_streams_returns_results_in_order_and_reraises(T):
    """``tst_amd.run_on_streams``: parts of one batch decoded at the same time equal the whole batch; a job's exception reaches
    the caller; tensors made on the caller's stream before the call are complete inside the jobs."""
    tmodel, model, vocab, _, _ = mt_case(T, 300, 4, 9, 2.0, 6)
    r = np.random.RandomState(3)
    B, S = 96, 9
    lens = r.randint(1, S + 1, size=B)
    encs = r.standard_normal((B, S, 256)).astype(np.float32)
    mask = (np.arange(S)[None] < lens[:, None])[:, None, :]
    whole = decode(T, model, vocab, encs, mask, 18, 4)[0]
    scale = torch.full((1,), 1.0, device='cuda')
    e_dev = torch.from_numpy(encs).cuda() * scale          # produced on the caller's stream just before the call
    m_dev = torch.from_numpy(mask).cuda()

    def part(p):
        st = T.InferenceState({'model': model.decoder, 'inputs': e_dev[p::3].contiguous(), 'enc_mask': m_dev[p::3].contiguous(),
                               'vocab': vocab, 'max_len': 18, 'beam_size': 4})
        return [s.cpu().tolist() for s in st.inference()]

    parts = T.run_on_streams([(lambda p=p: part(p)) for p in range(3)])
    merged = [None] * B
    for p in range(3):
        merged[p::3] = parts[p]
    assert merged == whole

    def boom():
        raise ValueError('from a job')

    with pytest.raises(ValueError, match='from a job'):
        T.run_on_streams([lambda: 1, boom])
    assert T.run_on_streams([lambda: 7]) == [7] and T.run_on_streams([]) == []


def test_run_on_streams_reuses_its_streams_so_cached_memory_does_not_grow(T):
    """PyTorch caches freed blocks per stream: side streams that were new for every call reserved the jobs' buffers again every
    time (65 GB per call for a config-4 batch, until the device was full).  Repeated calls reserve once; results handed to the
    caller's stream stay valid while later calls reuse the side streams' blocks."""
    def job(i):
        a = torch.full((64 << 20,), float(i), device='cuda')        # 256 MB of work memory on the side stream
        return (a[:1000] * 2).clone(), {'sum': a[:10].sum()}

    reserved, kept = [], []
    for it in range(5):
        out = T.run_on_streams([(lambda i=i: job(i + 10 * it)) for i in range(3)])
        kept.append(out)
        torch.cuda.synchronize()
        reserved.append(torch.cuda.memory_reserved())
    assert reserved[4] == reserved[1], reserved
    for it, out in enumerate(kept):
        for i, (t, d) in enumerate(out):
            assert float(t[0]) == 2.0 * (i + 10 * it) and float(d['sum']) == 10.0 * (i + 10 * it)


def test_in_place_weight_edits_reach_the_kernels(T):
    """The kernels read converted copies of the weights (transposed, concatenated, per-token tables) that are made once.  A
    parameter edited in place after a decode (what an optimizer step over ``parameters()`` does between two validation passes,
    /root/reference/src/trainers/transformer_mt_trainer.py:80-98) must reach the next decode: outputs equal those of a fresh
    model loaded with the edited state."""
    tmodel, model, vocab, encs, mask = mt_case(T, 300, 40, 9, 2.0, 6)
    before, _ = decode(T, model, vocab, encs, mask, 12, 1)
    names = list(model.decoder._params.keys())
    params = dict(zip(names, model.decoder.parameters()))
    g = torch.Generator().manual_seed(1)
    for n in ('layers.0.self_attn.q.weight', 'layers.1.ff.w2.weight', 'out.weight'):
        params[n].add_(0.05 * torch.randn(params[n].shape, generator=g).cuda())           # in place: same tensor objects
    after, _ = decode(T, model, vocab, encs, mask, 12, 1)
    fresh = T.Transformer(T.HP(R.mt_hp()), vocab, vocab)
    fresh.load_state_dict(model.state_dict())
    want, _ = decode(T, fresh, vocab, encs, mask, 12, 1)
    assert after == want and after != before
    # a GRU decoder with the per-token table of layer 0 and a bf16 language model (copies per arithmetic type)
    tv = R.synthetic_vocab(90)
    vg = T.Vocab.synthetic(90)
    gru = T.RNNDecoder(64, 128, vg)
    gru.load_state_dict(R.synthetic_init(R.RNNDecoder(64, 128, tv), 3, 1.0, 2.0, seed=3).state_dict())
    z = torch.randn((30, 128), generator=torch.Generator().manual_seed(2)).cuda()

    def run(m, **kw):
        st = T.InferenceState(dict({'model': m, 'inputs': z, 'vocab': vg, 'max_len': 10}, **kw))
        return [s.cpu().tolist() for s in st.inference()]
    a = run(gru)
    dict(zip(gru._params.keys(), gru.parameters()))['embed.weight'].mul_(-1.0)
    b = run(gru)
    gru2 = T.RNNDecoder(64, 128, vg)
    gru2.load_state_dict(gru.state_dict())
    assert b == run(gru2) and a != b
    hp = dict(model_size=128, n_layers=2, n_heads=4, d_ff=256, pe_max_len=40)
    lm = T.TransformerLM(T.HP(hp), vg, dtype='bf16')
    lm.load_state_dict(R.synthetic_init(R.TransformerLM(R.HP(hp), tv), 3, 2.0, 2.0, seed=4).state_dict())
    a = run(lm)
    dict(zip(lm._params.keys(), lm.parameters()))['out.bias'][3] += 40.0                  # every sentence now ends at once
    assert all(len(s) == 1 for s in run(lm)) and any(len(s) > 1 for s in a)


@pytest.mark.parametrize('mode', ['inference_mode', 'no_grad'])
def test_whole_path_under_inference_mode_and_no_grad(T, mode):
    """Callers wrap validation in ``torch.no_grad()`` or ``torch.inference_mode()``; tensors made under the latter keep no
    version counter (reading it raises).  Building the model, loading weights, the encoder pass, the decode and an in-place
    edit of the encoder outputs between the two all work there and give the outputs of the plain call."""
    hp = R.mt_hp()
    tv = R.synthetic_vocab(200)
    tmodel = R.synthetic_init(R.Transformer(R.HP(hp), tv, tv), 3, 2.0, 2.0, seed=5)
    vocab = T.Vocab.synthetic(200)
    src = torch.from_numpy(np.random.RandomState(2).randint(4, 200, size=(23, 8))).cuda()

    def run(edit):
        model = T.Transformer(T.HP(hp), vocab, vocab)
        model.load_state_dict(tmodel.state_dict())
        encs, mask = model.encoder(src)
        if edit:
            encs.mul_(0.5)
        st = T.InferenceState({'model': model.decoder, 'inputs': encs, 'enc_mask': mask, 'vocab': vocab, 'max_len': 9, 'beam_size': 3})
        out = [s.cpu().tolist() for s in st.inference()]
        names = list(model.decoder._params.keys())
        dict(zip(names, model.decoder.parameters()))['out.bias'][3] += 40.0          # in place, under the same mode
        st = T.InferenceState({'model': model.decoder, 'inputs': encs, 'enc_mask': mask, 'vocab': vocab, 'max_len': 9})
        return out, [s.cpu().tolist() for s in st.inference()]
    plain = run(False), run(True)
    with getattr(torch, mode)():
        inside = run(False), run(True)
    assert inside == plain
    assert plain[0][0] != plain[1][0]                       # the edit of the encoder outputs changed the decode
    assert all(len(s) == 1 for s in plain[0][1])            # and the edited bias ended every sentence at once


@pytest.mark.parametrize('B,V', [(1024, 300), (1030, 256), (4095, 70), (4096, 513), (4100, 300), (4163, 1000)])
def test_projection_with_weights_in_registers_at_its_tile_and_size_boundaries(T, B, V):
    """The vocabulary projection of d_model-256 steps runs the weights-in-registers kernel from 1024 rows (32-row tiles) and its
    64-row form from 4096: row counts at and around both switches, with a last tile that is partial, and vocabulary sizes that
    are no multiple of the 256-column group (a partial last group, a single partial group): the logits of EVERY row of the
    first two steps equal the canonical oracle's bit for bit."""
    tmodel, model, vocab, encs, mask = mt_case(T, V, B, 3, 0.0, 11)
    st = T.InferenceState({'model': model.decoder, 'inputs': torch.from_numpy(encs).cuda(), 'enc_mask': torch.from_numpy(mask).cuda(),
                           'vocab': vocab, 'max_len': 2})
    out = [s.cpu().tolist() for s in st.inference(keep_logits=2)]
    sess = C.CanonTransformerDecoder(tmodel.decoder, 2).session(encs, mask, 1, 2)
    want, logits, rows = C.greedy_decode(sess, B, 2, 2, 3, 2)
    assert out == want
    for i in range(len(logits)):
        lg, sent, _ = st.loop.kept[i]
        assert np.array_equal(sent.cpu().numpy(), rows[i])
        assert np.array_equal(lg.cpu().numpy()[:, :V], logits[i]), (B, V, i)
