"""Generates the golden fixtures in this directory.  Run in the build container only:

    python tests/golden/make_golden.py

It imports the REAL decode loop, ``/root/reference/src/inference.py`` (``InferenceState``,
``simple_inference``), and drives it with the model specifications of ``oracle/torch_ref.py`` on CPU in
fp32.  The reference file needs ``firelab.utils.training_utils.cudable`` (a "move to GPU if there is
one" helper, ``/root/reference/src/inference.py:5``), a third-party package that is not installed; an
in-memory module whose ``cudable`` is the identity (its CPU behaviour) stands in for it.  Nothing of
the reference is copied: the fixtures hold inputs and outputs only.

Each ``<case>.npz`` holds: ``config`` (JSON: model kind, sizes, seeds, decode keys), the input arrays
(``encs``/``enc_mask`` or ``z``), the reference's output sequences (``tokens`` padded with -1,
``lens``) and, for the first steps, the last-position logits the model returned to the reference loop
(``logits_s<i>``) with the original batch positions of the rows of that step (``rows_s<i>``).

Cases whose name starts with ``beam_`` are NOT outputs of the reference (it has no beam search:
``/root/reference/README.md:27,30``); they are made by ``oracle.torch_ref.ref_beam_search``, the
torch statement of this build's own beam specification, and pin only the canonical oracle.
"""
import json
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from oracle import torch_ref as R


def import_reference():
    firelab = types.ModuleType('firelab')
    utils = types.ModuleType('firelab.utils')
    tu = types.ModuleType('firelab.utils.training_utils')
    tu.cudable = lambda x: x
    firelab.utils, utils.training_utils = utils, tu
    sys.modules.update({'firelab': firelab, 'firelab.utils': utils, 'firelab.utils.training_utils': tu})
    sys.path.insert(0, '/root/reference')
    from src.inference import InferenceState, simple_inference
    return InferenceState, simple_inference


class Recorder:
    """Wraps a model callable and records the last-position logits of the first ``keep`` calls."""

    def __init__(self, model, keep):
        self.model, self.keep, self.logits = model, keep, []

    def __call__(self, *a, **kw):
        out = self.model(*a, **kw)
        lg = out[0] if isinstance(out, tuple) else out
        if len(self.logits) < self.keep:
            self.logits.append(lg[:, -1].detach().clone().numpy())
        return out


def pack_tokens(seqs):
    lens = np.array([len(s) for s in seqs], dtype=np.int32)
    tok = np.full((len(seqs), int(lens.max())), -1, dtype=np.int32)
    for i, s in enumerate(seqs):
        tok[i, :len(s)] = np.asarray(s, dtype=np.int32)
    return tok, lens


def rows_per_step(seqs_tokens, eos_free_lens, n_steps):
    """Original batch positions still active at the start of each step (a row leaves after the step
    in which it emitted EOS, i.e. a sequence stored with length L was active for steps 0..L-1)."""
    rows = []
    for s in range(n_steps):
        rows.append(np.array([i for i, L in enumerate(eos_free_lens) if L - 1 >= s], dtype=np.int32))
    return rows


def save(name, config, arrays, seqs, rec, finished_lens):
    tok, lens = pack_tokens(seqs)
    out = dict(arrays)
    out['tokens'], out['lens'] = tok, lens
    rows = rows_per_step(seqs, finished_lens, len(rec.logits))
    for i, lg in enumerate(rec.logits):
        assert lg.shape[0] == len(rows[i]), (name, i, lg.shape, len(rows[i]))
        out['logits_s%d' % i] = lg.astype(np.float32)
        out['rows_s%d' % i] = rows[i]
    out['config'] = np.frombuffer(json.dumps(config).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print('%-16s B=%d lens=%s' % (name, len(seqs), lens.tolist()))


def finished_lens_of(seqs, max_len):
    """Number of steps each row was active: len(seq) for rows that emitted EOS (stored BOS-led and
    EOS-stripped, so len = steps), max_len for rows cut off (stored with 1 + max_len tokens)."""
    return [min(len(s), max_len) if len(s) <= max_len else max_len for s in seqs]


def transformer_case(InferenceState, name, V, B, S_max, max_len, eos_bias, out_scale, seed, through_encoder, keep=6):
    rs = np.random.RandomState(seed)
    vocab = R.synthetic_vocab(V)
    hp = R.mt_hp()
    model = R.synthetic_init(R.Transformer(hp, vocab, vocab), vocab.stoi['<eos>'], eos_bias, out_scale, seed=42)
    lens = rs.randint(4, S_max + 1, size=B)
    lens[0] = S_max
    arrays = {}
    if through_encoder:
        src = np.full((B, S_max), vocab.stoi['<pad>'], dtype=np.int64)
        for b in range(B):
            src[b, :lens[b]] = rs.randint(4, V, size=lens[b])
        with torch.no_grad():
            encs, enc_mask = model.encoder(torch.from_numpy(src))
        arrays['src'] = src.astype(np.int32)
    else:
        encs = torch.from_numpy(rs.standard_normal((B, S_max, hp.d_model)).astype(np.float32))
        enc_mask = torch.from_numpy(np.arange(S_max)[None, :] < lens[:, None]).unsqueeze(1)
    arrays['encs'] = encs.numpy()
    arrays['enc_mask'] = enc_mask.numpy()
    rec = Recorder(model.decoder, keep)
    with torch.no_grad():
        seqs = InferenceState({'model': rec, 'inputs': encs, 'enc_mask': enc_mask, 'vocab': vocab,
                               'max_len': max_len}).inference()
    seqs = [s.tolist() for s in seqs]
    config = dict(kind='transformer', V=V, hp=dict(hp), weight_seed=42, eos_bias=eos_bias, out_scale=out_scale,
                  max_len=max_len, data_seed=seed, through_encoder=through_encoder, source='reference src/inference.py')
    save(name, config, arrays, seqs, rec, finished_lens_of(seqs, max_len))
    return model, vocab, encs, enc_mask


def beam_case(name, encs, enc_mask, beam, alpha, max_len, base_config):
    vocab = R.synthetic_vocab(base_config['V'])
    model = R.synthetic_init(R.Transformer(R.HP(base_config['hp']), vocab, vocab), vocab.stoi['<eos>'],
                             base_config['eos_bias'], base_config['out_scale'], seed=base_config['weight_seed'])
    dec = model.decoder

    def step_fn(prefixes, alive):
        idx = torch.tensor([s for s in alive for _ in range(beam)])
        width = max(len(p) for p in prefixes)  # finished beams are shorter; their logits are unused
        ys = torch.tensor([p + [1] * (width - len(p)) for p in prefixes], dtype=torch.long)
        with torch.no_grad():
            return dec(encs[idx], ys, enc_mask[idx])[:, -1]

    res = R.ref_beam_search(step_fn, encs.shape[0], beam, max_len, vocab.stoi['<bos>'], vocab.stoi['<eos>'], alpha)
    tok, lens = pack_tokens([r[0] for r in res])
    cfg = dict(base_config, beam=beam, alpha=alpha, max_len=max_len, source='build-defined beam spec (torch)')
    np.savez_compressed(os.path.join(HERE, name + '.npz'), tokens=tok, lens=lens,
                        scores=np.array([r[1] for r in res], dtype=np.float32),
                        encs=encs.numpy(), enc_mask=enc_mask.numpy(),
                        config=np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8))
    print('%-16s B=%d lens=%s' % (name, len(res), lens.tolist()))


def gru_case(simple_inference, InferenceState, name, E, H, V, B, max_len, eos_token, eos_bias, out_scale, seed, keep=6):
    rs = np.random.RandomState(seed)
    vocab = R.synthetic_vocab(V, extra=('|',))
    model = R.synthetic_init(R.RNNDecoder(E, H, vocab), vocab.stoi[eos_token], eos_bias, out_scale, seed=42)
    z = torch.from_numpy(rs.standard_normal((B, H)).astype(np.float32))
    rec = Recorder(model, keep)
    with torch.no_grad():
        seqs = simple_inference(rec, z, vocab, max_len=max_len, eos_token=eos_token)
    config = dict(kind='gru', V=V, E=E, H=H, weight_seed=42, eos_bias=eos_bias, out_scale=out_scale, max_len=max_len,
                  eos_token=eos_token, data_seed=seed, source='reference src/inference.py simple_inference')
    save(name, config, {'z': z.numpy()}, seqs, rec, finished_lens_of(seqs, max_len))


def stateful_case(InferenceState, name, H, V, n_layers, B, max_len, eos_bias, out_scale, seed, keep=6):
    """The call pattern of ``/root/reference/src/prod/dialog_model.py:108-120`` with greedy selection
    (sampling there is random and has no fixed output)."""
    rs = np.random.RandomState(seed)
    vocab = R.synthetic_vocab(V, extra=('|',))
    model = R.synthetic_init(R.RNNLM(H, vocab, n_layers), vocab.stoi['|'], eos_bias, out_scale, seed=42)
    z = torch.from_numpy(rs.standard_normal((n_layers, B, H)).astype(np.float32))
    rec = Recorder(model, keep)
    with torch.no_grad():
        seqs = InferenceState({'model': rec, 'inputs': z, 'vocab': vocab, 'max_len': max_len, 'bos_token': '|',
                               'eos_token': '|', 'substitute_inputs': True, 'inputs_batch_dim': 1,
                               'kwargs': {'return_z': True}}).inference()
    seqs = [s.tolist() for s in seqs]
    config = dict(kind='rnnlm_stateful', V=V, H=H, n_layers=n_layers, weight_seed=42, eos_bias=eos_bias,
                  out_scale=out_scale, max_len=max_len, data_seed=seed, source='reference src/inference.py (stateful)')
    save(name, config, {'z': z.numpy()}, seqs, rec, finished_lens_of(seqs, max_len))


def tlm_case(InferenceState, name, hp, V, B, max_len, eos_bias, out_scale, seed, keep=6):
    """Transformer language model conditioned on z (``lm(z, x)``, /root/reference/src/trainers/lm_trainer.py:72-74)
    driven by the reference loop in its default (whole-prefix) mode."""
    rs = np.random.RandomState(seed)
    vocab = R.synthetic_vocab(V)
    model = R.synthetic_init(R.TransformerLM(R.HP(hp), vocab), vocab.stoi['<eos>'], eos_bias, out_scale, seed=42)
    z = torch.from_numpy(rs.standard_normal((B, hp['model_size'])).astype(np.float32))
    rec = Recorder(model, keep)
    with torch.no_grad():
        seqs = InferenceState({'model': rec, 'inputs': z, 'vocab': vocab, 'max_len': max_len}).inference()
    seqs = [s.tolist() for s in seqs]
    config = dict(kind='tlm', V=V, hp=dict(hp), weight_seed=42, eos_bias=eos_bias, out_scale=out_scale, max_len=max_len,
                  data_seed=seed, source='reference src/inference.py')
    save(name, config, {'z': z.numpy()}, seqs, rec, finished_lens_of(seqs, max_len))


def prefix_cases(InferenceState):
    """The real loop started from a caller-provided ``active_seqs`` of token ids (``/root/reference/src/inference.py:37``):
    a transformer decoder (whole-prefix mode) and a stateful GRU language model (which gets the whole prefix at its
    first step, ``:128``)."""
    rs = np.random.RandomState(110)
    vocab = R.synthetic_vocab(64)
    hp = R.mt_hp()
    model = R.synthetic_init(R.Transformer(hp, vocab, vocab), vocab.stoi['<eos>'], 1.5, 2.0, seed=42)
    B, S, max_len, t0 = 5, 9, 10, 3
    lens = rs.randint(3, S + 1, size=B)
    encs = torch.from_numpy(rs.standard_normal((B, S, hp.d_model)).astype(np.float32))
    enc_mask = torch.from_numpy(np.arange(S)[None, :] < lens[:, None]).unsqueeze(1)
    prefix = np.concatenate([np.full((B, 1), vocab.stoi['<bos>']), rs.randint(4, 64, size=(B, t0 - 1))], 1).astype(np.int64)
    rec = Recorder(model.decoder, 4)
    with torch.no_grad():
        seqs = InferenceState({'model': rec, 'inputs': encs, 'enc_mask': enc_mask, 'vocab': vocab, 'max_len': max_len,
                               'active_seqs': torch.from_numpy(prefix)}).inference()
    seqs = [s.tolist() for s in seqs]
    config = dict(kind='transformer', V=64, hp=dict(hp), weight_seed=42, eos_bias=1.5, out_scale=2.0, max_len=max_len,
                  prefix_len=t0, source='reference src/inference.py with active_seqs')
    save('prefix_tfm', config, {'encs': encs.numpy(), 'enc_mask': enc_mask.numpy(), 'prefix': prefix.astype(np.int32)}, seqs, rec,
         [min(len(s) - (t0 - 1), max_len) for s in seqs])

    vocab = R.synthetic_vocab(50, extra=('|',))
    lm = R.synthetic_init(R.RNNLM(64, vocab, 2), vocab.stoi['|'], 1.0, 2.0, seed=42)
    B, max_len, t0 = 4, 9, 4
    z = torch.from_numpy(rs.standard_normal((2, B, 64)).astype(np.float32))
    prefix = np.concatenate([np.full((B, 1), vocab.stoi['|']), rs.randint(5, 50, size=(B, t0 - 1))], 1).astype(np.int64)
    rec = Recorder(lm, 4)
    with torch.no_grad():
        seqs = InferenceState({'model': rec, 'inputs': z, 'vocab': vocab, 'max_len': max_len, 'bos_token': '|', 'eos_token': '|',
                               'substitute_inputs': True, 'inputs_batch_dim': 1, 'kwargs': {'return_z': True},
                               'active_seqs': torch.from_numpy(prefix)}).inference()
    seqs = [s.tolist() for s in seqs]
    config = dict(kind='rnnlm_stateful', V=50, H=64, n_layers=2, weight_seed=42, eos_bias=1.0, out_scale=2.0, max_len=max_len,
                  prefix_len=t0, source='reference src/inference.py (stateful) with active_seqs')
    save('prefix_rnnlm', config, {'z': z.numpy(), 'prefix': prefix.astype(np.int32)}, seqs, rec,
         [min(len(s) - (t0 - 1), max_len) for s in seqs])


def charlm_case(InferenceState):
    """The REAL ``CharLMFromEmbs`` (/root/reference/src/prod/throwaway_models.py:38-55) around a specification RNNLM:
    ``init_z`` from a style embedding, then the stateful loop of /root/reference/src/prod/dialog_model.py:97-120 (greedy)."""
    from src.prod.throwaway_models import CharLMFromEmbs
    import torch.nn as nn
    vocab = R.synthetic_vocab(50, extra=('|',))
    lm = R.synthetic_init(R.RNNLM(64, vocab, 2), vocab.stoi['|'], 0.2, 2.0, seed=42)
    rs = np.random.RandomState(111)
    emb = nn.Embedding(2, 64)
    with torch.no_grad():
        emb.weight.copy_(torch.from_numpy(rs.standard_normal((2, 64)).astype(np.float32)))
    model = CharLMFromEmbs(lm, emb, 2).eval()
    B, max_len = 4, 10
    text = torch.from_numpy(rs.randint(5, 50, size=(B, 6)))
    with torch.no_grad():
        z0 = model.init_z(B, 1)
        z = model(z0, text, return_z=True)[1]
        rec = Recorder(model, 4)
        seqs = InferenceState({'model': rec, 'inputs': z, 'vocab': vocab, 'max_len': max_len, 'bos_token': '|', 'eos_token': '|',
                               'substitute_inputs': True, 'inputs_batch_dim': 1, 'kwargs': {'return_z': True}}).inference()
    seqs = [s.tolist() for s in seqs]
    config = dict(kind='rnnlm_stateful', V=50, H=64, n_layers=2, weight_seed=42, eos_bias=0.2, out_scale=2.0, max_len=max_len,
                  source='reference CharLMFromEmbs + src/inference.py (stateful)')
    save('charlm_embs', config, {'style_embed': emb.weight.detach().numpy(), 'text': text.numpy().astype(np.int32),
                                 'z0': z0.numpy(), 'z': z.numpy()}, seqs, rec, finished_lens_of(seqs, max_len))


def conditional_case(InferenceState):
    """A style-conditioned GRU language model under the real loop in the call pattern of
    /root/reference/src/prod/dialog_model.py:98-120: context pass ``lm(zeros, text, style=1, return_z=True)[1]``, then the
    stateful loop with the style in ``kwargs`` (greedy: sampling there has no fixed output)."""
    vocab = R.synthetic_vocab(50, extra=('|',))
    lm = R.synthetic_init(R.ConditionalLM(64, vocab), vocab.stoi['|'], 0.8, 2.0, seed=42)
    rs = np.random.RandomState(112)
    B, max_len = 5, 10
    text = torch.from_numpy(rs.randint(5, 50, size=(B, 6)))
    with torch.no_grad():
        z = lm(torch.zeros(2, B, 64), text, style=1, return_z=True)[1]
        rec = Recorder(lm, 4)
        seqs = InferenceState({'model': rec, 'inputs': z, 'vocab': vocab, 'max_len': max_len, 'bos_token': '|', 'eos_token': '|',
                               'substitute_inputs': True, 'inputs_batch_dim': 1,
                               'kwargs': {'return_z': True, 'style': 1}}).inference()
        other = InferenceState({'model': lm, 'inputs': z, 'vocab': vocab, 'max_len': max_len, 'bos_token': '|', 'eos_token': '|',
                                'substitute_inputs': True, 'inputs_batch_dim': 1,
                                'kwargs': {'return_z': True, 'style': 0}}).inference()
    seqs = [s.tolist() for s in seqs]
    config = dict(kind='conditional_lm', V=50, H=64, n_layers=2, weight_seed=42, eos_bias=0.8, out_scale=2.0, max_len=max_len,
                  style=1, seqs_style0=[s.tolist() for s in other],
                  source='specification ConditionalLM + reference src/inference.py (stateful, style in kwargs)')
    save('conditional_lm', config, {'text': text.numpy().astype(np.int32), 'z': z.numpy()}, seqs, rec, finished_lens_of(seqs, max_len))


def ensemble_members(V, H, n_layers, seeds, eos_bias, out_scale):
    vocab = R.synthetic_vocab(V, extra=('|',))
    return vocab, [R.synthetic_init(R.RNNLM(H, vocab, n_layers), vocab.stoi['|'], eos_bias, out_scale, seed=sd) for sd in seeds]


def ensemble_case(InferenceState, name, V, H, n_layers, B, max_len, weights, seeds, eos_bias, out_scale, seed, keep=6):
    """The REAL ``WeightedLMEnsemble`` (/root/reference/src/prod/throwaway_models.py:9-36) over two GRU language models,
    driven by the real loop in the stateful mode of /root/reference/src/prod/dialog_model.py:103,108-120
    (``inputs_batch_dim`` 2), with greedy selection (sampling there has no fixed output)."""
    from src.prod.throwaway_models import WeightedLMEnsemble
    rs = np.random.RandomState(seed)
    vocab, members = ensemble_members(V, H, n_layers, seeds, eos_bias, out_scale)
    ens = WeightedLMEnsemble(members, weights).eval()
    zs = torch.from_numpy(rs.standard_normal((len(members), n_layers, B, H)).astype(np.float32))
    rec = Recorder(ens, keep)
    with torch.no_grad():
        seqs = InferenceState({'model': rec, 'inputs': zs, 'vocab': vocab, 'max_len': max_len, 'bos_token': '|',
                               'eos_token': '|', 'substitute_inputs': True, 'inputs_batch_dim': 2,
                               'kwargs': {'return_z': True}}).inference()
    seqs = [s.tolist() for s in seqs]
    config = dict(kind='ensemble', V=V, H=H, n_layers=n_layers, weights=weights, member_seeds=seeds, eos_bias=eos_bias,
                  out_scale=out_scale, max_len=max_len, data_seed=seed,
                  source='reference src/inference.py (stateful) + src/prod/throwaway_models.py WeightedLMEnsemble')
    save(name, config, {'zs': zs.numpy()}, seqs, rec, finished_lens_of(seqs, max_len))


def gumbel_case(InferenceState, name, kind, V, B, max_len, temperature, eos_bias, out_scale, seed, noise_seed, keep=4):
    """The gumbel branch of the reference loop (``/root/reference/src/inference.py:83-88,104-105,136-138``,
    ``/root/reference/src/utils/gumbel.py:11-21``): it runs when the caller provides ``active_seqs`` as
    distributions ``[B, 1, V]`` (one-hot BOS here) and a model that takes distributions
    (``oracle.torch_ref.SoftInput``).  The uniform draws of ``torch.rand`` are recorded per step so that the
    noise does not depend on a torch version; outputs are the ``[1, t * V]`` tensors the loop returns."""
    import src.utils.gumbel as G
    rs = np.random.RandomState(seed)
    arrays = {}
    if kind == 'gru':
        vocab = R.synthetic_vocab(V, extra=('|',))
        model = R.synthetic_init(R.RNNDecoder(32, 32, vocab), vocab.stoi['<eos>'], eos_bias, out_scale, seed=42)
        inputs = torch.from_numpy(rs.standard_normal((B, 32)).astype(np.float32))
        arrays['z'] = inputs.numpy()
        enc_mask = None
        core = model
        extra = dict(E=32, H=32)
    else:
        vocab = R.synthetic_vocab(V)
        hp = R.mt_hp()
        model = R.synthetic_init(R.Transformer(hp, vocab, vocab), vocab.stoi['<eos>'], eos_bias, out_scale, seed=42)
        S = 10
        lens = rs.randint(3, S + 1, size=B)
        inputs = torch.from_numpy(rs.standard_normal((B, S, hp.d_model)).astype(np.float32))
        enc_mask = torch.from_numpy(np.arange(S)[None, :] < lens[:, None]).unsqueeze(1)
        arrays['encs'], arrays['enc_mask'] = inputs.numpy(), enc_mask.numpy()
        core = model.decoder
        extra = dict(hp=dict(hp))
    rec = Recorder(R.SoftInput(core, V), keep)
    prefix = torch.zeros(B, 1, V)
    prefix[:, 0, vocab.stoi['<bos>']] = 1.0
    draws = []
    real_rand = torch.rand

    def recording_rand(*a, **kw):
        u = real_rand(*a, **kw)
        draws.append(u.clone().numpy())
        return u

    state = {'model': rec, 'inputs': inputs, 'vocab': vocab, 'max_len': max_len, 'gumbel': True,
             'temperature': temperature, 'active_seqs': prefix}
    if enc_mask is not None:
        state['enc_mask'] = enc_mask
    torch.manual_seed(noise_seed)
    torch.rand = recording_rand
    try:
        with torch.no_grad():
            seqs = InferenceState(state).inference()
    finally:
        torch.rand = real_rand
    lens_out = np.array([s.shape[1] // V for s in seqs], dtype=np.int32)
    dists = np.zeros((B, int(lens_out.max()), V), dtype=np.float32)
    for i, sq in enumerate(seqs):
        assert tuple(sq.shape) == (1, lens_out[i] * V)
        dists[i, :lens_out[i]] = sq.reshape(-1, V).numpy()
    out = dict(arrays, dists=dists, lens=lens_out)
    for i, u in enumerate(draws):
        out['u_s%d' % i] = u.astype(np.float32)
    for i, lg in enumerate(rec.logits):
        out['logits_s%d' % i] = lg.astype(np.float32)
    config = dict(kind='gumbel_' + kind, V=V, weight_seed=42, eos_bias=eos_bias, out_scale=out_scale, max_len=max_len,
                  temperature=temperature, data_seed=seed, noise_seed=noise_seed, n_steps=len(draws),
                  source='reference src/inference.py (gumbel=True) + src/utils/gumbel.py', **extra)
    out['config'] = np.frombuffer(json.dumps(config).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print('%-16s B=%d lens=%s steps=%d' % (name, B, lens_out.tolist(), len(draws)))


def gumbel_op_case():
    """Outputs of the reference's own ``gumbel_softmax_sample`` (``/root/reference/src/utils/gumbel.py:18-21``) on
    fixed inputs, with the uniform draws recorded."""
    import src.utils.gumbel as G
    rs = np.random.RandomState(11)
    out = {}
    real_rand = torch.rand
    for i, (rows, V, temp, scale) in enumerate(((5, 37, 1.0, 1.0), (3, 1000, 0.5, 3.0), (2, 4099, 2.0, 1.0))):
        logits = torch.from_numpy((scale * rs.standard_normal((rows, V))).astype(np.float32))
        draws = []

        def recording_rand(*a, **kw):
            u = real_rand(*a, **kw)
            draws.append(u.clone().numpy())
            return u

        torch.manual_seed(100 + i)
        torch.rand = recording_rand
        try:
            y = G.gumbel_softmax_sample(torch.softmax(logits, dim=1), temp)   # as src/inference.py:137-138 calls it
        finally:
            torch.rand = real_rand
        out['logits_%d' % i], out['u_%d' % i], out['y_%d' % i] = logits.numpy(), draws[0], y.numpy()
        out['temp_%d' % i] = np.float32(temp)
    np.savez_compressed(os.path.join(HERE, 'gumbel_op.npz'), **out)
    print('gumbel_op        3 cases')


def postprocess_case():
    """Outputs of the reference's own ``itos_many`` and ``compute_bleu_for_sents`` on synthetic sentences."""
    from src.utils.data_utils import itos_many
    from src.losses.bleu import compute_bleu, compute_bleu_for_sents
    rs = np.random.RandomState(7)
    words = ['<unk>', '<pad>', '<bos>', '<eos>'] + ['tok%d' % i for i in range(20)] + ['ab@@', 'cd@@', 'ef']
    vocab = R.RefVocab(words[4:])
    seqs = []
    for _ in range(12):
        n = rs.randint(0, 9)
        body = rs.randint(4, len(vocab), size=n).tolist()
        seqs.append([2] + body + ([3] if rs.rand() < 0.5 else []) + [1] * rs.randint(0, 3))
    cases = {'vocab': vocab.itos, 'seqs': seqs, 'itos_default': itos_many(seqs, vocab),
             'itos_keep_special': itos_many(seqs, vocab, remove_special=False), 'itos_chars': itos_many(seqs, vocab, sep='')}
    hyps, refs = [], []
    for _ in range(10):
        n = rs.randint(1, 12)
        ref = ['w%d' % i for i in rs.randint(0, 6, size=n)]
        hyp = [w if rs.rand() < 0.7 else 'w%d' % rs.randint(0, 6) for w in ref][:rs.randint(1, n + 1)] + \
              ['w%d' % i for i in rs.randint(0, 6, size=rs.randint(0, 3))]
        hyps.append(' '.join(hyp))
        refs.append(' '.join(ref))
    cases['bleu'] = dict(hyps=hyps, refs=refs, plain=compute_bleu_for_sents(hyps, refs),
                         smooth=compute_bleu_for_sents(hyps, refs, smooth=True),
                         order2=compute_bleu_for_sents(hyps, refs, max_order=2),
                         full=list(compute_bleu([[r.split()] for r in refs], [h.split() for h in hyps])),
                         per_sentence=[compute_bleu_for_sents([h], [r], smooth=True) for h, r in zip(hyps, refs)],
                         multi_ref=list(compute_bleu([[refs[0].split(), refs[1].split()], [refs[2].split()]],
                                                     [hyps[0].split(), hyps[2].split()])))
    with open(os.path.join(HERE, 'postprocess.json'), 'w') as f:
        json.dump(cases, f, indent=1)
    print('postprocess      bleu=%.4f smooth=%.4f' % (cases['bleu']['plain'], cases['bleu']['smooth']))


def chain_case(InferenceState):
    """The validation chain of ``/root/reference/src/trainers/transformer_mt_trainer.py:91-101``: encoder pass, the REAL decode
    loop, the REAL ``itos_many`` on predictions and targets, the REAL ``compute_bleu_for_sents``.  The vocabulary carries
    BPE continuation markers (every 5th word ends in ``@@``) so that the un-BPE step has something to do; targets are the
    predictions with a third of the words replaced, so that BLEU is neither 0 nor 1."""
    from src.utils.data_utils import itos_many
    from src.losses.bleu import compute_bleu_for_sents
    V, B, S_max, max_len, eos_bias, out_scale, seed = 500, 24, 14, 18, 2.5, 2.0, 110
    words = [('w%d@@' % i if i % 5 == 0 else 'w%d' % i) for i in range(V - 4)]
    vocab = R.RefVocab(words)
    hp = R.mt_hp()
    model = R.synthetic_init(R.Transformer(hp, vocab, vocab), vocab.stoi['<eos>'], eos_bias, out_scale, seed=42)
    rs = np.random.RandomState(seed)
    lens = rs.randint(4, S_max + 1, size=B)
    src = np.full((B, S_max), vocab.stoi['<pad>'], dtype=np.int64)
    for b in range(B):
        src[b, :lens[b]] = rs.randint(4, V, size=lens[b])
    with torch.no_grad():
        encs, enc_mask = model.encoder(torch.from_numpy(src))
        preds = InferenceState({'model': model.decoder, 'inputs': encs, 'enc_mask': enc_mask, 'vocab': vocab,
                                'max_len': max_len}).inference()
    preds = [p.tolist() for p in preds]
    width = max(len(p) for p in preds) + 2
    trg = np.full((B, width), vocab.stoi['<pad>'], dtype=np.int64)
    for b, p in enumerate(preds):
        body = [t if rs.rand() < 0.67 else int(rs.randint(4, V)) for t in p[1:]]
        row = [vocab.stoi['<bos>']] + body + [vocab.stoi['<eos>']]
        trg[b, :len(row)] = row
    pred_str, trg_str = itos_many(preds, vocab), itos_many(trg, vocab)
    case = dict(config=dict(kind='transformer', V=V, hp=dict(hp), weight_seed=42, eos_bias=eos_bias, out_scale=out_scale,
                            max_len=max_len, data_seed=seed, source='reference src/inference.py + data_utils.itos_many + bleu.py'),
                words=words, src=src.tolist(), trg=trg.tolist(), preds=preds, pred_str=pred_str, trg_str=trg_str,
                bleu=compute_bleu_for_sents(pred_str, trg_str), bleu_smooth=compute_bleu_for_sents(pred_str, trg_str, smooth=True))
    with open(os.path.join(HERE, 'chain_tfm.json'), 'w') as f:
        json.dump(case, f)
    print('chain_tfm        B=%d bleu=%.4f lens=%s' % (B, case['bleu'], [len(p) for p in preds]))


def main():
    torch.set_num_threads(4)
    InferenceState, simple_inference = import_reference()
    if sys.argv[1:] == ['conditional']:      # only the newest case (the others are unchanged)
        conditional_case(InferenceState)
        return
    if sys.argv[1:] == ['chain']:
        chain_case(InferenceState)
        return
    postprocess_case()
    m, v, e, mk = transformer_case(InferenceState, 'tfm_v64', V=64, B=6, S_max=12, max_len=16, eos_bias=1.5,
                                   out_scale=2.0, seed=101, through_encoder=False)
    base = dict(kind='transformer', V=64, hp=dict(R.mt_hp()), weight_seed=42, eos_bias=0.25, out_scale=2.0)
    beam_case('beam_tfm_v64_k4', e, mk, beam=4, alpha=0.0, max_len=12, base_config=base)
    beam_case('beam_tfm_v64_k3a', e, mk, beam=3, alpha=0.6, max_len=12, base_config=base)
    transformer_case(InferenceState, 'tfm_v1000', V=1000, B=8, S_max=14, max_len=20, eos_bias=2.2,
                     out_scale=2.0, seed=102, through_encoder=True)
    gru_case(simple_inference, InferenceState, 'gru_h32', E=32, H=32, V=40, B=5, max_len=12, eos_token='<eos>',
             eos_bias=1.0, out_scale=2.0, seed=103)
    gru_case(simple_inference, InferenceState, 'gru_h1024', E=1024, H=1024, V=128, B=4, max_len=24, eos_token='|',
             eos_bias=0.5, out_scale=2.0, seed=104)
    tlm_case(InferenceState, 'tlm_d64', dict(model_size=64, n_layers=2, n_heads=4, d_ff=128, pe_max_len=32), V=300, B=6,
             max_len=14, eos_bias=4.0, out_scale=2.0, seed=106)
    stateful_case(InferenceState, 'rnnlm_stateful', H=64, V=50, n_layers=2, B=4, max_len=14, eos_bias=1.0,
                  out_scale=2.0, seed=105)
    gumbel_op_case()
    prefix_cases(InferenceState)
    charlm_case(InferenceState)
    conditional_case(InferenceState)
    chain_case(InferenceState)
    ensemble_case(InferenceState, 'ensemble_rnnlm', V=50, H=32, n_layers=2, B=5, max_len=12, weights=[0.25, 0.75],
                  seeds=[42, 43], eos_bias=1.0, out_scale=2.0, seed=109)
    gumbel_case(InferenceState, 'gumbel_gru', 'gru', V=40, B=5, max_len=10, temperature=0.7, eos_bias=1.0, out_scale=2.0,
                seed=107, noise_seed=7)
    gumbel_case(InferenceState, 'gumbel_tfm', 'tfm', V=64, B=4, max_len=9, temperature=1.0, eos_bias=1.5, out_scale=2.0,
                seed=108, noise_seed=8)


if __name__ == '__main__':
    main()
