"""GPU: the multi-GPU code path with one rank on the real device: RCCL (``nccl`` backend) process group, shard plan,
walk over on-GPU batches, all-gather of device tensors, un-permutation.  A miniature of BASELINE config 5 (a corpus
decoded in on-GPU batches); the outputs must equal the decode of the whole corpus as one batch, sentence by sentence."""
import importlib
import json
import os
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def test_sharded_corpus_decode_one_rank_rccl(tmp_path):
    import tst_amd
    from tst_amd import synthetic
    shard = importlib.import_module('text-style-transfer_amd.shard')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', init_method='file://%s' % (tmp_path / 'store'), rank=0, world_size=1,
                            device_id=torch.device('cuda', 0))
    try:
        V, N, batch, beam = 512, 1000, 384, 4
        hp = dict(d_model=256, dropout=0.1, n_layers=2, pe_max_len=60, n_heads=2, d_ff=1024)
        vocab = tst_amd.Vocab.synthetic(V)
        model = tst_amd.Transformer(tst_amd.HP(hp), vocab, vocab, device='cuda:0')
        model.load_state_dict(synthetic.synthetic_state_dict(model, 3, 3.0, 2.0, seed=42))
        src = synthetic.multi30k_like_sources(N, V, seed=77)
        lens = (src != 1).sum(1)

        def decode(src_batch):
            encs, mask = model.encoder(src_batch.cuda())
            st = tst_amd.InferenceState({'model': model.decoder, 'inputs': encs, 'enc_mask': mask, 'vocab': vocab,
                                         'max_len': 20, 'beam_size': beam})
            tok, ln, _ = st.decode_packed()
            return tok, ln

        tok, ln = shard.decode_sharded(decode, src, lens, batch)
        assert tok.is_cuda and tuple(tok.shape) == (N, 21) and tuple(ln.shape) == (N,)
        whole_tok, whole_ln = decode(src)                      # the corpus as one batch, in input order
        assert torch.equal(ln, whole_ln)
        for i in range(N):
            n = int(ln[i])
            assert torch.equal(tok[i, :n], whole_tok[i, :n]), i
    finally:
        dist.destroy_process_group()


def test_bench_config5_two_ranks_equal_one_rank(tmp_path):
    """``bench.py --config 5`` (a corpus dealt to the ranks, walked in on-GPU batches, all-gather, un-permute) started the way the
    driver starts it, ``python bench.py --gpus N``: the 2-rank run (both ranks on this one GPU, gloo rehearsal backend; the
    parent starts the rank processes itself) returns byte-identical outputs to the 1-rank run, in input order, and the same
    checksum in its JSON line."""
    from helpers import ROOT
    bench = os.path.join(ROOT, 'bench.py')
    common = ['--config', '5', '--sentences', '3000', '--batch', '640', '--steps', '1', '--warmup', '0', '--no-cpu-baseline',
              '--no-roofline']
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    lines = {}
    for n, extra in ((1, []), (2, ['--share-gpu', '--backend', 'gloo'])):
        dump = str(tmp_path / ('out_n%d.pt' % n))
        r = subprocess.run([sys.executable, bench, '--gpus', str(n), '--dump', dump] + common + extra, capture_output=True, text=True,
                           env=env, timeout=280)
        assert r.returncode == 0, r.stderr[-2000:]
        lines[n] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    a, b = torch.load(str(tmp_path / 'out_n1.pt')), torch.load(str(tmp_path / 'out_n2.pt'))
    assert tuple(a[0].shape) == (3000, 51) and torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert lines[1]['output_checksum'] == lines[2]['output_checksum']
    assert lines[2]['n_gpus'] == 2 and lines[2]['config']['ranks_seen_by_backend'] == 2 and lines[2]['scaling'] == 'strong'
    assert lines[1]['config']['batches_per_step_rank0'] == 5 and lines[2]['config']['batches_per_step_rank0'] == 3
    # what makes a scaling curve readable: every rank's own time over the timed region and its time in the gather
    assert lines[1]['rank_ms'] is None and lines[1]['gather_ms'] is None
    assert len(lines[2]['rank_ms']) == 2 and len(lines[2]['gather_ms']) == 2
    assert all(0 < g < r for g, r in zip(lines[2]['gather_ms'], lines[2]['rank_ms']))
    assert abs(max(lines[2]['rank_ms']) - lines[2]['ms_per_step']) < 1e-6 * lines[2]['ms_per_step']
    # against the oracle, not only against the one-rank run: 32 sampled sentences of the corpus (the benchmark's model and
    # corpus, rebuilt here from their seeds) decoded by the canonical CPU oracle equal the sharded run's rows bit for bit
    import numpy as np
    import bench as B
    from oracle import canon as C
    from oracle import torch_ref as R
    from tst_amd import synthetic
    tv = R.synthetic_vocab(B.VOCAB)
    tmodel = R.synthetic_init(R.Transformer(R.HP(B.HP), tv, tv), 3, 5.5, B.OUT_SCALE, seed=42)
    pick = np.sort(np.random.RandomState(21).choice(3000, size=32, replace=False))
    src = synthetic.multi30k_like_sources(3000, B.VOCAB, seed=1240, rows=pick)
    encs, mask = C.CanonTransformerEncoder(tmodel.encoder, B.HP['n_heads'])(src.numpy())
    sess = C.CanonTransformerDecoder(tmodel.decoder, B.HP['n_heads']).session(encs, mask, 4, B.MAX_LEN)
    want = C.beam_decode(sess, 32, 4, B.MAX_LEN, 2, 3, 0.0)[0]
    tok2, len2 = b[0].numpy(), b[1].numpy()
    assert [tok2[i, :len2[i]].tolist() for i in pick] == want


def test_batch_trimmed_to_its_longest_sentence_decodes_identically():
    """The corpus walk cuts every batch to the columns of its longest sentence; the padding columns carry no information, so
    the outputs must not change (tokens, lengths, scores)."""
    import tst_amd
    from tst_amd import synthetic
    V = 512
    hp = dict(d_model=256, dropout=0.1, n_layers=2, pe_max_len=60, n_heads=2, d_ff=1024)
    vocab = tst_amd.Vocab.synthetic(V)
    model = tst_amd.Transformer(tst_amd.HP(hp), vocab, vocab, device='cuda:0')
    model.load_state_dict(synthetic.synthetic_state_dict(model, 3, 3.0, 2.0, seed=42))
    src = synthetic.multi30k_like_sources(300, V, seed=78)
    lens = (src != 1).sum(1)
    short = src[lens <= 14][:, :]
    assert short.shape[0] > 50 and short.shape[1] > 16

    def decode(batch):
        encs, mask = model.encoder(batch.cuda())
        st = tst_amd.InferenceState({'model': model.decoder, 'inputs': encs, 'enc_mask': mask, 'vocab': vocab, 'max_len': 20,
                                     'beam_size': 4})
        return st.decode_packed()

    wide = decode(short)
    narrow = decode(short[:, :14])
    for a, b in zip(wide, narrow):
        assert torch.equal(a, b)
