"""GPU: the multi-GPU code path with one rank on the real device: RCCL (``nccl`` backend) process group, shard plan,
walk over on-GPU batches, all-gather of device tensors, un-permutation.  A miniature of BASELINE config 5 (a corpus
decoded in on-GPU batches); the outputs must equal the decode of the whole corpus as one batch, sentence by sentence."""
import importlib
import os

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
