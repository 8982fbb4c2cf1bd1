"""CPU, two processes over gloo: the sharding / all-gather / un-permute logic of the multi-GPU path.
The decode function is a stand-in (a pure function of each sentence), so that the expected result of
every sentence is known regardless of which rank decoded it."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import ROOT


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def fake_decode(src, W=48):
    """'decodes' each sentence to its content tokens reversed; int32 [n, W] + lengths (W fixed: batches may be trimmed)."""
    n, S = src.shape
    lens = (src != 1).sum(1).to(torch.int32)
    out = torch.zeros((n, W), dtype=torch.int32)
    for i in range(n):
        out[i, :lens[i]] = src[i, :lens[i]].flip(0).to(torch.int32)
    return out, lens


def _worker(rank, world, port, n_sent, batch, result_dir, trim=False):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import importlib
    shard = importlib.import_module('text-style-transfer_amd.shard')
    from tst_amd import synthetic
    src = synthetic.multi30k_like_sources(n_sent, 100, seed=5)
    lens = (src != 1).sum(1)
    tok, ln = shard.decode_sharded(fake_decode, src, lens, batch, trim=trim)
    torch.save((tok, ln), os.path.join(result_dir, 'r%d.pt' % rank))
    dist.destroy_process_group()


@pytest.mark.parametrize('n_sent,batch,trim', [(37, 8, False), (64, 100, False), (3, 2, False), (1, 4, False), (53, 8, True)])
def test_two_rank_sharded_decode_equals_single_process(tmp_path, n_sent, batch, trim):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, n_sent, batch, str(tmp_path), trim), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    from tst_amd import synthetic
    src = synthetic.multi30k_like_sources(n_sent, 100, seed=5)
    want_tok, want_len = fake_decode(src)
    for r in range(world):
        tok, ln = torch.load(os.path.join(str(tmp_path), 'r%d.pt' % r))
        assert torch.equal(ln, want_len)
        assert torch.equal(tok, want_tok)


def test_shard_plan_deals_the_length_sorted_sentences_round_robin():
    import importlib
    sys.path.insert(0, ROOT)
    shard = importlib.import_module('text-style-transfer_amd.shard')
    lens = torch.tensor([5, 3, 9, 3, 7, 1, 8])
    order, per = shard.shard_plan(lens, 3)
    assert per == 3 and sorted(order.tolist()) == list(range(7))
    parts = [shard.my_shard(order, 3, r) for r in range(3)]
    assert [len(p) for p in parts] == [3, 2, 2]
    assert sorted(torch.cat(parts).tolist()) == list(range(7))
    for p in parts:   # every shard is sorted by length, and the shards interleave: similar length distributions
        assert lens[p].tolist() == sorted(lens[p].tolist())
    assert lens[parts[0]].tolist() == [1, 5, 9] and lens[parts[1]].tolist() == [3, 7] and lens[parts[2]].tolist() == [3, 8]
    big = torch.randint(4, 41, (10001,), generator=torch.Generator().manual_seed(0))
    order, per = shard.shard_plan(big, 8)
    sums = [int(big[shard.my_shard(order, 8, r)].sum()) for r in range(8)]
    assert max(sums) - min(sums) <= 40 * 2   # the ranks' total source lengths differ by less than two sentences
