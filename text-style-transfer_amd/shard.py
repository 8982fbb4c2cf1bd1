"""Data-parallel decoding over the GPUs of one node: one process per GPU, sentences partitioned into
independent shards, no collective inside the decode loop, one all-gather of the outputs at the end
(``torch.distributed``; the ``nccl`` backend is RCCL over xGMI on ROCm).

Sentences are sorted by source length and dealt to the ranks in contiguous equal-count blocks, so a
shard holds sentences of similar length (less padding, similar decode lengths); the gathered outputs are
put back into the caller's order.  The weights are replicated.  The reference's only multi-GPU code is
``nn.DataParallel`` around a training model (``/root/reference/src/trainers/dissonet_trainer.py:76-78``).
"""
import torch
import torch.distributed as dist


def shard_plan(src_lens, world_size):
    """Returns (order, per_rank): ``order`` sorts sentences by length (stable); rank r owns
    ``order[r * per_rank : (r + 1) * per_rank]`` (the last shard may be short)."""
    lens = torch.as_tensor(src_lens)
    order = torch.sort(lens, stable=True).indices
    per_rank = (len(lens) + world_size - 1) // world_size
    return order, per_rank


def my_shard(order, per_rank, rank):
    return order[rank * per_rank:(rank + 1) * per_rank]


def gather_outputs(tok, lens, n_total, order, per_rank, group=None):
    """All-gathers the per-rank outputs (``tok`` [n_local, W] int32, ``lens`` [n_local] int32; shards padded
    to ``per_rank`` rows) and returns them in the original sentence order: ([n_total, W], [n_total])."""
    world = dist.get_world_size(group)
    W = tok.shape[1]
    pad_rows = per_rank - tok.shape[0]
    if pad_rows:
        tok = torch.cat([tok, tok.new_zeros((pad_rows, W))])
        lens = torch.cat([lens, lens.new_zeros(pad_rows)])
    packed = torch.cat([tok, lens.unsqueeze(1)], dim=1).contiguous()
    out = torch.empty((world * per_rank, W + 1), dtype=packed.dtype, device=packed.device)
    dist.all_gather_into_tensor(out, packed, group=group)
    out = out[:n_total]
    unperm = torch.empty_like(out)
    unperm[order.to(out.device)] = out
    return unperm[:, :W], unperm[:, W]


def decode_sharded(decode_fn, src, src_lens, batch_size, group=None):
    """Decodes ``src`` [N, S] across the ranks of ``group``.  ``decode_fn(src_batch) -> (tok [n, W] int32,
    lens [n] int32)`` decodes one on-GPU batch (e.g. encoder + ``InferenceState``); every rank walks its
    own shard in batches of ``batch_size``.  Returns the outputs of all N sentences in input order, on
    every rank."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    order, per_rank = shard_plan(src_lens, world)
    mine = my_shard(order, per_rank, rank)
    toks, lens = [], []
    for i in range(0, len(mine), batch_size):
        t, l = decode_fn(src[mine[i:i + batch_size]])
        toks.append(t)
        lens.append(l)
    if toks:
        tok, ln = torch.cat(toks), torch.cat(lens)
    else:  # an empty shard still takes part in the gather
        probe_t, probe_l = decode_fn(src[:1])
        tok, ln = probe_t[:0], probe_l[:0]
    return gather_outputs(tok, ln, src.shape[0], order, per_rank, group)
