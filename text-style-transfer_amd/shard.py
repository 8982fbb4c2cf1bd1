"""Data-parallel decoding over the GPUs of one node: one process per GPU, sentences partitioned into
independent shards, no collective inside the decode loop, one all-gather of the outputs at the end
(``torch.distributed``; the ``nccl`` backend is RCCL over xGMI on ROCm).

Sentences are sorted by source length and dealt to the ranks round-robin (rank r owns the sorted positions r, r + N,
r + 2N, ...): every rank sees the same length distribution, hence the same amount of work (the metric divides by the
slowest rank), and its own shard is still sorted, so the on-GPU batches it walks hold sentences of similar length
(little padding, similar decode lengths).  The gathered outputs are put back into the caller's order.  The weights are
replicated.  The reference's only multi-GPU code is ``nn.DataParallel`` around a training model
(``/root/reference/src/trainers/dissonet_trainer.py:76-78``).
"""
import torch
import torch.distributed as dist


def shard_plan(src_lens, world_size):
    """Returns (order, per_rank): ``order`` sorts sentences by length (stable); rank r owns ``order[r::world_size]``,
    at most ``per_rank`` sentences (the gather pads every shard to that many rows)."""
    lens = torch.as_tensor(src_lens)
    order = torch.sort(lens, stable=True).indices
    per_rank = (len(lens) + world_size - 1) // world_size
    return order, per_rank


def my_shard(order, world_size, rank):
    return order[rank::world_size]


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
    # row j of rank r is sorted position j * world + r; the padding rows are exactly the positions >= n_total
    out = out.view(world, per_rank, W + 1).transpose(0, 1).reshape(world * per_rank, W + 1)[:n_total]
    unperm = torch.empty_like(out)
    unperm[order.to(out.device)] = out
    return unperm[:, :W], unperm[:, W]


def unpermute(tok, lens, order):
    """Outputs of the length-sorted sentences ``order`` (one rank, nothing to gather) back in the caller's order."""
    idx = order.to(tok.device)
    out_t, out_l = torch.empty_like(tok), torch.empty_like(lens)
    out_t[idx] = tok
    out_l[idx] = lens
    return out_t, out_l


def decode_sharded(decode_fn, src, src_lens, batch_size, group=None, trim=False):
    """Decodes ``src`` [N, S] across the ranks of ``group``.  ``decode_fn(src_batch) -> (tok [n, W] int32,
    lens [n] int32)`` decodes one on-GPU batch (e.g. encoder + ``InferenceState``); every rank walks its
    own shard in batches of ``batch_size``.  With ``trim`` a batch is cut to the columns of its longest sentence
    (``src_lens`` counts the non-padding tokens of a right-padded row).  Returns the outputs of all N sentences in
    input order, on every rank."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    order, per_rank = shard_plan(src_lens, world)
    mine = my_shard(order, world, rank)
    tok, ln = decode_shard(decode_fn, src, mine, batch_size, torch.as_tensor(src_lens) if trim else None)
    return gather_outputs(tok, ln, src.shape[0], order, per_rank, group)


def decode_shard(decode_fn, src, mine, batch_size, src_lens=None):
    """The walk of one rank over its shard ``mine`` (indices into ``src``); ``src`` may hold the whole corpus or, with
    ``mine`` = None, the rank's own rows already in walking order.  ``src_lens`` (host tensor indexed like ``src``)
    enables trimming each batch to its longest sentence."""
    n = src.shape[0] if mine is None else len(mine)
    toks, lens = [], []
    for i in range(0, n, batch_size):
        sel = slice(i, i + batch_size) if mine is None else mine[i:i + batch_size]
        batch = src[sel]
        if src_lens is not None:
            batch = batch[:, :int(src_lens[sel].max())]
        t, l = decode_fn(batch)
        toks.append(t)
        lens.append(l)
    if toks:
        return torch.cat(toks), torch.cat(lens)
    probe_t, probe_l = decode_fn(src[:1])   # an empty shard still takes part in the gather
    return probe_t[:0], probe_l[:0]
