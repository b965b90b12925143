"""Query-batch sharding across the GPUs of one node.

The path partitions by independent units (queries never interact, search_function.h:348); the
index is replicated in every GPU's HBM.  One process per GPU (torch.distributed, backend "nccl" =
RCCL over xGMI; "gloo" in the CPU tests).  The only exchange step is the gather of the uint32
answer ids -- 4 bytes per query.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_q, world, rank):
    """Contiguous block [lo, hi) of rank `rank`; sizes differ by at most one."""
    base, extra = divmod(n_q, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_pad(n_q, world):
    """Size of the largest block: every rank pads its answers to this many entries for the all-gather."""
    return (n_q + world - 1) // world


def gather_ids(local_ids, n_q, group=None):
    """All-gathers per-rank answer blocks (int32 tensors, block r = shard_bounds(n_q, W, r)) into
    the full [n_q] vector on every rank.  Blocks are padded to the largest shard so a single
    all_gather_into_tensor moves everything."""
    world = dist.get_world_size(group)
    if world == 1:
        return local_ids
    rank = dist.get_rank(group)
    width = shard_pad(n_q, world)
    lo, hi = shard_bounds(n_q, world, rank)
    assert local_ids.numel() == hi - lo
    send = torch.full((width,), -1, dtype=local_ids.dtype, device=local_ids.device)
    send[:hi - lo] = local_ids
    recv = torch.empty(world * width, dtype=local_ids.dtype, device=local_ids.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    parts = []
    for r in range(world):
        a, b = shard_bounds(n_q, world, r)
        parts.append(recv[r * width:r * width + (b - a)])
    return torch.cat(parts)


def sharded_search(search_fn, queries, n_q=None, group=None):
    """Runs `search_fn(query_block) -> int32 ids` on this rank's block and gathers the answers.
    `queries` is the full batch (replicated) or a callable lo,hi -> block."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n_q = queries.shape[0] if n_q is None else n_q
    lo, hi = shard_bounds(n_q, world, rank)
    block = queries(lo, hi) if callable(queries) else queries[lo:hi]
    ids = search_fn(block)
    if world == 1:
        return ids
    return gather_ids(ids, n_q, group)
