"""Query sharding across the GPUs of one node.

Every query's result depends only on the read-only reference cloud, so the path shards by
queries with no collective inside the search: the reference cloud is broadcast once
(torch.distributed: RCCL over xGMI on GPUs, gloo in the CPU tests), each rank searches a
contiguous shard, results stay sharded (or are gathered when a consumer needs them).
torch.distributed is plumbing here; nothing in this module computes a search.
"""
from __future__ import annotations


def shard_range(n_total: int, rank: int, world: int):
    """contiguous shard [start, start+count) of n_total queries for `rank`; the first
    n_total % world ranks take one extra query."""
    base, extra = divmod(n_total, world)
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


def broadcast_cloud(tensor, dist=None, src: int = 0):
    """one broadcast of the reference cloud from rank `src` (RCCL/xGMI with the nccl backend)."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(tensor, src=src)
    return tensor


def sharded_search(search, queries, rank: int, world: int):
    """run `search(shard) -> (idx, d2)` on this rank's shard; returns (start, idx, d2)."""
    start, count = shard_range(len(queries), rank, world)
    idx, d2 = search(queries[start:start + count])
    return start, idx, d2


def gather_shards(idx, d2, n_total: int, dist, device=None):
    """all_gather the shard results into full-length tensors (only when a consumer needs the
    concatenation; the bench keeps results sharded)."""
    import torch
    world = dist.get_world_size()
    counts = [shard_range(n_total, r, world)[1] for r in range(world)]
    mx = max(counts)
    pad_i = torch.full((mx,), -1, dtype=torch.int32, device=device)
    pad_d = torch.full((mx,), float("inf"), dtype=torch.float32, device=device)
    pad_i[:len(idx)] = torch.as_tensor(idx)
    pad_d[:len(d2)] = torch.as_tensor(d2)
    gi = [torch.empty_like(pad_i) for _ in range(world)]
    gd = [torch.empty_like(pad_d) for _ in range(world)]
    dist.all_gather(gi, pad_i)
    dist.all_gather(gd, pad_d)
    return (torch.cat([g[:c] for g, c in zip(gi, counts)]), torch.cat([g[:c] for g, c in zip(gd, counts)]))
