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


def _mat4_mul_f32(a, b):
    """4x4 float product with pcc_icp_align's accumulation order: acc = ((0 + a0 b0) + a1 b1) + a2 b2) + a3 b3"""
    import numpy as np
    out = np.zeros((4, 4), dtype=np.float32)
    for r in range(4):
        for c in range(4):
            acc = np.float32(0)
            for k in range(4):
                acc = np.float32(acc + np.float32(a[r, k] * b[k, c]))
            out[r, c] = acc
    return out


def icp_align_sharded(step, transform, solve, src_shard, max_iter: int, dist=None, fixed: bool = True):
    """ICP with the SOURCE cloud sharded over the ranks and the target index replicated: each
    rank finds the correspondences of its shard, the 17 double sums are all-reduced (the path's one real
    exchange: 136 bytes per iteration over RCCL), every rank solves the same transform and moves its shard.

    step(points) -> sums[17] (numpy float64; capi.Index.icp_step(points, want_corr=False)[2]),
    transform(T, points) -> points (capi.Index.transform), solve(sums) -> 4x4 (capi.rigid_from_sums).
    For a cloud at large coordinates bind the SAME centre on every rank in both callables
    (icp_step(..., center=c) and rigid_from_sums(sums, center=c)): sums about the origin cancel there.
    Returns (T_total 4x4 float32, iterations, mean squared distance of the last pass over ALL shards)."""
    import numpy as np
    import torch
    T = np.eye(4, dtype=np.float32)
    cur = src_shard
    prev = float("inf")
    it, mse = 0, float("inf")
    while it < max_iter:
        sums = np.asarray(step(cur), dtype=np.float64)
        if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
            t = torch.from_numpy(sums.copy())
            if dist.get_backend() == "nccl":
                t = t.cuda()
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            sums = t.cpu().numpy()
        Ti = solve(sums)
        cur = transform(Ti, cur)
        T = _mat4_mul_f32(np.asarray(Ti, dtype=np.float32), T)
        mse = sums[15] / sums[16]
        it += 1
        if not fixed and abs(mse - prev) < 1e-12:
            break
        prev = mse
    return T, it, mse


def sor_sharded(partial, threshold, n_total: int, rank: int, world: int, mean_k: int = 50, stddev_mult: float = 1.5, dist=None):
    """StatisticalOutlierRemoval with the cloud's POINTS sharded over the ranks ("SOR adds an all-reduce of
    (sum, sum of squares, count)"): each rank takes the mean distances of its contiguous shard, the four statistics are
    combined over the ranks -- (+, +) for the sums, (min, min) for the smallest positive terms that decide whether the sums
    carry PCL's in-order bits -- and every rank derives the same threshold.

    partial(start, count) -> (mean_dist[count] float32, sums[4])   capi.Index.sor_partial
    threshold(sums[4]) -> (thr, exact)                             capi.sor_threshold bound to n_valid / mean_k / stddev_mult
    Returns (start, mean_dist of the shard, inlier mask of the shard, threshold, kept over ALL shards, exact)."""
    import numpy as np
    import torch
    start, count = shard_range(n_total, rank, world)
    md, sums = partial(start, count)
    sums = np.asarray(sums, dtype=np.float64)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        on_gpu = dist.get_backend() == "nccl"
        add, low = torch.from_numpy(sums[:2].copy()), torch.from_numpy(sums[2:].copy())
        if on_gpu:
            add, low = add.cuda(), low.cuda()
        dist.all_reduce(add, op=dist.ReduceOp.SUM)
        dist.all_reduce(low, op=dist.ReduceOp.MIN)
        sums = np.concatenate([add.cpu().numpy(), low.cpu().numpy()])
    thr, exact = threshold(sums)
    inl = ~(np.asarray(md, dtype=np.float32).astype(np.float64) > thr)
    kept = torch.tensor([int(inl.sum())], dtype=torch.int64)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "nccl":
            kept = kept.cuda()
        dist.all_reduce(kept, op=dist.ReduceOp.SUM)
    return start, md, inl, thr, int(kept.item()), exact
