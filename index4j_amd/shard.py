"""Multi-GPU plumbing: pattern batches shard embarrassingly (every query is an independent read of
an immutable index, FM:82 @ThreadSafe), so the only collectives are the one-off broadcast of the
index blob and an optional gather of results.  One process per GPU; backend "nccl" is RCCL on ROCm
(xGMI); the CPU test-suite runs the same code over gloo."""
import numpy as np


def shard_range(n, world, rank):
    """contiguous slice [lo, hi) of n items owned by `rank` (slices differ by at most one item)"""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_blob(dist, blob_host, device, src=0):
    """broadcast the flat index image from `src`; returns a uint8 torch tensor on `device`.
    blob_host: numpy uint8 array on src, ignored elsewhere."""
    import torch

    rank = dist.get_rank()
    size = torch.zeros(1, dtype=torch.int64, device=device)
    if rank == src:
        size[0] = len(blob_host)
    dist.broadcast(size, src)
    buf = torch.empty(int(size.item()), dtype=torch.uint8, device=device)
    if rank == src:
        buf.copy_(torch.from_numpy(np.ascontiguousarray(blob_host)))
    dist.broadcast(buf, src)
    return buf


def gather_concat(dist, local, counts_per_rank, device, dst=0):
    """gather variable-length int32 shards onto `dst` in rank order (the final 'gather' of the north star)"""
    import torch

    world, rank = dist.get_world_size(), dist.get_rank()
    mx = max(counts_per_rank)
    pad = torch.zeros(mx, dtype=torch.int32, device=device)
    src = local if isinstance(local, torch.Tensor) else torch.as_tensor(np.asarray(local), dtype=torch.int32)
    pad[: len(local)] = src.to(device=device, dtype=torch.int32)
    out = [torch.zeros(mx, dtype=torch.int32, device=device) for _ in range(world)]
    dist.all_gather(out, pad)
    if rank != dst:
        return None
    return np.concatenate([out[r][: counts_per_rank[r]].cpu().numpy() for r in range(world)])
