"""Multi-GPU plumbing: pattern batches shard embarrassingly (every query is an independent read of
an immutable index, FM:82 @ThreadSafe), so the only collectives are the one-off broadcast of the
index blob, the hand-out of the pattern shards and an optional gather of results.  One process per GPU;
backend "nccl" is RCCL on ROCm (xGMI); the CPU test-suite runs the same code over gloo."""
import numpy as np


def shard_range(n, world, rank):
    """contiguous slice [lo, hi) of n items owned by `rank` (slices differ by at most one item)"""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_blob(dist, blob_host, device, src=0, fan_out=None):
    """the flat index image from `src` to every rank; returns a uint8 torch tensor on `device`.
    blob_host: numpy uint8 array on src, ignored elsewhere.
    fan_out (default off; env FMX_FAN_OUT_BROADCAST=1 turns it on above two ranks): instead of one broadcast — a single
    ring / tree out of `src`, bound by ONE xGMI link — `src` scatters the image in `world` slices (its egress goes over
    all links to its peers at once), then every rank all-gathers the slices from its peers.  Same bytes
    either way; a one-off start-up cost in both forms, so the plain broadcast stays the default until the slice form has
    been timed on an 8-GPU node."""
    import torch

    world, rank = dist.get_world_size(), dist.get_rank()
    size = torch.zeros(1, dtype=torch.int64, device=device)
    if rank == src:
        size[0] = len(blob_host)
    dist.broadcast(size, src)
    n = int(size.item())
    if fan_out is None:
        import os

        fan_out = world > 2 and os.environ.get("FMX_FAN_OUT_BROADCAST", "0") == "1"
    if not fan_out or world == 1:
        buf = torch.empty(n, dtype=torch.uint8, device=device)
        if rank == src:
            buf.copy_(torch.from_numpy(np.ascontiguousarray(blob_host)))
        dist.broadcast(buf, src)
        return buf
    per = (n + world - 1) // world
    per = (per + 255) // 256 * 256  # slices start on 256-byte boundaries
    full = torch.empty(per * world, dtype=torch.uint8, device=device)
    parts = None
    if rank == src:
        full[:n].copy_(torch.from_numpy(np.ascontiguousarray(blob_host)))
        full[n:].zero_()
        parts = [full[r * per:(r + 1) * per] for r in range(world)]
    mine = torch.empty(per, dtype=torch.uint8, device=device)
    dist.scatter(mine, parts, src=src)
    dist.all_gather([full[r * per:(r + 1) * per] for r in range(world)], mine)
    return full[:n]


def scatter_rows(dist, rows_host, row_len, total_rows, device, dtype, src=0):
    """hand out a (total_rows, row_len) array that lives on `src` in contiguous shards (shard_range): every rank
    gets its rows as a torch tensor on `device`.  One collective (scatter of equal, padded shards)."""
    import torch

    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_range(total_rows, world, r)[1] - shard_range(total_rows, world, r)[0] for r in range(world)]
    mx = max(sizes) if sizes else 0
    out = torch.zeros((mx, row_len), dtype=dtype, device=device)
    parts = None
    if rank == src:
        full = torch.from_numpy(np.ascontiguousarray(rows_host).reshape(total_rows, row_len))
        parts = []
        for r in range(world):
            lo, hi = shard_range(total_rows, world, r)
            t = torch.zeros((mx, row_len), dtype=dtype, device=device)
            t[: hi - lo] = full[lo:hi].to(device=device, dtype=dtype)
            parts.append(t)
    dist.scatter(out, parts, src=src)
    return out[: sizes[rank]].contiguous()


def gather_concat(dist, local, counts_per_rank, device, dst=0, dtype=None):
    """gather variable-length shards onto `dst` in rank order (the final 'gather' of the north star)"""
    import torch

    world, rank = dist.get_world_size(), dist.get_rank()
    mx = max(counts_per_rank)
    src = local if isinstance(local, torch.Tensor) else torch.as_tensor(np.asarray(local))
    if dtype is None:
        dtype = src.dtype if src.dtype in (torch.int32, torch.int64) else torch.int32
    pad = torch.zeros(mx, dtype=dtype, device=device)
    pad[: len(local)] = src.to(device=device, dtype=dtype)
    out = [torch.zeros(mx, dtype=dtype, device=device) for _ in range(world)]
    dist.all_gather(out, pad)
    if rank != dst:
        return None
    return np.concatenate([out[r][: counts_per_rank[r]].cpu().numpy() for r in range(world)])


def ranks_seen(dist, device, local_rank, device_index):
    """[(rank, local_rank, device index)] of every process that joined, gathered over the backend itself —
    a line printed by rank 0 then proves how many GPUs really took part"""
    import torch

    me = torch.tensor([dist.get_rank(), local_rank, device_index], dtype=torch.int64, device=device)
    out = [torch.zeros(3, dtype=torch.int64, device=device) for _ in range(dist.get_world_size())]
    dist.all_gather(out, me)
    return [[int(v) for v in t.cpu().tolist()] for t in out]
