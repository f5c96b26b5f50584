"""N>1 path on CPU: world_size-2 gloo run of the sharding plumbing bench.py uses (broadcast of the index
blob, contiguous pattern shards, rank-ordered gather).  Queries run through the test-only host
simulation here; on the GPU box the same plumbing runs over RCCL with the HIP kernels."""
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    import hostsim
    import index4j_amd as ia
    from index4j_amd.shard import broadcast_blob, gather_concat, shard_range

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    text = ia.synth_log(1 << 16)
    pat, off, _ = ia.synth_patterns(text, 8, 1001)
    blob = None
    if rank == 0:
        fm = ia.FmIndex(text, 32, True, device=None)
        blob = fm.blob()
    buf = broadcast_blob(dist, blob, torch.device("cpu"))

    class Holder:  # HostSim over the received image
        def blob(self_inner):
            return buf.numpy()

    h = hostsim.HostSim(Holder())
    lo, hi = shard_range(1001, world, rank)
    cnt, st, lf, _ = h.count_batch(pat[off[lo]:off[hi]], off[lo:hi + 1] - off[lo])
    sizes = [shard_range(1001, world, r)[1] - shard_range(1001, world, r)[0] for r in range(world)]
    allc = gather_concat(dist, cnt, sizes, torch.device("cpu"))
    if rank == 0:
        full, _, _, _ = hostsim.HostSim(fm).count_batch(pat, off)
        q.put(bool((allc == full).all()) and len(allc) == 1001)
    dist.destroy_process_group()


def test_two_rank_shard_broadcast_gather():
    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
    assert ok


def test_shard_range_partitions():
    from index4j_amd.shard import shard_range

    for n in (0, 1, 7, 1000, 1 << 20):
        for w in (1, 2, 3, 8):
            parts = [shard_range(n, w, r) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in parts) - min(b - a for a, b in parts) <= 1
