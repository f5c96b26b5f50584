"""N>1 path on CPU: world_size-2 gloo run of the sharding plumbing bench.py uses (broadcast of the index
blob, contiguous pattern shards, rank-ordered gather).  Queries run through the test-only host
simulation here; on the GPU box the same plumbing runs over RCCL with the HIP kernels."""
import json
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    import hostsim
    import index4j_amd as ia
    from index4j_amd.shard import broadcast_blob, gather_concat, scatter_rows, shard_range

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
    # the slice fan-out (scatter + all-gather of slices, the default above two ranks) delivers the same bytes
    buf2 = broadcast_blob(dist, blob, torch.device("cpu"), fan_out=True)
    assert buf2.shape == buf.shape and bool((buf2 == buf).all())

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


@pytest.mark.parametrize("world", [2, 8])
def test_shard_broadcast_gather(world):
    """two ranks, and the eight of one node: image broadcast (plain and slice fan-out), contiguous shards, gather"""
    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
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


def _run_bench(extra, env_extra=None, timeout=600):
    import json
    import subprocess

    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    env["FMX_CACHE"] = os.path.join("/tmp", "fmx_cache_test_%d" % os.getpid())
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--text-log2", "16", "--patterns", "3001",
                        "--patterns-total", "5003", "--segment-log2", "14", "--segments", "3", "--batches", "2", "--steps", "2",
                        "--warmup", "1"] + extra, capture_output=True, text=True, env=env, timeout=timeout)
    detail = [ln[len("BENCH_DETAIL "):] for ln in r.stdout.splitlines() if ln.startswith("BENCH_DETAIL ")]
    if r.returncode == 0:
        _check_contract_line(r.stdout, json.loads(detail[-1]))
    return r.returncode, (json.loads(detail[-1]) if detail else None), r.stderr


def _check_contract_line(stdout, detail):
    """what the driver's parser sees: the LAST line of the last 8,000 characters of stdout is the whole contract line,
    below 4 KB, and agrees with the detail record (round 3's one 21.7 KB line went unparsed)"""
    tail = stdout[-8000:]
    last = tail.rstrip("\n").splitlines()[-1]
    assert stdout.rstrip("\n").splitlines()[-1] == last, "the contract line does not fit the driver's 8,000-character tail"
    assert len(last) < 4096, len(last)
    c = json.loads(last)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "ranks_seen"):
        assert k in c, k
        if k not in ("config", "roofline", "cpu_baseline"):
            assert c[k] == detail[k], k
    assert isinstance(c["config"]["workload"], str) and "model" not in c["config"]
    # process start -> timed loop (index build + image broadcasts): on the line, so that a first real multi-GPU run that is slow to
    # set up shows where the time went
    assert isinstance(c.get("setup_s"), float) and c["setup_s"] >= 0.0


def test_bench_launches_itself_and_reports_the_ranks_that_really_ran():
    """bench.py's own N>1 code path, rehearsed on CPU over gloo (--dry-run: launch, index build on rank 0, image
    broadcast, one batch handed out in shards, gather, JSON line — no queries): `--gpus 2` without a launcher starts
    torch.distributed.run as a child, and the line carries the ranks that joined"""
    rc, out, err = _run_bench(["--gpus", "2"])
    assert rc == 0, err[-2000:]
    assert out["dry_run"] and out["n_gpus"] == 2 and sorted(r[0] for r in out["ranks_seen"]) == [0, 1]
    assert out["value"] is None and out["scaling"] == "weak"
    rc, out, err = _run_bench(["--gpus", "2", "--workload", "segments"])
    assert rc == 0, err[-2000:]
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["patterns_total"] == 5003
    assert sorted(r[0] for r in out["ranks_seen"]) == [0, 1]


def test_bench_eight_rank_rehearsal_of_both_workloads_with_the_slice_fan_out():
    """the driver's N = 8 launch, rehearsed over gloo: `bench.py --gpus 8` carries the weak-scaling headline AND the
    configs[4] line (segment images broadcast, ONE batch sharded over 8 ranks) in `secondary`; the images travel by the
    slice fan-out (scatter of slices from rank 0 + all-gather among the ranks)"""
    rc, out, err = _run_bench(["--gpus", "8"], {"FMX_FAN_OUT_BROADCAST": "1"}, timeout=900)
    assert rc == 0, err[-2000:]
    assert out["dry_run"] and out["n_gpus"] == 8 and sorted(r[0] for r in out["ranks_seen"]) == list(range(8))
    seg = out["secondary"][0]
    assert seg["n_gpus"] == 8 and seg["scaling"] == "strong" and seg["config"]["patterns_total"] == 5003
    assert sorted(r[0] for r in seg["ranks_seen"]) == list(range(8))
    rc, out, err = _run_bench(["--gpus", "8", "--workload", "segments"], {"FMX_FAN_OUT_BROADCAST": "1"}, timeout=900)
    assert rc == 0, err[-2000:]
    assert out["n_gpus"] == 8 and out["scaling"] == "strong" and sorted(r[0] for r in out["ranks_seen"]) == list(range(8))


def test_bench_refuses_a_rank_count_that_differs_from_gpus():
    rc, out, err = _run_bench(["--gpus", "2"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0",
                                                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert rc == 2 and out is None and "mislabelled" in err
    rc, out, err = _run_bench(["--gpus", "1"])
    assert rc == 0 and out["n_gpus"] == 1 and out["ranks_seen"] == [[0, 0, 0]]
