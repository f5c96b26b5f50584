#!/usr/bin/env python3
"""bench.py — headline benchmark of the backward-search hot path on MI355X.

Workload (BASELINE.json configs[1]): count() of 1,048,576 random 8-char patterns (substrings of the
text at SplitMix64(43) positions) on the FM-index (sampleRate 32) of 256 MiB of synthetic log text
(SplitMix64(42)).  One "step" = one pass of the batch through fmx_count_batch_dev with the
patterns already resident in HBM.  With --gpus N (one process per GPU, launched by
torch.distributed.run) rank 0 builds the index, its flat HBM image is broadcast over RCCL, and every
rank counts its own 1,048,576-pattern shard (weak scaling, no collective on the data path).

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline` and `cpu_baseline`.
PyTorch is only plumbing here: device memory, the stream, HIP events and torch.distributed.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_latest.json")  # rocprofv3 --pmc passes of this very command


def pmc_traffic(text_log2, patterns, sample_rate):
    """fabric-side bytes per k_count launch from the committed PMC passes (FETCH_SIZE + WRITE_SIZE, KiB units,
    collected in separate passes by tools/profile.sh); None when no profile matches the workload"""
    try:
        p = json.load(open(PMC_FILE))
        w = p["workload"]
        if (w["text_log2"], w["patterns"], w["sample_rate"]) != (text_log2, patterns, sample_rate):
            return None
        return (p["k_count"]["FETCH_SIZE_KiB"] + p["k_count"]["WRITE_SIZE_KiB"]) * 1024.0
    except (OSError, KeyError, ValueError):
        return None


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_or_load_index(ia, text_log2, sample_rate, cache_dir, build_device=0):
    """index of 2^text_log2 chars of synthetic log; the serialized form is cached under cache_dir.
    Construction runs its suffix-array stage on GPU `build_device` (same index, byte for byte: fmx_build_on_device);
    None = host builder."""
    n = 1 << text_log2
    t0 = time.time()
    text = ia.synth_log(n, seed=42)
    key = hashlib.sha256(text[: 1 << 16].tobytes() + b"%d-%d-v1" % (n, sample_rate)).hexdigest()[:16]
    path = os.path.join(cache_dir, "fmx_%s.ser" % key)
    t1 = time.time()
    if os.path.exists(path):
        fm = ia.FmIndex.read(open(path, "rb").read(), device=None)
        log("[bench] index loaded from %s in %.1fs" % (path, time.time() - t1))
    else:
        fm = ia.FmIndex(text, sample_rate, True, device=None, build_device=build_device)
        log("[bench] text %.1fs, index built in %.1fs (%s, %d host cores)"
            % (t1 - t0, time.time() - t1, "suffix array on GPU %d" % build_device if build_device is not None else "host builder",
               os.cpu_count()))
        try:
            os.makedirs(cache_dir, exist_ok=True)
            tmp = path + ".%d.tmp" % os.getpid()
            with open(tmp, "wb") as f:
                f.write(fm.write(False))
            os.replace(tmp, path)
        except OSError as e:
            log("[bench] could not cache the index: %s" % e)
    return text, fm, path


def cpu_baseline(path, pat, off, budget_s, alg_bytes_holder):
    """the oracle (plain-C port of the reference path) timed on host cores over a bounded sample"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc

    ref = orc.OracleFmIndex.read(open(path, "rb").read())
    n = len(off) - 1
    chunk = 20000
    done = 0
    orc.counters_reset()
    t0 = time.time()
    first = None
    while done < n and time.time() - t0 < budget_s:
        hi = min(n, done + chunk)
        c, s = ref.count_batch(pat[off[done]: off[hi]], off[done: hi + 1] - off[done], threads=1)
        if first is None:
            first = (c.copy(), done, hi)
        done = hi
    dt = time.time() - t0
    cnt = orc.counters()
    alg_bytes_holder["bytes_per_step"] = cnt["alg_bytes"] / max(1, cnt["lf_steps"])
    alg_bytes_holder["levels_per_step"] = cnt["wt_levels"] / max(1, cnt["lf_steps"])
    one = {"value": done / dt, "unit": "patterns/s", "cores": 1, "kind": "port",
           "lf_steps_per_s": cnt["lf_steps"] / dt,
           "sample": "first %d of the %d patterns of the same batch, oracle/index4j_oracle.c (C port of index4j's "
                     "count path; the Java reference cannot run here: no JDK), 1 thread, %.1f s" % (done, n, dt)}
    # all host cores (OpenMP over patterns), reported beside it
    cores = os.cpu_count() or 1
    if cores > 1:
        m = min(n, max(chunk, done * min(cores, 16) // 2))
        t0 = time.time()
        ref.count_batch(pat[: off[m]], off[: m + 1], threads=cores)
        dt = time.time() - t0
        one["all_cores"] = {"value": m / dt, "unit": "patterns/s", "cores": cores, "sample": "first %d patterns" % m}
    return one, first


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--text-log2", type=int, default=28, help="log2 of the text length in chars (28 = 256 MiB)")
    ap.add_argument("--patterns", type=int, default=1 << 20)
    ap.add_argument("--pattern-len", type=int, default=8)
    ap.add_argument("--sample-rate", type=int, default=32)
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of oracle time for cpu_baseline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cache-dir", default=os.environ.get("FMX_CACHE", "/tmp/fmx_cache"))
    args = ap.parse_args()

    import torch

    import index4j_amd as ia

    if not torch.cuda.is_available() or ia.lib.fmx_device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if local_rank >= torch.cuda.device_count():  # launcher that gives every rank its own visible-device mask
        local_rank = 0
    if world != args.gpus:
        log("[bench] WORLD_SIZE=%d but --gpus %d; using WORLD_SIZE" % (world, args.gpus))
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    # ---- index: rank 0 builds (or loads the cached serialized form), the HBM image is broadcast ----
    text = fm = path = None
    if rank == 0:
        text, fm, path = build_or_load_index(ia, args.text_log2, args.sample_rate, args.cache_dir, build_device=local_rank)
    if world > 1:
        size = torch.zeros(1, dtype=torch.int64, device=dev)
        if rank == 0:
            host_blob = fm.blob()
            size[0] = len(host_blob)
        dist.broadcast(size, 0)
        d_blob = torch.empty(int(size.item()), dtype=torch.uint8, device=dev)
        if rank == 0:
            d_blob.copy_(torch.from_numpy(host_blob))
        t0 = time.time()
        dist.broadcast(d_blob, 0)  # RCCL over xGMI: the immutable index, once
        torch.cuda.synchronize()
        if rank == 0:
            log("[bench] index blob %.1f MB broadcast to %d GPUs in %.3fs" % (size.item() / 1e6, world, time.time() - t0))
        q = ia.FmIndex.attach_device_blob(d_blob.data_ptr(), d_blob.numel(), local_rank)
        if rank != 0:
            text = ia.synth_log(1 << args.text_log2, seed=42)
    else:
        fm.to_device(local_rank)
        q = fm

    # ---- this rank's shard of patterns (resident in HBM before the timed region) ----
    n = args.patterns
    pat, off, _pos = ia.synth_patterns(text, args.pattern_len, n, seed=43 + rank)
    d_pat = torch.from_numpy(pat.view(np.int16)).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    d_cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    d_lf = torch.zeros(n, dtype=torch.int32, device=dev)
    d_st = torch.zeros(n, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream()

    def step(with_steps):
        rc = ia.lib.fmx_count_batch_dev(q.handle, d_pat.data_ptr(), d_off.data_ptr(), n, d_cnt.data_ptr(),
                                        d_lf.data_ptr() if with_steps else None, d_st.data_ptr() if with_steps else None,
                                        C.c_void_p(stream.cuda_stream))
        if rc != 0:
            raise RuntimeError("fmx_count_batch_dev failed: %s" % ia.lib.fmx_last_error().decode())

    step(True)  # also yields the exact LF-step count of the batch
    torch.cuda.synchronize()
    lf_steps_per_launch = int(d_lf.sum(dtype=torch.int64).item())
    if int(d_st.max().item()) != 0:
        raise RuntimeError("unexpected per-query status in the benchmark batch")
    checksum = int(d_cnt.sum(dtype=torch.int64).item())
    for _ in range(args.warmup):
        step(False)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step(False)  # the whole hot path: suffix-key sort of the batch + k_count
    ev1.record(stream)
    barrier()
    wall = time.perf_counter() - t0
    step_ms = ev0.elapsed_time(ev1) / args.steps

    # the dominant kernel alone (k_count over the same processing order), HIP events on its stream
    perm = C.c_void_p()
    rc = ia.lib.fmx_count_plan_dev(q.handle, d_pat.data_ptr(), d_off.data_ptr(), n, C.byref(perm), C.c_void_p(stream.cuda_stream))
    if rc != 0:
        raise RuntimeError("fmx_count_plan_dev failed: %s" % ia.lib.fmx_last_error().decode())
    torch.cuda.synchronize()
    ev2, ev3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev2.record(stream)
    for _ in range(args.steps):
        rc = ia.lib.fmx_count_ordered_dev(q.handle, d_pat.data_ptr(), d_off.data_ptr(), perm, n, d_cnt.data_ptr(), None,
                                          None, C.c_void_p(stream.cuda_stream))
        if rc != 0:
            raise RuntimeError("fmx_count_ordered_dev failed: %s" % ia.lib.fmx_last_error().decode())
    ev3.record(stream)
    torch.cuda.synchronize()
    kernel_ms = ev2.elapsed_time(ev3) / args.steps
    if int(d_cnt.sum(dtype=torch.int64).item()) != checksum:
        raise RuntimeError("counts changed between launches")
    if dist is not None:
        tw = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())
        tot = torch.tensor([lf_steps_per_launch], dtype=torch.int64, device=dev)
        dist.all_reduce(tot)
        lf_total = int(tot.item())
    else:
        lf_total = lf_steps_per_launch

    # final gather of the shards' results on rank 0 (outside the timed region: the shards are independent)
    gathered_checksum = None
    if dist is not None:
        from index4j_amd.shard import gather_concat

        all_counts = gather_concat(dist, d_cnt, [n] * world, dev)
        if rank == 0:
            gathered_checksum = int(all_counts.astype(np.int64).sum())
            if len(all_counts) != n * world:
                raise RuntimeError("gather returned %d results for %d patterns" % (len(all_counts), n * world))
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    # ---- CPU baseline + algorithmic bytes per LF-step from the oracle's counting mode ----
    holder = {}
    base = None
    if world > 1 and not args.no_cpu_baseline:
        # cpu_baseline is reported at N=1 only; a 2,000-pattern oracle pass still yields bytes per LF-step
        cpu_baseline(path, pat[: 2000 * args.pattern_len], off[:2001], 1e9, holder)
    elif not args.no_cpu_baseline:
        base, first = cpu_baseline(path, pat, off, args.cpu_budget, holder)
        c, lo, hi = first
        if not (d_cnt[lo:hi].cpu().numpy() == c).all():
            raise RuntimeError("GPU counts differ from the oracle on the baseline sample")
    bytes_per_step = holder.get("bytes_per_step")
    ms_per_step = wall * 1e3 / args.steps
    patterns_per_s = world * n * args.steps / wall
    roof = None
    if bytes_per_step:
        # a measured streaming figure beside the vendor peak: device-to-device copy of 1 GiB, read + write
        src = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
        dst = torch.empty_like(src)
        dst.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        copy_gbs = 5 * 2 * (1 << 30) / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del src, dst
        achieved = bytes_per_step * lf_steps_per_launch / (kernel_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic(args.text_log2, n, args.sample_rate), "kernel": "k_count",
                "kernel_ms": kernel_ms, "step_ms_incl_sort": step_ms, "alg_bytes_per_lf_step": bytes_per_step,
                "lf_steps_per_launch": lf_steps_per_launch,
                "wt_levels_per_lf_step": holder.get("levels_per_step"),
                "measured_copy_GBps": copy_gbs, "frac_of_measured_copy": achieved / copy_gbs}
    out = {
        "metric": "patterns/sec + LF-steps/sec, 1M x 8-char count() on 256 MiB log index",
        "value": patterns_per_s,
        "unit": "patterns/s",
        "lf_steps_per_sec": lf_total * args.steps / wall,
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int32",
        "data": "synthetic",
        "config": {"workload": "count() batch of %d random %d-char patterns per GPU on %d MiB synthetic log text, "
                               "sampleRate=%d (BASELINE.json configs[1])" % (n, args.pattern_len,
                                                                             (1 << args.text_log2) >> 20, args.sample_rate),
                   "text_chars": 1 << args.text_log2, "patterns_per_gpu": n, "pattern_len": args.pattern_len,
                   "sample_rate": args.sample_rate, "parallelism": "patterns sharded x%d, index replicated" % world,
                   "count_checksum": checksum, "gathered_checksum_all_ranks": gathered_checksum},
        "roofline": roof,
        "cpu_baseline": base,
    }
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
