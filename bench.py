#!/usr/bin/env python3
"""bench.py — headline benchmark of the backward-search hot path on MI355X.

Workload `count` (default; BASELINE.json configs[1]): count() of 1,048,576 random 8-char patterns per GPU
(substrings of the text at SplitMix64 positions) on the FM-index (sampleRate 32) of 256 MiB of synthetic log
text (SplitMix64(42)).  One "step" = one pass of a batch through fmx_count_batch_dev (plan stage + k_count) with
the patterns already resident in HBM; the timed loop rotates through `--batches` distinct batches (seeds 43..).
With --gpus N (one process per GPU over RCCL) rank 0 builds the index, its flat HBM image is broadcast, every
batch is ONE batch of N x 1,048,576 patterns made on rank 0 and handed out in contiguous shards, every rank
counts its shard (weak scaling, no collective on the data path) and the results are gathered on rank 0.

Workload `segments` (BASELINE.json configs[4]): count() + locate(maxMatches 16) of ONE batch of 8,388,608 patterns
over a 2 GiB text held as 8 segment indexes (a Java int cannot address 2^31 chars), the images broadcast, the batch
sharded over the ranks (strong scaling).

`python bench.py --gpus N` without a launcher starts `torch.distributed.run` itself as a child process; a run
whose rank count differs from --gpus fails instead of printing a mislabelled line.

Output (rank 0): the DETAIL record — everything measured, with prose — as the stdout line `BENCH_DETAIL {...}` and as
gpurun_out/bench_detail.json; then, LAST, one compact JSON line (< 4 KB for every workload and N) with the driver's
contract fields, `roofline`, `cpu_baseline`, `ranks_seen` and one {config, ms, frac} triple per secondary config
(configs[2], [3]; the reference-shaped series only with --series).  tools/bench_detail.py extracts the detail from a
captured stdout.  All times are MEANS over the timed calls.  PyTorch is only plumbing here: device memory, the stream,
HIP events and torch.distributed.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import shutil
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_latest.json")  # rocprofv3 --pmc passes of this very command
KERNEL_SOURCES = ["fmx_kernels.hip", "fmx_device.hpp", "fmx_blob.hpp", "fmx_blob.cpp", "fmx_plan.hpp"]


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def env_options(ia):
    """options FMX_OPTIONS set at import (index4j_amd/_lib.py): what a block that changes one for a while puts back"""
    import index4j_amd._lib as il

    return getattr(il, "ENV_OPTIONS", {})


def kernel_source_sha():
    """digest of the sources that decide what k_count reads and how (stamped into profiles/pmc_latest.json)"""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "index4j_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def pmc_counters(text_log2, patterns, sample_rate):
    """per-launch PMC averages of k_count from the committed passes (tools/profile.sh) — only if they were taken on
    THIS workload and THESE kernel sources; otherwise None (a stale profile must not dress up a new kernel)"""
    try:
        p = json.load(open(PMC_FILE))
        w = p["workload"]
        if (w["text_log2"], w["patterns"], w["sample_rate"]) != (text_log2, patterns, sample_rate):
            return None, "profile is of another workload"
        if p.get("kernel_source_sha") != kernel_source_sha():
            return None, "profiles/pmc_latest.json was taken on other kernel sources (%s, now %s): re-run tools/profile.sh" % (
                p.get("kernel_source_sha"), kernel_source_sha())
        k = dict(p["k_count"])
        k["calibration"] = p.get("calibration")
        return k, None
    except (OSError, KeyError, ValueError) as e:
        return None, "no usable profile (%s)" % e


def pmc_row_lookup():
    """-> lookup(row key, queries) = bytes the memory system moved per call of that row (FETCH_SIZE x calibration + WRITE_SIZE
    summed over the row's kernels: tools/profile_rows.sh -> profiles/pmc_latest.json `rows`), or None when the counters on file
    were taken on other kernel sources or another batch size"""
    try:
        p = json.load(open(PMC_FILE))
    except (OSError, ValueError):
        return lambda key, queries: None
    rows = p.get("rows") or {}
    fresh = p.get("rows_kernel_source_sha") == kernel_source_sha()
    # (tools/calibrate_fetch.py: a scattered 16-byte load that misses fills a 64-byte sector, which is what FETCH_SIZE tallies: x 1)
    factor = ((p.get("calibration") or {}).get("fabric_bytes_per_FETCH_SIZE_byte")) or 1.0

    def lookup(key, queries):
        r = rows.get(key)
        if not fresh or not r or (queries is not None and r.get("queries") not in (None, queries)):
            return None
        return (r["FETCH_SIZE_KiB"] * factor + r["WRITE_SIZE_KiB"]) * 1024.0

    return lookup


def ref_series_module():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import ref_series

    return ref_series


def index_cache_path(text, text_log2, sample_rate, cache_dir):
    key = hashlib.sha256(text[: 1 << 16].tobytes() + b"%d-%d-v1" % (1 << text_log2, sample_rate)).hexdigest()[:16]
    return os.path.join(cache_dir, "fmx_%s.ser" % key)


def build_or_load_index(ia, text_log2, sample_rate, cache_dir, build_device=0, seed=42):
    """index of 2^text_log2 chars of synthetic log; the serialized form is cached under cache_dir.
    Construction runs its suffix-array stage on GPU `build_device` (same index, byte for byte: fmx_build_on_device);
    None = host builder."""
    n = 1 << text_log2
    t0 = time.time()
    text = ia.synth_log(n, seed=seed)
    path = index_cache_path(text, text_log2, sample_rate, cache_dir)
    t1 = time.time()
    if os.path.exists(path):
        fm = ia.FmIndex.read(open(path, "rb").read(), device=None)
        log("[bench] index loaded from %s in %.1fs" % (path, time.time() - t1))
    else:
        fm = ia.FmIndex(text, sample_rate, True, device=None, build_device=build_device)
        log("[bench] text %.1fs, index (sampleRate %d) built in %.1fs (%s, %d host cores)"
            % (t1 - t0, sample_rate, time.time() - t1,
               "suffix array on GPU %d" % build_device if build_device is not None else "host builder", os.cpu_count()))
        try:
            os.makedirs(cache_dir, exist_ok=True)
            tmp = path + ".%d.tmp" % os.getpid()
            with open(tmp, "wb") as f:
                f.write(fm.write(False))
            os.replace(tmp, path)
        except OSError as e:
            log("[bench] could not cache the index: %s" % e)
    return text, fm, path


# ---------------------------------------------------------------------------------------------------------------
# CPU baselines (rank 0, N = 1 only): the oracle = plain-C port of the reference path, and — if a JVM and an index4j
# jar exist on the box — index4j itself
# ---------------------------------------------------------------------------------------------------------------
def oracle_module():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc

    return orc


def jvm_leg(ser_framed_path, pat, m, cache_dir):
    """index4j's own count() on the host cores (countBenchmark, FmIndexThroughputBenchmark.java:191-199), if `java`
    and INDEX4J_JAR exist; else a record of what the probe found"""
    java = shutil.which("java")
    jar = os.environ.get("INDEX4J_JAR")
    probe = {"java": java, "INDEX4J_JAR": jar}
    if not java or not jar or not os.path.exists(jar):
        return {"available": False, "probe": probe,
                "note": "index4j JVM: not available (java: %s, INDEX4J_JAR: %s); the reference publishes ~1e6 LF-steps/s/core "
                        "(BASELINE.md)" % (java, jar if jar else "unset")}
    try:
        pfile = os.path.join(cache_dir, "patterns_%d.bin" % os.getpid())
        n = len(pat) // m
        with open(pfile, "wb") as f:
            f.write(np.array([n, m], np.int32).tobytes())
            f.write(np.ascontiguousarray(pat, np.uint16).tobytes())
        src = os.path.join(ROOT, "tools", "jvm", "CountBench.java")
        out = {"available": True, "probe": probe, "runs": []}
        for threads in (1, os.cpu_count() or 1):
            r = subprocess.run([java, "-Xmx16g", "-cp", jar, src, ser_framed_path, pfile, str(threads)], capture_output=True,
                               text=True, timeout=900)
            if r.returncode != 0:
                out["runs"].append({"threads": threads, "error": r.stderr[-400:]})
                continue
            out["runs"].append(json.loads(r.stdout.strip().splitlines()[-1]))
        os.unlink(pfile)
        return out
    except Exception as e:  # noqa: BLE001 - a broken JVM leg must not take the benchmark down
        return {"available": False, "probe": probe, "note": "JVM leg failed: %r" % e}


def cpu_baseline(ref, pat, off, budget_s, orc):
    """the oracle timed on host cores over a bounded sample: 1 thread for `budget_s` seconds, then all cores"""
    n = len(off) - 1
    chunk = 20000
    done = 0
    orc.counters_reset()
    t0 = time.time()
    while done < n and time.time() - t0 < budget_s:
        hi = min(n, done + chunk)
        ref.count_batch(pat[off[done]: off[hi]], off[done: hi + 1] - off[done], threads=1)
        done = hi
    dt = time.time() - t0
    cnt = orc.counters()
    one = {"value": done / dt, "unit": "patterns/s", "cores": 1, "kind": "port",
           "lf_steps_per_s": cnt["lf_steps"] / dt,
           "sample": "first %d of the %d patterns of batch 0, oracle/index4j_oracle.c (C port of index4j's count path), "
                     "1 thread, %.1f s" % (done, n, dt)}
    cores = os.cpu_count() or 1
    if cores > 1:
        t0 = time.time()
        ref.count_batch(pat, off, threads=cores)
        dt = time.time() - t0
        one["all_cores"] = {"value": n / dt, "unit": "patterns/s", "cores": cores, "sample": "all %d patterns" % n}
    return one


# ---------------------------------------------------------------------------------------------------------------
def self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher: start torch.distributed.run as a CHILD process (never exec
    from a process that may touch the GPU) and exit with its code"""
    import torch

    if not args.dry_run and not args.share_one_gpu and torch.cuda.device_count() < args.gpus:
        log("[bench] --gpus %d but only %d HIP device(s) visible: refusing to print a mislabelled line"
            % (args.gpus, torch.cuda.device_count()))
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    log("[bench] no launcher in the environment: starting %s" % " ".join(cmd[1:9]))
    return subprocess.call(cmd, env=env)


class Ctx:
    """what every workload needs: rank info, device, torch, dist (or None), the package"""


# ---------------------------------------------------------------------------------------------------------------
# what goes to stdout: the DETAIL (everything measured, with prose) as an earlier line and as a file, and — LAST — one
# compact line of the contract's fields.  The driver keeps the last 8,000 characters of stdout: the contract line must
# fit there whole (round 3's 21.7 KB line did not, and went unparsed).
# ---------------------------------------------------------------------------------------------------------------
COMPACT_LIMIT = 4096
DETAIL_FILE = os.path.join(ROOT, "gpurun_out", "bench_detail.json")


def _pick(d, keys):
    return None if not d else {k: d[k] for k in keys if k in d and d[k] is not None}


def _short(s, n=140):
    s = str(s)
    return s if len(s) <= n else s[: n - 3] + "..."


def _sig(x, digits=5):
    """floats of the optional blocks to `digits` significant digits (the contract's own fields stay exact)"""
    if isinstance(x, float):
        return float("%.*g" % (digits, x))
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


def compact_line(out):
    """the contract's fields of a detail record, bounded below COMPACT_LIMIT characters for every workload and N"""
    c = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                 "scaling", "vs_baseline", "dtype", "data")}
    for k in ("lf_steps_per_sec", "lf_steps_per_sec_reference_equivalent", "lf_steps_per_sec_count_stage", "ms_per_step_ranks",
              "dry_run", "parity", "setup_s", "segments_block"):
        if out.get(k) is not None:
            c[k] = out[k]
    if out.get("rehearsal"):
        c["rehearsal"] = _short(out["rehearsal"], 60)
    c["ranks_seen"] = out.get("ranks_seen")
    cfg = out.get("config") or {}
    c["config"] = {"workload": _short(cfg.get("workload"), 260)}
    for k in ("image", "text_chars", "patterns_per_gpu", "pattern_len", "sample_rate", "batches", "segments", "patterns_total", "max_matches",
              "count_checksum", "patterns_checked_vs_oracle", "distinct_patterns_batch0"):
        if cfg.get(k) is not None:
            c["config"][k] = cfg[k]
    c["config"]["parallelism"] = "dp%d" % (out.get("n_gpus") or 1)
    r = out.get("roofline")
    c["roofline"] = _pick(r, ("bound", "achieved", "peak", "unit", "frac")) if r else None
    if r:
        c["roofline"]["traffic"] = r.get("traffic")
        c["roofline"].update(_pick(r, ("kernel_ms", "alg_bytes_executed_per_launch", "alg_bytes_per_lf_step",
                                       "lf_steps_executed_per_launch", "frac_whole_step", "traffic_frac", "frac_algorithmic", "step_ms_incl_plan",
                                       "fabric_line_fills_per_lf_step_executed", "resident_bytes_per_text_byte",
                                       "resident_bytes_count_path_per_text_byte", "stage_ms_this_rank")) or {})
        c["roofline"]["kernel"] = _short(r.get("kernel"), 60)
        if r.get("plan_stage") is not None:
            c["roofline"]["plan_stage"] = r["plan_stage"]
    b = out.get("cpu_baseline")
    c["cpu_baseline"] = _pick(b, ("value", "unit", "cores", "kind", "lf_steps_per_s")) if b else None
    if b:
        c["cpu_baseline"]["sample"] = _short(b.get("sample"), 150)
        if b.get("all_cores"):
            c["cpu_baseline"]["all_cores"] = _pick(b["all_cores"], ("value", "cores"))
        jv = b.get("index4j_jvm")
        if jv is not None:
            c["cpu_baseline"]["index4j_jvm"] = "timed: see detail" if jv.get("available") else "not available (no JVM on this box)"
    if out.get("overlapped"):
        c["overlapped_ms_per_step"] = out["overlapped"].get("ms_per_step")
    if out.get("suffix_table"):
        st = out["suffix_table"]
        c["suffix_table"] = {"chars": st.get("chars"), "bytes": st.get("bytes"),
                             "ms_per_step_without": (st.get("without_it") or {}).get("ms_per_step")}
        dp = st.get("one_level_deeper") or {}
        if dp.get("ms_per_step"):  # [characters, bytes, ms per step] under suffix_table_image_fraction = 2
            c["suffix_table"]["deeper"] = [dp["chars"], dp["bytes"], dp["ms_per_step"]]
    hb = out.get("host_buffers")
    if hb:
        c["host_buffers"] = _pick(hb, ("ms_per_call", "ms_per_call_registered_buffers", "ratio_to_max_of_floor_and_device_step",
                                       "ratio_registered_to_max_of_floor_and_device_step", "pcie_floor_ms_in", "stat"))
        if hb.get("ms_per_call_all") and hb.get("ms_per_call_registered_buffers_all"):  # (the detail record keeps the long form)
            c["host_buffers"]["stat"] = "means of 7 calls; medians %.3f / %.3f, minima %.3f / %.3f ms (pageable / registered)" % (
                hb.get("ms_per_call_median", 0), hb.get("ms_per_call_registered_buffers_median", 0), min(hb["ms_per_call_all"]),
                min(hb["ms_per_call_registered_buffers_all"]))
        sc = hb.get("scalar_call_us") or {}
        if sc.get("count"):  # microseconds per call of ONE query: count, locate (16 slots), extract (64 characters)
            c["host_buffers"]["scalar_us"] = _sig([sc["count"], sc.get("locate16"), sc.get("extract64")], 3)
    if out.get("index_broadcast"):
        c["index_broadcast"] = _pick(out["index_broadcast"], ("broadcast_s", "fan_out_s", "kept", "bytes"))
    if out.get("plan_net"):
        c["plan_net_ms"] = _sig([out["plan_net"]["planned_ms_per_step"], out["plan_net"]["callers_order_ms_per_step"]], 4)
    if out.get("launch"):
        c["launch"] = _short(out["launch"], 90)
    if out.get("single_process"):
        c["single_process"] = _sig(_pick(out["single_process"], ("ms_per_step", "patterns_per_s", "replicate_s", "devices", "skipped", "error")))
        if c["single_process"].get("error"):
            c["single_process"]["error"] = _short(c["single_process"]["error"], 120)
    sec = []
    for row in out.get("secondary") or []:
        if row is None:
            continue
        if "series" in row:
            for sr in (row["series"] or {}).get("rows", []):
                name = "series %s s=%s%s" % (sr.get("benchmark"), sr.get("sample_rate", sr.get("sample_size")),
                                             "" if sr.get("max_matches") is None else " max=%s" % sr["max_matches"])
                if sr.get("density") is not None:
                    name += " d=%g" % sr["density"]
                rf = sr.get("roofline") or {}
                sec.append({"config": _short(sr.get("key") or name, 44), "ms": sr.get("ms_per_batch", None if sr.get("build_s") is None else sr["build_s"] * 1e3),
                            "frac": rf.get("frac"), "alg": rf.get("frac_algorithmic"), "tfrac": rf.get("traffic_frac")})
            continue
        if row.get("forms"):  # the footprint block: [form, resident bytes per text byte (s = 32 / 64), configs[2] ms, configs[3] ms]
            c["footprint"] = _sig([[_short(f["form"], 28), f.get("resident_bytes_per_text_byte"), f.get("resident_bytes_per_text_byte_s64"),
                                    f.get("configs2_ms"), f.get("configs3_ms")] for f in row["forms"]], 3)
            c["footprint_keys"] = "form, resident B per text B (s=32, s=64), configs[2] ms, configs[3] ms; index4j serialized: %.2f" % (
                row.get("index4j_serialized_bytes_per_text_byte") or 0.0)
            continue
        cfg_name = row.get("config") or row.get("metric")
        if isinstance(cfg_name, dict):  # a whole line of another workload (configs[4] inside the N > 1 default run)
            cfg_name = "configs[4] segments: " + str(cfg_name.get("workload"))
        rf = row.get("roofline") or {}
        sec.append({"config": _short(str(cfg_name).replace("BASELINE.json ", ""), 48),
                    "ms": row.get("ms", row.get("ms_per_step")), "frac": rf.get("frac"), "alg": rf.get("frac_algorithmic"),
                    "tfrac": rf.get("traffic_frac")})
        if row.get("error"):
            sec[-1]["error"] = _short(row["error"], 120)
    if sec:
        c["secondary"] = sec
        c["secondary_keys"] = "frac = the row's roofline fraction (rule: ref_series.settle_frac), alg = by algorithmic bytes, tfrac = by counter traffic"
    exact = ("bound", "achieved", "peak", "unit", "frac", "traffic", "value", "cores", "kind")
    for blk in ("roofline", "cpu_baseline"):
        if c.get(blk):
            c[blk] = {k: (v if k in exact else _sig(v)) for k, v in c[blk].items()}
    for blk in ("secondary", "host_buffers", "index_broadcast", "suffix_table", "overlapped_ms_per_step", "ms_per_step_ranks"):
        if blk in c:
            c[blk] = _sig(c[blk])
    c["detail"] = "gpurun_out/bench_detail.json (+ the stdout line before this one)"
    line = json.dumps(c, separators=(",", ":"))
    if len(line) >= COMPACT_LIMIT:  # never let an over-long line out again: shed the optional blocks, largest first
        for k in ("secondary", "footprint", "host_buffers", "index_broadcast", "suffix_table", "ranks_seen"):
            if k in c and len(line) >= COMPACT_LIMIT:
                c[k] = "see detail"
                line = json.dumps(c, separators=(",", ":"))
    if len(line) >= COMPACT_LIMIT:
        raise RuntimeError("bench.py's contract line is %d characters (limit %d)" % (len(line), COMPACT_LIMIT))
    return line


def emit(out):
    """detail first (stdout line + file), the compact contract line LAST"""
    detail = json.dumps(out)
    try:
        os.makedirs(os.path.dirname(DETAIL_FILE), exist_ok=True)
        with open(DETAIL_FILE, "w") as f:
            f.write(detail + "\n")
    except OSError as e:
        log("[bench] could not write %s: %s" % (DETAIL_FILE, e))
    print("BENCH_DETAIL " + detail, flush=True)
    print(compact_line(out), flush=True)


def hip_events(torch):
    return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def check_rc(ia, rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %s" % (what, ia.lib.fmx_last_error().decode()))


def barrier(ctx):
    if ctx.dist is not None:
        ctx.dist.barrier()
    if not ctx.dry:
        ctx.torch.cuda.synchronize()


def attach_everywhere(ctx, fm):
    """rank 0's index -> every rank: broadcast of the flat image, attached in place (validated by the library)"""
    from index4j_amd.shard import broadcast_blob

    ia, torch = ctx.ia, ctx.torch
    if ctx.dist is None:
        if ctx.dry:
            return None, None, len(fm.blob())
        fm.to_device(ctx.local_rank)
        return fm, None, fm.device_blob()[1]
    host = fm.blob() if ctx.rank == 0 else None
    times = {}
    buf = None
    # RCCL over xGMI: the immutable index, once.  Above two ranks both forms are timed (same bytes; the second is kept):
    # one broadcast out of rank 0, and the slice fan-out (scatter of slices + all-gather among the ranks)
    for form in ([False, True] if ctx.world > 2 else [False]):
        barrier(ctx)
        t0 = time.time()
        got = broadcast_blob(ctx.dist, host, ctx.cdev, fan_out=form)
        barrier(ctx)
        times["fan_out_s" if form else "broadcast_s"] = time.time() - t0
        if buf is not None and not bool((buf == got).all()):
            raise RuntimeError("the slice fan-out delivered other bytes than the broadcast")
        buf = got
    # the faster form serves every later image of this run (the segment images of configs[4]); all ranks must agree: rank 0 decides
    # (deterministic inside the noise: the slice fan-out is only kept where it beat the plain broadcast by 10 % or more)
    keep_fan_out = "fan_out_s" in times and times["fan_out_s"] < 0.9 * times["broadcast_s"]
    pick = torch.tensor([1 if keep_fan_out else 0], dtype=torch.int32, device=ctx.cdev)
    ctx.dist.broadcast(pick, 0)
    ctx.fan_out = bool(int(pick.item()))
    ctx.broadcast_times = dict(times, bytes=int(buf.numel()), ranks=ctx.world, kept="fan_out_s" if ctx.fan_out else "broadcast_s",
                               note="first use of a collective includes its set-up; host staging included")
    if ctx.rank == 0:
        log("[bench] index image %.1f MB to %d rank(s): %s" % (buf.numel() / 1e6, ctx.world, times))
    if ctx.dry:
        return None, buf, buf.numel()
    buf = buf.to(ctx.dev)
    q = ia.FmIndex.attach_device_blob(buf.data_ptr(), buf.numel(), ctx.local_rank)
    return q, buf, buf.numel()


def hand_out_patterns(ctx, pat_host, m, total):
    """the batch's chars (uint16, made on rank 0) -> this rank's contiguous shard in device memory, as a flat int16
    tensor.  Travels as bytes: neither RCCL nor gloo moves 16-bit integers."""
    from index4j_amd.shard import scatter_rows, shard_range

    torch = ctx.torch
    rows = None if pat_host is None else np.ascontiguousarray(pat_host, dtype=np.uint16).view(np.uint8)
    if ctx.dist is None:
        t = torch.from_numpy(rows.reshape(total, 2 * m)).to(ctx.dev)
    else:
        t = scatter_rows(ctx.dist, rows, 2 * m, total, ctx.cdev, torch.uint8).to(ctx.dev)
        lo, hi = shard_range(total, ctx.world, ctx.rank)
        assert t.shape[0] == hi - lo
    return t.contiguous().view(torch.int16).reshape(-1)


def single_process_leg(ctx, args, q, text, n, m, n_batches):
    """At N > 1 in the launcher form: rank 0 ALSO measures the single-process form (fmx_replicate + fmx_count_batch_multi_dev from one
    host process) over the same GPUs, so that one driver run yields both forms side by side.  The other ranks wait on the
    rendezvous store — a host-side wait: no collective's kernel spins on their GPUs meanwhile — with their own work drained."""
    import datetime

    dist, torch, ia, world = ctx.dist, ctx.torch, ctx.ia, ctx.world
    key = "fmx_single_process_leg_done"
    result = None
    barrier(ctx)
    try:
        store = dist.distributed_c10d._get_default_store()
    except Exception as e:  # noqa: BLE001 - an optional leg: every rank sees the same failure and moves on
        return {"skipped": "no rendezvous store to wait on: %r" % (e,)}
    if ctx.rank == 0:
        try:
            spent = time.time() - ctx.t_start
            if spent > 0.4 * args.time_budget:
                result = {"skipped": "%.0f s of the %.0f s budget spent before it" % (spent, args.time_budget)}
            elif not ctx.shared and torch.cuda.device_count() < world:
                result = {"skipped": "rank 0 sees %d device(s): the launcher masks the others" % torch.cuda.device_count()}
            else:
                devices = [0] * world if ctx.shared else list(range(world))
                r = single_process_steps(ia, torch, q, devices, text, n, m, n_batches, args.steps, args.warmup, host_call=False)
                result = {"what": "the same steps from ONE host process: fmx_replicate + fmx_count_batch_multi_dev (the C ABI's replica "
                                  "calls, per-device worker threads), measured by rank 0 while the other ranks idle",
                          "devices": devices, "ms_per_step": r["ms_per_step"], "patterns_per_s": r["patterns_per_s"],
                          "replicate_s": r["replicate_s"], "count_checksum_all_shards": r["count_checksums"][0]}
                torch.cuda.set_device(ctx.local_rank)
        except Exception as e:  # noqa: BLE001 - reported on the line, never fatal for the contract's own measurement
            log("[bench] single-process leg FAILED: %r" % (e,))
            result = {"error": repr(e)[:300]}
        finally:
            store.set(key, "1")
    else:
        store.wait([key], datetime.timedelta(seconds=max(60.0, args.time_budget)))
    barrier(ctx)
    return result


# ---------------------------------------------------------------------------------------------------------------
# workload `count` — BASELINE.json configs[1] (+ configs[2], [3] as `secondary` at N = 1)
# ---------------------------------------------------------------------------------------------------------------
def run_count(ctx, args):
    from index4j_amd import workload
    from index4j_amd.shard import gather_concat, shard_range

    ia, torch, dist, dev = ctx.ia, ctx.torch, ctx.dist, ctx.dev
    m, n, world = args.pattern_len, args.patterns, ctx.world
    text = fm = path = None
    if ctx.rank == 0:
        text, fm, path = build_or_load_index(ia, args.text_log2, args.sample_rate, args.cache_dir,
                                             build_device=None if ctx.dry else ctx.local_rank)
    q, blob_buf, image_bytes = attach_everywhere(ctx, fm)

    # ---- batches: each ONE batch of world x n patterns made on rank 0, handed out in contiguous shards ----
    n_batches = max(1, args.batches)
    host_batches = []
    distinct_patterns = None
    d_pats = []
    for b in range(n_batches):
        pat = None
        if ctx.rank == 0:
            pat, _off, _pos = workload.count_batch_patterns(text, world * n, m, seed=workload.PATTERN_SEED + b)
            host_batches.append(pat)
            if b == 0 and m <= 8:  # (rank 0's own shard: what one GPU's launch sees)
                rows = np.zeros((n, 8), np.uint16)
                rows[:, :m] = pat[: n * m].reshape(n, m)
                distinct_patterns = int(len(np.unique(rows.view(np.uint64), axis=0)))
        d_pats.append(hand_out_patterns(ctx, pat, m, world * n))
    lo, hi = shard_range(world * n, world, ctx.rank)
    assert hi - lo == n
    off_host = (np.arange(n + 1, dtype=np.int64) * m).astype(np.int32)
    d_off = torch.from_numpy(off_host).to(dev)
    d_cnt = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(n_batches)]
    d_lf = torch.zeros(n, dtype=torch.int32, device=dev)
    d_st = torch.zeros(n, dtype=torch.int32, device=dev)
    stream = None if ctx.dry else torch.cuda.current_stream()
    sp = None if ctx.dry else C.c_void_p(stream.cuda_stream)

    def step(b, with_steps=False):
        if ctx.dry:
            return
        check_rc(ia, ia.lib.fmx_count_batch_dev(q.handle, d_pats[b].data_ptr(), d_off.data_ptr(), n, d_cnt[b].data_ptr(),
                                                d_lf.data_ptr() if with_steps else None,
                                                d_st.data_ptr() if with_steps else None, sp), "fmx_count_batch_dev")

    lf_steps = []
    checksums = []
    for b in range(n_batches):  # also yields the exact LF-step count of every batch
        step(b, True)
        if not ctx.dry:
            torch.cuda.synchronize()
            if int(d_st.max().item()) != 0:
                raise RuntimeError("unexpected per-query status in the benchmark batch")
        lf_steps.append(int(d_lf.sum(dtype=torch.int64).item()))
        checksums.append(int(d_cnt[b].sum(dtype=torch.int64).item()))
    # the LF-steps k_count really evaluates (without those the suffix table answers), exact for every batch and rank
    lf_executed = list(lf_steps)
    if not ctx.dry:
        ia.lib.fmx_set_option(b"lf_steps_executed_only", 1)
        try:
            for b in range(n_batches):
                step(b, True)
                torch.cuda.synchronize()
                lf_executed[b] = int(d_lf.sum(dtype=torch.int64).item())
        finally:
            ia.lib.fmx_set_option(b"lf_steps_executed_only", 0)
    for i in range(args.warmup):
        step(i % n_batches)

    barrier(ctx)
    ctx.setup_s = time.time() - ctx.t_start  # process start -> timed loop: text, index build, image broadcast(s), batches, warm-up
    t0 = time.perf_counter()
    if not ctx.dry:
        ev0, ev1 = hip_events(torch)
        ev0.record(stream)
    for i in range(args.steps):
        step(i % n_batches)  # the whole hot path: plan stage of the batch + k_count
    if not ctx.dry:
        ev1.record(stream)
    barrier(ctx)
    wall = time.perf_counter() - t0
    step_ms = 0.0 if ctx.dry else ev0.elapsed_time(ev1) / args.steps

    # the dominant kernel alone, HIP events on its stream.  A batch the library plans (suffix order first: fmx.h
    # fmx_count_batch_is_planned) is counted over its processing order (plan made once, k_count launched `reps` times); a batch
    # it counts in the caller's order IS one k_count launch per call.
    kernel_ms_per_batch = []
    planned = False if ctx.dry else bool(ia.lib.fmx_count_batch_is_planned(q.handle, n))
    if not ctx.dry:
        reps = max(1, args.steps // n_batches)
        for b in range(n_batches):
            plan = C.c_void_p()
            if planned:
                check_rc(ia, ia.lib.fmx_count_plan_dev(q.handle, d_pats[b].data_ptr(), d_off.data_ptr(), n, C.byref(plan), sp),
                         "fmx_count_plan_dev")
            torch.cuda.synchronize()
            e0, e1 = hip_events(torch)
            e0.record(stream)
            for _ in range(reps):
                if planned:
                    check_rc(ia, ia.lib.fmx_count_ordered_dev(q.handle, d_pats[b].data_ptr(), d_off.data_ptr(), plan, n,
                                                              d_cnt[b].data_ptr(), None, None, sp), "fmx_count_ordered_dev")
                else:
                    step(b)
            e1.record(stream)
            torch.cuda.synchronize()
            kernel_ms_per_batch.append(e0.elapsed_time(e1) / reps)
            if int(d_cnt[b].sum(dtype=torch.int64).item()) != checksums[b]:
                raise RuntimeError("counts changed between launches")
    kernel_ms = float(np.mean(kernel_ms_per_batch)) if kernel_ms_per_batch else 0.0

    # The index's suffix table (fmx.h: fmx_suffix_table_info) answers the first 2 * (chars - 1) rank evaluations of every
    # pattern with one load.  The same steps once more with launches told to ignore it (option), for the record.
    table_chars, table_bytes = (0, 0) if ctx.dry else q.suffix_table_info()
    window_bytes = 0 if ctx.dry else q.window_cells_bytes()  # the window directory (fmx.h: fmx_window_cells_info): used by the LF-walks, resident all the same
    without_table = None
    if not ctx.dry and table_chars and not args.profiling:
        ia.lib.fmx_set_option(b"suffix_table", 0)
        try:
            for i in range(max(2, args.warmup)):
                step(i % n_batches)
            torch.cuda.synchronize()
            e0, e1 = hip_events(torch)
            e0.record(stream)
            for i in range(args.steps):
                step(i % n_batches)
            e1.record(stream)
            torch.cuda.synchronize()
            without_table = {"ms_per_step": e0.elapsed_time(e1) / args.steps}
            for b in range(n_batches):
                if int(d_cnt[b].sum(dtype=torch.int64).item()) != checksums[b]:
                    raise RuntimeError("counts differ without the suffix table")
        finally:
            ia.lib.fmx_set_option(b"suffix_table", 1)

    # ... and with a table one level DEEPER (option suffix_table_image_fraction 2: the table may take half the image's bytes instead
    # of an eighth — a deployment's choice of bytes against time, like the window directory): the same index made resident once
    # more under that option, the same K steps, counts checked.  N = 1 only (rank 0 holds the serialized index).
    deeper_table = None
    if not ctx.dry and path and world == 1 and not args.profiling:
        q2 = None
        try:
            check_rc(ia, ia.lib.fmx_set_option(b"suffix_table_image_fraction", 2), "fmx_set_option")
            try:
                q2 = ia.FmIndex.read(open(path, "rb").read(), device=ctx.local_rank)
            finally:
                ia.lib.fmx_set_option(b"suffix_table_image_fraction", 8)
            chars2, bytes2 = q2.suffix_table_info()
            if chars2 > table_chars:
                def step2(b):
                    check_rc(ia, ia.lib.fmx_count_batch_dev(q2.handle, d_pats[b].data_ptr(), d_off.data_ptr(), n, d_cnt[b].data_ptr(),
                                                            None, None, sp), "fmx_count_batch_dev")
                for i in range(max(2, args.warmup)):
                    step2(i % n_batches)
                torch.cuda.synchronize()
                e0, e1 = hip_events(torch)
                e0.record(stream)
                for i in range(args.steps):
                    step2(i % n_batches)
                e1.record(stream)
                torch.cuda.synchronize()
                for b in range(n_batches):
                    if int(d_cnt[b].sum(dtype=torch.int64).item()) != checksums[b]:
                        raise RuntimeError("counts differ with the deeper suffix table")
                deeper_table = {"chars": chars2, "bytes": bytes2, "ms_per_step": e0.elapsed_time(e1) / args.steps,
                                "option": "suffix_table_image_fraction = 2 (default 8)"}
        except Exception as e:  # noqa: BLE001 - an extra figure: reported, never fatal
            log("[bench] deeper-table leg FAILED: %r" % (e,))
            deeper_table = {"error": repr(e)[:200]}
        finally:
            if q2 is not None:
                q2.close()

    # What the plan stage buys NET: the same K steps with the batch counted in the caller's order (no plan
    # stage: k_count maps the characters itself) — the library's policy plans this batch because that is the faster of the two.
    plan_net = None
    if not ctx.dry and planned and not args.profiling:
        ia.lib.fmx_set_option(b"plan_sa_min", 2**31 - 1)
        try:
            for i in range(max(2, args.warmup)):
                step(i % n_batches)
            torch.cuda.synchronize()
            e0, e1 = hip_events(torch)
            e0.record(stream)
            for i in range(args.steps):
                step(i % n_batches)
            e1.record(stream)
            torch.cuda.synchronize()
            unplanned_ms = e0.elapsed_time(e1) / args.steps
            for b in range(n_batches):
                if int(d_cnt[b].sum(dtype=torch.int64).item()) != checksums[b]:
                    raise RuntimeError("counts differ in the caller's order")
            plan_net = {"planned_ms_per_step": step_ms, "callers_order_ms_per_step": unplanned_ms,
                        "plan_stage_saves": 1.0 - step_ms / unplanned_ms,
                        "what": "the timed step (plan stage + k_count over the plan's order) against the same batches counted in the "
                                "caller's order (one k_count launch, no plan stage), one stream"}
        finally:
            ia.lib.fmx_set_option(b"plan_sa_min", 786432)

    # Beside the contract's line (one batch after the other on one stream): the same K steps with TWO batches in
    # flight — step i on stream i mod 2, as a service with several clients would issue them — so that one batch's plan
    # stage overlaps the other's k_count.  Reported as `overlapped`, never as `value`.
    overlapped = None
    if not ctx.dry and args.overlap_streams > 1 and not args.profiling:
        side = [torch.cuda.Stream(device=dev) for _ in range(args.overlap_streams)]

        def step_on(i):
            b = i % n_batches
            check_rc(ia, ia.lib.fmx_count_batch_dev(q.handle, d_pats[b].data_ptr(), d_off.data_ptr(), n, d_cnt[b].data_ptr(), None,
                                                    None, C.c_void_p(side[i % len(side)].cuda_stream)), "fmx_count_batch_dev")

        for i in range(max(args.warmup, 2 * len(side))):
            step_on(i)
        barrier(ctx)
        t1 = time.perf_counter()
        for i in range(args.steps):
            step_on(i)
        barrier(ctx)
        wall2 = time.perf_counter() - t1
        for b in range(n_batches):
            if int(d_cnt[b].sum(dtype=torch.int64).item()) != checksums[b]:
                raise RuntimeError("counts changed in the overlapped run")
        overlapped = {"streams": len(side), "wall_s_this_rank": wall2}

    single = None
    if world > 1 and dist is not None and not ctx.dry and not args.no_single_process_leg and not args.profiling:
        single = single_process_leg(ctx, args, q, text, n, m, n_batches)
    lf_local = sum(lf_steps[i % n_batches] for i in range(args.steps))
    lf_exec_local = sum(lf_executed[i % n_batches] for i in range(args.steps))
    seen = [[0, ctx.local_rank, ctx.local_rank]]
    gathered = None
    if dist is not None:
        from index4j_amd.shard import ranks_seen

        cdev = ctx.cdev
        tw = torch.tensor([wall], dtype=torch.float64, device=cdev)
        tmin = tw.clone()
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        wall_ranks = {"min": float(tmin.item()) * 1e3 / args.steps, "max": float(tw.item()) * 1e3 / args.steps}
        wall = float(tw.item())
        if overlapped:
            tw2 = torch.tensor([overlapped["wall_s_this_rank"]], dtype=torch.float64, device=cdev)
            dist.all_reduce(tw2, op=dist.ReduceOp.MAX)
            overlapped["wall_s_this_rank"] = float(tw2.item())
        tot = torch.tensor([lf_local, lf_exec_local], dtype=torch.int64, device=cdev)
        dist.all_reduce(tot)
        lf_total, lf_exec_total = int(tot[0].item()), int(tot[1].item())
        seen = ranks_seen(dist, cdev, ctx.local_rank, ctx.local_rank if not ctx.dry else -1)
        # final gather of the shards' results on rank 0 (outside the timed region: the shards are independent)
        gathered = gather_concat(dist, d_cnt[0], [n] * world, cdev)
    else:
        lf_total, lf_exec_total = lf_local, lf_exec_local
        wall_ranks = {"min": wall * 1e3 / args.steps, "max": wall * 1e3 / args.steps}
    # At N > 1 the same launch also measures BASELINE.json configs[4] (the 8M-pattern batch over the 2 GiB text's 8
    # segment indexes, strong scaling): one `bench.py --gpus N` yields the weak-scaling headline and this figure
    segments_line = None
    segments_skipped = None
    if world > 1 and not args.no_secondary:
        # bounded: the driver gives the whole command a limit; rank 0 builds 8 more images and sends 1.35 GB for this block.  It is
        # skipped (and the line says so) once more than half of --time-budget is gone — every rank takes rank 0's decision
        spent = torch.tensor([time.time() - ctx.t_start], dtype=torch.float64, device=ctx.cdev)
        dist.broadcast(spent, 0)
        if float(spent.item()) > 0.5 * args.time_budget:
            segments_skipped = "configs[4] block skipped: %.0f s of the %.0f s budget spent before it" % (float(spent.item()), args.time_budget)
            log("[bench] " + segments_skipped)
        else:
            segments_line = run_segments(ctx, args)
    if ctx.rank != 0:
        return None
    if len(seen) != args.gpus or sorted(r[0] for r in seen) != list(range(args.gpus)):
        raise RuntimeError("ranks seen %r do not match --gpus %d" % (seen, args.gpus))

    # ---- parity of the run itself + algorithmic bytes per LF-step from the oracle's counting mode ----
    bytes_per_step = levels_per_step = None
    exec_steps_launch = exec_bytes_launch = None
    table_steps = 0
    oracle_checked = 0
    base = None
    ref = orc = None
    if not ctx.dry and not args.no_cpu_baseline:
        orc = oracle_module()
        ref = orc.OracleFmIndex.read(open(path, "rb").read())
        cores = os.cpu_count() or 1
        all0 = gathered if gathered is not None else d_cnt[0].cpu().numpy()
        off_all = (np.arange(world * n + 1, dtype=np.int64) * m).astype(np.int32)
        orc.counters_reset()
        oc, ost = ref.count_batch(host_batches[0], off_all, threads=cores)  # EVERY pattern of batch 0, all ranks' shards
        cnt = orc.counters()
        if not (all0 == oc).all() or int(ost.max()) != 0:
            raise RuntimeError("GPU counts differ from the oracle on batch 0")
        oracle_checked += world * n
        bytes_per_step = cnt["alg_bytes"] / max(1, cnt["lf_steps"])
        levels_per_step = cnt["wt_levels"] / max(1, cnt["lf_steps"])
        table_steps = table_alg_bytes = 0
        if table_chars and m >= table_chars:
            # what the table answers = ALL steps of the same patterns cut to their last `table_chars` characters (rank 0's shard)
            tail = np.ascontiguousarray(host_batches[0][: n * m].reshape(n, m)[:, m - table_chars:]).reshape(-1)
            off_tail = (np.arange(n + 1, dtype=np.int64) * table_chars).astype(np.int32)
            orc.counters_reset()
            ref.count_batch(tail, off_tail, threads=cores)
            ct = orc.counters()
            table_steps, table_alg_bytes = ct["lf_steps"], ct["alg_bytes"]
            orc.counters_reset()
            ref.count_batch(host_batches[0][: n * m], off_host, threads=cores)
            full_shard = orc.counters()
            exec_steps_launch = full_shard["lf_steps"] - table_steps
            exec_bytes_launch = full_shard["alg_bytes"] - table_alg_bytes
        else:
            exec_steps_launch = exec_bytes_launch = None
        oracle_checksum = int(oc.astype(np.int64).sum())
        for b in range(1, n_batches):  # rank 0's shard of the other batches
            oc_b, _ = ref.count_batch(host_batches[b][: n * m], off_host, threads=cores)
            if not (d_cnt[b].cpu().numpy() == oc_b).all():
                raise RuntimeError("GPU counts differ from the oracle on batch %d" % b)
            oracle_checked += n
        if world == 1:
            base = cpu_baseline(ref, host_batches[0], off_host, args.cpu_budget, orc)
            ser_framed = os.path.join(args.cache_dir, "fmx_framed_%d.ser" % os.getpid())
            try:
                with open(ser_framed, "wb") as f:
                    f.write(fm.write(True))
                base["index4j_jvm"] = jvm_leg(ser_framed, host_batches[0], m, args.cache_dir)
            finally:
                if os.path.exists(ser_framed):
                    os.unlink(ser_framed)
    else:
        oracle_checksum = None

    ms_per_step = wall * 1e3 / args.steps
    patterns_per_s = world * n * args.steps / wall
    roof = None
    if bytes_per_step:
        # a measured streaming figure beside the vendor peak: device-to-device copy of 1 GiB, read + write
        src = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
        dst = torch.empty_like(src)
        dst.copy_(src)
        e0, e1 = hip_events(torch)
        e0.record()
        for _ in range(5):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        copy_gbs = 5 * 2 * (1 << 30) / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del src, dst
        lf_per_launch = float(np.mean(lf_steps))
        alg_bytes = bytes_per_step * lf_per_launch
        alg_bytes_reference = alg_bytes
        if exec_bytes_launch is not None:  # only the LF-steps k_count really executes count (batch 0's figures)
            alg_bytes = float(exec_bytes_launch)
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        pmc, why = pmc_counters(args.text_log2, n, args.sample_rate)
        # FETCH_SIZE tallies a fabric read request at 64 bytes; tools/calibrate_fetch.py measures what a request carries on
        # this GPU — 128 bytes for a streamed read, a 64-byte sector for a scattered 16-byte load that misses (its `halves` case,
        # round 5; rounds 3 / 4 had assumed whole lines and doubled the figure) — and stores the factor beside the counters
        cal = (pmc or {}).get("calibration") or {}
        fetch_factor = cal.get("fabric_bytes_per_FETCH_SIZE_byte") or 1.0
        traffic_raw = (pmc["FETCH_SIZE_KiB"] + pmc["WRITE_SIZE_KiB"]) * 1024.0 if pmc else None
        traffic = (pmc["FETCH_SIZE_KiB"] * fetch_factor + pmc["WRITE_SIZE_KiB"]) * 1024.0 if pmc else None
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": "k_count" if planned else "k_count (the batch is counted in the caller's order: no plan stage, the kernel maps "
                                                    "the characters itself — fmx_count_batch_is_planned)",
                "plan_stage": planned,
                "what_frac_means": "ALGORITHMIC bytes of the reference's layout (oracle counting mode) per k_count launch / "
                                   "launch time / peak: distance from a perfect streaming of the reference's own reads, not "
                                   "HBM saturation — the image is L2 / Infinity-Cache resident, see traffic_frac",
                "kernel_ms": kernel_ms, "kernel_ms_per_batch": kernel_ms_per_batch,
                "kernel_ms_spread": (max(kernel_ms_per_batch) - min(kernel_ms_per_batch)) / kernel_ms,
                "step_ms_incl_plan": step_ms, "frac_whole_step": alg_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "alg_bytes_per_lf_step": bytes_per_step, "lf_steps_per_launch": lf_per_launch,
                "lf_steps_executed_per_launch": float(np.mean(lf_executed)), "lf_steps_executed_per_batch": lf_executed,
                "alg_bytes_executed_per_launch": alg_bytes, "alg_bytes_reference_per_launch": alg_bytes_reference,
                "suffix_table_note": None if exec_steps_launch is None else
                "achieved / frac count ONLY the algorithmic bytes of the LF-steps k_count executes (oracle counting mode on the "
                "batch minus the same on the patterns' last %d characters, which the suffix table answers)" % table_chars,
                "lf_steps_per_batch": lf_steps, "wt_levels_per_lf_step": levels_per_step,
                "traffic_frac": traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if traffic else None,
                "traffic_note": why if traffic is None else
                "FETCH_SIZE x %.2f + WRITE_SIZE of the committed rocprofv3 --pmc passes (profiles/pmc_latest.json, kernel sources %s); "
                "the factor is tools/calibrate_fetch.py's: %s" % (fetch_factor, kernel_source_sha(),
                                                                 cal.get("reading", "no calibration stored: raw counter values")),
                "traffic_raw_counters": traffic_raw,
                # the kernel's line fills per second against what the chip delivers to this access pattern (random 16-byte loads)
                "fabric_line_fills_per_launch": pmc["FETCH_SIZE_KiB"] * 1024.0 / 64.0 if pmc else None,
                "fabric_line_fills_per_lf_step_executed": pmc["FETCH_SIZE_KiB"] * 1024.0 / 64.0 / float(np.mean(lf_executed)) if pmc else None,
                "fabric_line_fills_Gper_s": pmc["FETCH_SIZE_KiB"] * 1024.0 / 64.0 / (kernel_ms * 1e-3) / 1e9 if pmc else None,
                "random_line_rate_Glines_per_s": (cal.get("random_line_rate_Glines_per_s") or {}).get("192_MiB"),
                "frac_of_random_line_rate": (pmc["FETCH_SIZE_KiB"] * 1024.0 / 64.0 / (kernel_ms * 1e-3) / 1e9 /
                                             cal["random_line_rate_Glines_per_s"]["192_MiB"])
                if pmc and (cal.get("random_line_rate_Glines_per_s") or {}).get("192_MiB") else None,
                "l1_line_bytes": pmc["TCP_TOTAL_CACHE_ACCESSES"] * 64.0 if pmc and pmc.get("TCP_TOTAL_CACHE_ACCESSES") else None,
                "l1_line_accesses_per_lf_step": pmc["TCP_TOTAL_CACHE_ACCESSES"] / float(exec_steps_launch or lf_per_launch)
                if pmc and pmc.get("TCP_TOTAL_CACHE_ACCESSES") else None,
                "l1_line_accesses_per_lf_step_note": "per EXECUTED LF-step of a launch",
                "image_bytes": image_bytes, "image_bytes_per_text_byte": image_bytes / float(1 << args.text_log2),
                "suffix_table_bytes": table_bytes, "window_directory_bytes": window_bytes,
                # everything resident; the window directory serves the LF-walks only — count(), this line's workload, reads the image
                # and the suffix table (resident_bytes_count_path_per_text_byte)
                "resident_bytes_per_text_byte": (image_bytes + table_bytes + window_bytes) / float(1 << args.text_log2),
                "resident_bytes_count_path_per_text_byte": (image_bytes + table_bytes) / float(1 << args.text_log2),
                "frac_reference_equivalent": alg_bytes_reference / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "measured_copy_GBps": copy_gbs, "frac_of_measured_copy": achieved / copy_gbs}
        ref_series_module().settle_frac(roof, kernel_ms, traffic, ratio_rule=False)
    secondary = None
    if world == 1 and ref is not None and not args.no_secondary:
        try:
            secondary = run_secondary(ctx, args, q, ref, orc, text, host_batches[0], off_host)
        except Exception as e:  # noqa: BLE001 - the headline's line is printed whatever happens to the rows beside it, with the reason
            log("[bench] secondary configs FAILED: %r" % (e,))
            secondary = [{"config": "BASELINE.json configs[2] / [3]", "ms": None, "error": repr(e)[:300]}]
        if secondary is not None and not args.no_segments_share and not args.profiling:
            try:
                secondary.append(segments_share_row(ctx, args))
            except Exception as e:  # noqa: BLE001 - as above: the headline's line survives a failed secondary block, and says so
                log("[bench] configs[4] per-GPU share FAILED: %r" % (e,))
                secondary.append({"config": "BASELINE.json configs[4] per-GPU share", "ms": None, "error": repr(e)[:300]})
    elif segments_line is not None:
        secondary = [segments_line]
    host_buffers = None
    if world == 1 and not ctx.dry and not args.profiling and oracle_checksum is not None:
        try:
            host_buffers = measure_host_buffers(ctx, args, q, host_batches[0][: n * m], off_host, d_cnt[0].cpu().numpy(), ms_per_step)
        except Exception as e:  # noqa: BLE001 - never `value`: its failure is reported, not fatal
            log("[bench] host-buffer measurement FAILED: %r" % (e,))
            host_buffers = {"error": repr(e)[:300]}
    # LF-steps the suffix table answered over the timed steps
    executed_less = lf_total - lf_exec_total  # counted by the kernel itself, every batch of every rank
    if not ctx.dry and exec_steps_launch is not None and exec_steps_launch != lf_executed[0]:
        raise RuntimeError("executed LF-steps: the kernel says %d, the oracle's accounting %d" % (lf_executed[0], exec_steps_launch))
    out = {
        "metric": "patterns/sec + LF-steps/sec, 1M x 8-char count() on 256 MiB log index",
        "value": None if ctx.dry else patterns_per_s,
        "unit": "patterns/s",
        "lf_steps_per_sec": None if ctx.dry else (lf_total - executed_less) / wall,
        "lf_steps_per_sec_reference_equivalent": None if ctx.dry else lf_total / wall,
        "n_gpus": world,
        "ranks_seen": seen,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "ms_per_step_ranks": wall_ranks,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int32",
        "data": "synthetic",
        "config": {"workload": "count() batch of %d random %d-char patterns per GPU on %d MiB synthetic log text, "
                               "sampleRate=%d (BASELINE.json configs[1]); %d distinct batches rotate through the timed loop"
                               % (n, m, (1 << args.text_log2) >> 20, args.sample_rate, n_batches),
                   "image": "compact" if args.image_compact else "expanded",
                   "text_chars": 1 << args.text_log2, "patterns_per_gpu": n, "pattern_len": m,
                   "sample_rate": args.sample_rate, "batches": n_batches,
                   # the synthetic log repeats itself: windows drawn from it at random coincide.  Every pattern is searched (no
                   # result is shared between equal patterns); equal and near-equal patterns read the same lines, which is what the
                   # plan's order and the caches turn into speed — a text of higher entropy gains less.
                   "distinct_patterns_batch0": distinct_patterns,
                   "parallelism": "one batch of %d patterns sharded x%d (contiguous shards from rank 0), index image "
                                  "broadcast and replicated" % (world * n, world),
                   "count_checksums_rank0": checksums, "count_checksum": checksums[0],
                   "gathered_checksum_all_ranks": int(gathered.astype(np.int64).sum()) if gathered is not None else None,
                   "oracle_checksum_batch0_all_ranks": oracle_checksum, "patterns_checked_vs_oracle": oracle_checked},
        "suffix_table": None if ctx.dry else {
            "chars": table_chars, "bytes": table_bytes,
            "what": "SA interval of every string of `chars` codes that occurs in the text, grown level by level by the index's own "
                    "rank code and hashed when the index becomes resident (fmx_suffix_table_info); a batch starts from it: one 16-byte "
                    "slot instead of %d rank evaluations per pattern. `lf_steps_per_sec` counts executed LF-steps only; `value` is "
                    "patterns/s" % (2 * max(0, table_chars - 1)),
            "lf_steps_answered_per_launch": int(table_steps) if bytes_per_step and exec_steps_launch is not None else None,
            "without_it": None if not without_table else {
                "ms_per_step": without_table["ms_per_step"],
                "patterns_per_s": n / (without_table["ms_per_step"] * 1e-3)},
            "one_level_deeper": deeper_table},
        "index_broadcast": getattr(ctx, "broadcast_times", None),
        "setup_s": getattr(ctx, "setup_s", None),
        "segments_block": segments_skipped,
        "overlapped": None if not overlapped else {
            "what": "the same %d steps with %d batches in flight (step i on stream i mod %d): one batch's plan stage overlaps "
                    "another's k_count; max over ranks; not the contract's `value`" % (args.steps, overlapped["streams"], overlapped["streams"]),
            "streams": overlapped["streams"], "ms_per_step": overlapped["wall_s_this_rank"] * 1e3 / args.steps,
            "patterns_per_s": world * n * args.steps / overlapped["wall_s_this_rank"],
            "lf_steps_per_s": (lf_total - executed_less) / overlapped["wall_s_this_rank"],
            "lf_steps_per_s_reference_equivalent": lf_total / overlapped["wall_s_this_rank"]},
        "roofline": roof,
        "cpu_baseline": base,
        "host_buffers": host_buffers,
        "secondary": secondary,
        "single_process": single,
        "plan_net": plan_net,
    }
    if ctx.dry:
        out["dry_run"] = True
    if getattr(ctx, "shared", False):
        out["rehearsal"] = "N ranks sharing ONE GPU, collectives over gloo on host tensors: checks the N > 1 code path end to end, measures nothing"
    return out


# ---------------------------------------------------------------------------------------------------------------
# --single-process: ONE host process drives N GPUs through the C ABI's replica calls (include/fmx.h "replicas") — what a Java
# host does over JNI: fmx_replicate (peer copies of the image out of device 0's HBM, every replica growing its own tables),
# then per step ONE fmx_count_batch_multi_dev call that has the library's per-device worker threads issue all shards' launches
# at once.  No torch.distributed, no collective.  Same workload, same contract line as the launcher form.
# ---------------------------------------------------------------------------------------------------------------
def single_process_steps(ia, torch, fm, devices, text, n, m, n_batches, steps, warmup, check_oracle=None, cores=1, host_call=True):
    """K steps of the headline workload over a replica set; returns a dict of measurements.  `fm` must be resident.
    check_oracle: an OracleFmIndex — every pattern of batch 0 (all shards) is checked against it."""
    from index4j_amd import workload

    world = len(devices)
    t0 = time.time()
    rs = ia.ReplicaSet(fm, devices)
    for d in sorted(set(devices)):
        torch.cuda.synchronize(d)
    replicate_s = time.time() - t0
    out = {"devices": list(devices), "replicate_s": replicate_s, "resident_bytes_per_replica": rs.resident_bytes()[0]}
    try:
        vp = C.c_void_p * world
        streams = [torch.cuda.Stream(device=torch.device("cuda", d)) for d in devices]
        sp = vp(*[st.cuda_stream for st in streams])
        off_host = (np.arange(n + 1, dtype=np.int64) * m).astype(np.int32)
        d_off = [torch.from_numpy(off_host).to(torch.device("cuda", d)) for d in devices]
        host_batches, d_pats, d_cnt = [], [], []
        for b in range(n_batches):
            pat, _off, _pos = workload.count_batch_patterns(text, world * n, m, seed=workload.PATTERN_SEED + b)
            host_batches.append(pat)
            rows = pat.view(np.int16).reshape(world, n * m)
            d_pats.append([torch.from_numpy(rows[r].copy()).to(torch.device("cuda", devices[r])) for r in range(world)])
            d_cnt.append([torch.zeros(n, dtype=torch.int32, device=torch.device("cuda", devices[r])) for r in range(world)])
        d_lf = [torch.zeros(n, dtype=torch.int32, device=torch.device("cuda", d)) for d in devices]
        d_st = [torch.zeros(n, dtype=torch.int32, device=torch.device("cuda", d)) for d in devices]
        ns = (C.c_int32 * world)(*([n] * world))
        a_off = vp(*[t.data_ptr() for t in d_off])
        a_pat = [vp(*[t.data_ptr() for t in d_pats[b]]) for b in range(n_batches)]
        a_cnt = [vp(*[t.data_ptr() for t in d_cnt[b]]) for b in range(n_batches)]
        a_lf, a_st = vp(*[t.data_ptr() for t in d_lf]), vp(*[t.data_ptr() for t in d_st])

        def sync():
            check_rc(ia, ia.lib.fmx_multi_synchronize(rs.handles, world, sp), "fmx_multi_synchronize")

        def step(b, with_steps=False):
            check_rc(ia, ia.lib.fmx_count_batch_multi_dev(rs.handles, world, a_pat[b], a_off, ns, a_cnt[b], a_lf if with_steps else None,
                                                          a_st if with_steps else None, sp), "fmx_count_batch_multi_dev")

        for d in sorted(set(devices)):
            torch.cuda.synchronize(d)
        lf_steps, checksums = [], []
        for b in range(n_batches):
            step(b, True)
            sync()
            if max(int(t.max().item()) for t in d_st) != 0:
                raise RuntimeError("unexpected per-query status in the benchmark batch")
            lf_steps.append(sum(int(t.sum(dtype=torch.int64).item()) for t in d_lf))
            checksums.append(sum(int(t.sum(dtype=torch.int64).item()) for t in d_cnt[b]))
        for i in range(warmup):
            step(i % n_batches)
        sync()
        t1 = time.perf_counter()
        for i in range(steps):
            step(i % n_batches)
        sync()
        wall = time.perf_counter() - t1
        for b in range(n_batches):
            if sum(int(t.sum(dtype=torch.int64).item()) for t in d_cnt[b]) != checksums[b]:
                raise RuntimeError("counts changed between launches")
        out.update({"wall_s": wall, "ms_per_step": wall * 1e3 / steps, "patterns_per_s": world * n * steps / wall,
                    "lf_steps_per_s_reference_equivalent": sum(lf_steps[i % n_batches] for i in range(steps)) / wall,
                    "count_checksums": checksums, "lf_steps_per_batch": lf_steps, "steps": steps, "warmup": warmup})
        all0 = np.concatenate([t.cpu().numpy() for t in d_cnt[0]])
        out["counts_batch0"] = all0
        out["host_batch0"] = host_batches[0]
        if check_oracle is not None:
            off_all = (np.arange(world * n + 1, dtype=np.int64) * m).astype(np.int32)
            oc, ost = check_oracle.count_batch(host_batches[0], off_all, threads=cores)
            if not (all0 == oc).all() or int(ost.max()) != 0:
                raise RuntimeError("single-process form: GPU counts differ from the oracle on batch 0")
            out["patterns_checked_vs_oracle"] = world * n
        if host_call:
            # the host-buffer form of the same step (what a JNI caller's arrays cost: PCIe in and out; never `value`): ONE
            # fmx_count_batch_multi over the whole world x n batch
            off_all = (np.arange(world * n + 1, dtype=np.int64) * m).astype(np.int32)
            cnt_h = np.zeros(world * n, np.int32)
            times = []
            for _ in range(5):
                t2 = time.perf_counter()
                check_rc(ia, ia.lib.fmx_count_batch_multi(rs.handles, world, host_batches[0].ctypes.data, off_all.ctypes.data, world * n,
                                                          cnt_h.ctypes.data, None, None), "fmx_count_batch_multi")
                times.append((time.perf_counter() - t2) * 1e3)
            if not (cnt_h == all0).all():
                raise RuntimeError("fmx_count_batch_multi differs from the device-resident form")
            out["host_buffers_ms_per_call"] = float(np.median(times))
    finally:
        rs.close()
    return out


def run_single_process(args, t_start):
    import torch

    import index4j_amd as ia

    if not torch.cuda.is_available() or ia.lib.fmx_device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    world = args.gpus
    have = torch.cuda.device_count()
    if args.share_one_gpu:
        devices = [0] * world
    elif have < world:
        log("[bench] --gpus %d --single-process but only %d HIP device(s) visible: refusing to print a mislabelled line" % (world, have))
        sys.exit(2)
    else:
        devices = list(range(world))
    if args.image_compact:
        check_rc(ia, ia.lib.fmx_set_option(b"image_compact", 1), "fmx_set_option")
    torch.cuda.set_device(0)
    m, n = args.pattern_len, args.patterns
    text, fm, path = build_or_load_index(ia, args.text_log2, args.sample_rate, args.cache_dir, build_device=0)
    fm.to_device(0)
    ref = None
    cores = os.cpu_count() or 1
    if not args.no_cpu_baseline:
        ref = oracle_module().OracleFmIndex.read(open(path, "rb").read())
    setup_s = time.time() - t_start
    r = single_process_steps(ia, torch, fm, devices, text, n, m, max(1, args.batches), args.steps, args.warmup, check_oracle=ref, cores=cores)
    image, table, window = r["resident_bytes_per_replica"]
    out = {
        "metric": "patterns/sec + LF-steps/sec, 1M x 8-char count() on 256 MiB log index",
        "value": r["patterns_per_s"], "unit": "patterns/s",
        "lf_steps_per_sec_reference_equivalent": r["lf_steps_per_s_reference_equivalent"],
        "n_gpus": world, "ranks_seen": [[k, k, d] for k, d in enumerate(devices)],
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "launch": "single-process: ONE host process, the C ABI's replica calls (fmx_replicate, fmx_count_batch_multi_dev: every shard's "
                  "launches issued by the library's per-device worker threads; no torch.distributed, no collective)",
        "setup_s": setup_s + r["replicate_s"],
        "config": {"workload": "count() batch of %d random %d-char patterns per GPU on %d MiB synthetic log text, sampleRate=%d "
                               "(BASELINE.json configs[1]); %d distinct batches rotate; single-process form over %d replica(s)"
                               % (n, m, (1 << args.text_log2) >> 20, args.sample_rate, max(1, args.batches), world),
                   "image": "compact" if args.image_compact else "expanded", "text_chars": 1 << args.text_log2, "patterns_per_gpu": n,
                   "pattern_len": m, "sample_rate": args.sample_rate, "batches": max(1, args.batches),
                   "count_checksum": r["count_checksums"][0], "patterns_checked_vs_oracle": r.get("patterns_checked_vs_oracle", 0)},
        "index_broadcast": {"broadcast_s": r["replicate_s"], "bytes": image, "kept": "peer copies (fmx_replicate)",
                            "note": "image copy + growth of each replica's suffix table and window directory, all replicas at once"},
        "host_buffers": {"ms_per_call": r.get("host_buffers_ms_per_call"),
                         "what": "fmx_count_batch_multi of the whole %d-pattern batch from pageable host arrays (median of 5)" % (world * n)},
        "resident_bytes_per_replica": {"image": image, "suffix_table": table, "window_directory": window},
        "roofline": None, "cpu_baseline": None,
    }
    if args.share_one_gpu:
        out["rehearsal"] = "N replicas sharing ONE GPU: checks the single-process N > 1 code path end to end, measures nothing"
    fm.close()
    emit(out)



SHARE_PATTERNS = 1 << 20  # configs[4]: 8,388,608 patterns over 8 GPUs


def segments_share_row(ctx, args, steps=8, warmup=2, check=4000):
    """One HBM-resident row on the N = 1 line: one GPU's share of BASELINE.json configs[4] — count() +
    locate(maxMatches 16) of 8,388,608 / 8 patterns over all 8 segment indexes.  The segment images (1.35 GB) do not fit the 256
    MiB Infinity Cache, so this row's bytes do come from HBM; every other row of the line works on an image the cache holds."""
    seg_args = argparse.Namespace(**vars(args))
    seg_args.patterns_total, seg_args.steps, seg_args.warmup, seg_args.segments_check = SHARE_PATTERNS, steps, warmup, check
    t0 = time.time()
    line = run_segments(ctx, seg_args)
    roof = line.get("roofline") or {}
    # HIP-event time of the one-pass call (as for every other secondary row); the wall-clock step of the segments workload — some
    # fifty launches per step issued by this thread — also counts the host's launch path, which a busy box stretches (6.3 -> 11.4 ms
    # seen once): kept beside it as ms_wall
    ms = ((roof.get("stage_ms_this_rank") or {}).get("count_and_locate_one_pass")) or line["ms_per_step"]
    alg = roof.get("alg_bytes_executed_per_step_per_gpu")
    row_roof = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "achieved": None if not alg else alg / (ms * 1e-3) / 1e9, "frac": None if not alg else alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "kernels": "8 x (k_count + k_segment_add_counts + walk order + k_locate_walk + k_segment_append_hits): "
                           "fmx_count_locate_segments_dev, one range search per segment for count() and locate() together",
                "stage_ms": roof.get("stage_ms_this_rank"), "frac_per_stage_algorithmic": roof.get("frac_per_stage"),
                "two_calls_ms": None if not roof.get("stage_ms_this_rank") else
                roof["stage_ms_this_rank"]["count"] + roof["stage_ms_this_rank"]["locate"],
                "image_bytes": (line.get("config") or {}).get("image_bytes_per_gpu"),
                "note": "both stages' algorithmic bytes (oracle counting mode on the checked sample, scaled to the batch, minus what the "
                        "segments' suffix tables answer) over the whole step's time; the image set is HBM-resident"}
    ref_series_module().settle_frac(row_roof, ms, pmc_row_lookup()("configs[4] share", SHARE_PATTERNS))
    log("[bench] configs[4] per-GPU share: %.2f ms per step (%.0f s with build and oracle sample)" % (ms, time.time() - t0))
    return {"config": "BASELINE.json configs[4] per-GPU share: count() + locate(maxMatches 16) in one pass of %d patterns over %d segment indexes of "
                      "2^%d chars, %.2f GB of images resident (beyond the Infinity Cache)"
                      % (SHARE_PATTERNS, args.segments, args.segment_log2, ((line.get("config") or {}).get("image_bytes_per_gpu") or 0) / 1e9),
            "ms": ms, "ms_wall": line["ms_per_step"], "patterns_per_s": SHARE_PATTERNS / ms * 1e3, "roofline": row_roof,
            "checked_vs_oracle": "%s patterns against the %d oracle indexes: counts, found, every position"
                                 % ((line.get("config") or {}).get("patterns_checked_vs_oracle"), args.segments),
            "count_checksum": (line.get("config") or {}).get("count_checksum_all_ranks"), "hits": (line.get("config") or {}).get("hits_all_ranks")}


def measure_host_buffers(ctx, args, q, pat, off, expect, step_ms):
    """The drop-in's real call path: fmx_count_batch with the caller's plain (pageable) arrays, what the JNI binding of
    FmIndex.count calls (bindings/jni/fmx_jni.c).  Wall time per call incl. both PCIe directions, beside the PCIe floor
    of the same bytes measured with pinned copies in this run.  Never `value`."""
    ia, torch, dev = ctx.ia, ctx.torch, ctx.dev
    n = len(off) - 1
    pat = np.ascontiguousarray(pat, dtype=np.uint16)
    off = np.ascontiguousarray(off, dtype=np.int32)
    counts = np.zeros(n, np.int32)
    status = np.zeros(n, np.int32)

    def call(pipe_min=None):
        if pipe_min is not None:
            ia.lib.fmx_set_option(b"host_pipeline_min", pipe_min)
        t0 = time.perf_counter()
        check_rc(ia, ia.lib.fmx_count_batch(q.handle, pat.ctypes.data, off.ctypes.data, n, counts.ctypes.data, None,
                                            status.ctypes.data), "fmx_count_batch")
        return (time.perf_counter() - t0) * 1e3

    try:
        first_calls = [call(), call()]  # untimed: the first two calls of a process pay ~6 ms each in their first result copies
        if not (counts == expect).all() or int(status.max()) != 0:
            raise RuntimeError("host-buffer counts differ from the device-pointer path")
        piped_all = [call() for _ in range(7)]
        piped = float(np.mean(piped_all))
        call(0)
        if not (counts == expect).all():
            raise RuntimeError("host-buffer counts (unpipelined) differ from the device-pointer path")
        plain = float(np.mean([call() for _ in range(5)]))
    finally:
        ia.lib.fmx_set_option(b"host_pipeline_min", 131072)
    # the same call with the caller's arrays pinned once (fmx_host_register: what a binding does with its direct buffers)
    registered = None
    regs = []
    try:
        for a in (pat, off, counts, status):
            check_rc(ia, ia.lib.fmx_host_register(a.ctypes.data, a.nbytes), "fmx_host_register")
            regs.append(a)
        call()
        if not (counts == expect).all():
            raise RuntimeError("host-buffer counts (registered buffers) differ from the device-pointer path")
        registered_all = [call() for _ in range(7)]
        registered = float(np.mean(registered_all))
    finally:
        for a in regs:
            ia.lib.fmx_host_unregister(a.ctypes.data)
    # what ONE query costs through the host entry points — a Java caller's count(char[]) / locate / extract is a batch of one:
    # microseconds per call (through one mapped pinned block: no copy calls; index4j's own count() of 8 characters is ~14 us on a core)
    scalar = None
    try:
        one_pat, one_off = pat[: off[1]].copy(), off[:2].copy()
        c1, s1 = np.zeros(1, np.int32), np.zeros(1, np.int32)
        l1, f1 = np.zeros(16, np.int32), np.zeros(1, np.int32)
        a1, b1 = np.array([1000], np.int32), np.array([1064], np.int32)
        d1, n1 = np.zeros(64, np.uint16), np.zeros(1, np.int32)
        calls = {
            "count": lambda: ia.lib.fmx_count_batch(q.handle, one_pat.ctypes.data, one_off.ctypes.data, 1, c1.ctypes.data, None, s1.ctypes.data),
            "locate16": lambda: ia.lib.fmx_locate_batch(q.handle, one_pat.ctypes.data, one_off.ctypes.data, 1, 16, l1.ctypes.data, 16,
                                                        f1.ctypes.data, None, s1.ctypes.data),
            "extract64": lambda: ia.lib.fmx_extract_batch(q.handle, a1.ctypes.data, b1.ctypes.data, 1, d1.ctypes.data, 64, 0, n1.ctypes.data,
                                                          None, s1.ctypes.data)}
        scalar = {}
        for name, f in calls.items():
            for _ in range(20):
                check_rc(ia, f(), name)
            t0 = time.perf_counter()
            for _ in range(200):
                f()
            scalar[name] = (time.perf_counter() - t0) / 200 * 1e6
        if int(c1[0]) != int(expect[0]):
            raise RuntimeError("the scalar count differs from the batch's first count")
    except Exception as e:  # noqa: BLE001 - an extra figure
        scalar = {"error": repr(e)[:200]}
    # PCIe rates of this box: pinned copies of 64 MiB, each direction
    hp = torch.empty(64 << 20, dtype=torch.uint8).pin_memory()
    dd = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    rates = {}
    for name, (dst, src) in (("h2d", (dd, hp)), ("d2h", (hp, dd))):
        dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        rates[name] = 3 * (64 << 20) / (time.perf_counter() - t0) / 1e9
    lens = np.diff(off)
    uniform = bool((lens == lens[0]).all())
    bytes_in = pat.nbytes + (0 if uniform else off.nbytes)
    bytes_out = counts.nbytes + status.nbytes
    floor_in = bytes_in / rates["h2d"] / 1e6
    floor = max(floor_in, step_ms)
    return {"what": "fmx_count_batch (host buffers, pageable numpy arrays; counts + statuses back) of batch 0: the JNI binding's call "
                    "path.  Chunks of 262,144 patterns travel while the previous chunk is counted (2 streams), results return through "
                    "pinned staging, offsets of equal-length runs are made on the device",
            "stat": "means of 7 calls after 2 untimed ones (%.1f, %.1f ms) (medians: pageable %.3f, registered %.3f ms; minima: %.3f, %.3f)"
                    % (first_calls[0], first_calls[1], float(np.median(piped_all)), float(np.median(registered_all)), min(piped_all),
                       min(registered_all)),
            "ms_per_call_all": [round(x, 3) for x in piped_all], "ms_per_call_registered_buffers_all": [round(x, 3) for x in registered_all],
            "ms_per_call_median": float(np.median(piped_all)), "ms_per_call_registered_buffers_median": float(np.median(registered_all)),
            "scalar_call_us": scalar,
            "patterns": n, "ms_per_call": piped, "patterns_per_s": n / piped * 1e3,
            "ms_per_call_unpipelined": plain, "ms_per_call_registered_buffers": registered,
            "ratio_registered_to_max_of_floor_and_device_step": registered / floor,
            "bytes_in": bytes_in, "bytes_out": bytes_out, "offsets_made_on_device": uniform,
            "pinned_copy_GBps": rates, "pcie_floor_ms_in": floor_in, "pcie_floor_ms_in_plus_out": floor_in + bytes_out / rates["d2h"] / 1e6,
            "device_step_ms": step_ms, "ratio_to_max_of_floor_and_device_step": piped / floor}


def run_secondary(ctx, args, q, ref, orc, text, pat, off):
    """BASELINE.json configs[2] and [3] on the same GPU, outside the headline's timed region: HIP-event time, every
    result checked against the oracle in this run, LF-steps and algorithmic bytes from the oracle's counting mode"""
    ia, torch, dev = ctx.ia, ctx.torch, ctx.dev
    m, cores = args.pattern_len, os.cpu_count() or 1
    stream = torch.cuda.current_stream()
    sp = C.c_void_p(stream.cuda_stream)
    K = min(100_000, len(off) - 1)
    M = 16
    res = []
    rs = ref_series_module()
    traffic_of = pmc_row_lookup()

    def timed(fn, reps):
        # MEAN over 3 x reps back-to-back calls between one HIP-event pair: the headline's standard (minima until round 3)
        fn()
        torch.cuda.synchronize()
        e0, e1 = hip_events(torch)
        e0.record(stream)
        for _ in range(3 * reps):
            fn()
        e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / (3 * reps)

    d_pat = torch.from_numpy(np.ascontiguousarray(pat[: K * m]).view(np.int16)).to(dev)
    d_off = torch.from_numpy(np.ascontiguousarray(off[: K + 1])).to(dev)
    d_locs = torch.zeros(K * M, dtype=torch.int32, device=dev)
    d_found = torch.zeros(K, dtype=torch.int32, device=dev)
    d_lf = torch.zeros(K, dtype=torch.int32, device=dev)
    d_st = torch.zeros(K, dtype=torch.int32, device=dev)
    d_rng = torch.zeros(2 * K, dtype=torch.int32, device=dev)

    def locate(index, with_lf=True):
        check_rc(ia, ia.lib.fmx_locate_batch_dev(index.handle, d_pat.data_ptr(), d_off.data_ptr(), K, M, d_locs.data_ptr(), M,
                                                 d_found.data_ptr(), d_lf.data_ptr() if with_lf else None, d_st.data_ptr(),
                                                 d_rng.data_ptr(), sp), "fmx_locate_batch_dev")

    # ---- configs[2]: locate, maxMatches 16, sampleRate-32 index ----
    d_lf.zero_()
    d_st.zero_()
    locate(q)
    torch.cuda.synchronize()
    locs = d_locs.cpu().numpy().reshape(K, M)
    found = d_found.cpu().numpy()
    lf_gpu = int(d_lf.sum(dtype=torch.int64).item())
    orc.counters_reset()
    olocs, ofound, ost = ref.locate_batch(pat[: K * m], off[: K + 1], M, threads=cores)
    c = orc.counters()
    live = np.arange(M)[None, :] < found[:, None]
    if not ((found == ofound).all() and (locs[live] == olocs[live]).all() and int(d_st.max().item()) == 0 and lf_gpu == c["lf_steps"]):
        raise RuntimeError("locate differs from the oracle")
    ms = timed(lambda: locate(q, False), 5)
    alg = c["alg_bytes"]
    # what the suffix table answers of the range search (the last `table_chars` characters of every pattern) is not executed
    table_chars = q.suffix_table_info()[0]
    table_alg = table_steps = 0
    if table_chars:
        tail = np.ascontiguousarray(pat[: K * m].reshape(K, m)[:, m - table_chars:]).reshape(-1)
        orc.counters_reset()
        ref.count_batch(tail, (np.arange(K + 1, dtype=np.int64) * table_chars).astype(np.int32), threads=cores)
        tc = orc.counters()
        table_alg, table_steps = tc["alg_bytes"], tc["lf_steps"]
    alg_exec = alg - table_alg
    res.append({"config": "BASELINE.json configs[2]: locate() of %d patterns, maxMatches %d, 256 MiB text, sampleRate %d"
                          % (K, M, args.sample_rate),
                "ms": ms, "patterns_per_s": K / ms * 1e3, "hits": int(found.sum()), "hits_per_s": int(found.sum()) / ms * 1e3,
                "lf_steps": c["lf_steps"], "lf_steps_executed": c["lf_steps"] - table_steps, "lf_steps_per_s": c["lf_steps"] / ms * 1e3,
                "alg_bytes_per_lf_step": alg / max(1, c["lf_steps"]),
                "roofline": {"bound": "hbm", "achieved": alg_exec / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": alg_exec / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "frac_reference_equivalent": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "kernels": "k_count + k_walk_hist + k_plan_scatter + k_plan_fine + k_locate_walk",
                             "note": "frac counts the LF-steps the kernels EXECUTE (the suffix table answers the range search of a "
                                     "pattern's last %d characters) at the oracle's algorithmic bytes per step, as if every step read "
                                     "HBM: hits of equal and nested ranges are walked side by side (walk order) and share their lines in "
                                     "L1 / L2, which is how the figure can pass what streaming from HBM allows" % table_chars},
                "checked_vs_oracle": "all %d patterns: found, every position (SA order), LF-step total" % K})
    rs.settle_frac(res[-1]["roofline"], ms, traffic_of("configs[2]", K))

    # ---- configs[3]: extractUntilBoundary('\n') of the first hit of each pattern, sampleRate-64 index ----
    _t, fm64, path64 = build_or_load_index(ia, args.text_log2, 64, args.cache_dir, build_device=ctx.local_rank)
    fm64.to_device(ctx.local_rank)
    ref64 = orc.OracleFmIndex.read(open(path64, "rb").read())
    d_st.zero_()
    locate(fm64, False)
    torch.cuda.synchronize()
    froms = np.ascontiguousarray(d_locs.cpu().numpy().reshape(K, M)[:, 0])
    cap = 1024
    d_from = torch.from_numpy(froms).to(dev)
    d_dst = torch.zeros(K * cap, dtype=torch.int16, device=dev)
    d_len = torch.zeros(K, dtype=torch.int32, device=dev)
    d_aux = torch.zeros(K, dtype=torch.int32, device=dev)

    def boundary(with_lf=True):
        check_rc(ia, ia.lib.fmx_extract_boundary_batch_dev(fm64.handle, d_from.data_ptr(), K, 10, 0, d_dst.data_ptr(), cap, 0,
                                                           d_len.data_ptr(), d_lf.data_ptr() if with_lf else None,
                                                           d_st.data_ptr(), d_aux.data_ptr(), sp), "fmx_extract_boundary_batch_dev")

    boundary()
    torch.cuda.synchronize()
    lf_gpu = int(d_lf.sum(dtype=torch.int64).item())
    orc.counters_reset()
    odst, olen, ost, oaux = ref64.extract_until_boundary_batch(0, froms, "\n", cap, threads=cores)
    c = orc.counters()
    dst = d_dst.cpu().numpy().view(np.uint16).reshape(K, cap)
    if not ((d_len.cpu().numpy() == olen).all() and (dst == odst).all() and (d_st.cpu().numpy() == ost).all()):
        raise RuntimeError("extractUntilBoundary differs from the oracle")
    ms = timed(lambda: boundary(False), 3)
    chars = int(olen.astype(np.int64).sum())
    alg = c["alg_bytes"]
    alg_exec = alg * (lf_gpu / float(max(1, c["lf_steps"])))  # the same bytes per step over the steps really walked
    res.append({"config": "BASELINE.json configs[3]: extractUntilBoundary('\\n') of %d hit locations, 256 MiB text, sampleRate 64, "
                          "destination %d chars" % (K, cap),
                "ms": ms, "queries_per_s": K / ms * 1e3, "chars": chars, "chars_per_s": chars / ms * 1e3,
                "lf_steps_reference": c["lf_steps"], "lf_steps_executed_on_gpu": lf_gpu,
                "lf_steps_per_s_reference_equivalent": c["lf_steps"] / ms * 1e3, "lf_steps_per_s_executed": lf_gpu / ms * 1e3,
                "alg_bytes_per_lf_step": alg / max(1, c["lf_steps"]),
                "roofline": {"bound": "hbm", "achieved": alg_exec / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": alg_exec / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "frac_reference_equivalent": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "kernels": "k_extract_boundary_group",
                             "note": "frac counts the LF-steps the kernel EXECUTES (it fetches every sample interval once) at the "
                                     "oracle's algorithmic bytes per step; frac_reference_equivalent the steps of the REFERENCE's walk "
                                     "(one re-seek per 4 characters, FM:697-743) over the same time"},
                "checked_vs_oracle": "all %d queries: lengths, statuses, whole destination rows" % K})
    rs.settle_frac(res[-1]["roofline"], ms, traffic_of("configs[3]", K))
    ms3_default = ms
    # ---- footprint: what is resident per text byte against what it buys ----
    # index4j exists for the space / time trade (README.md: the serialized index is 0.44-0.47 of the text,
    # FmIndexSerializedSizeBenchmark.java:57).  Three forms of the resident index, each with its bytes per text byte
    # (fmx_resident_bytes) and the time of configs[2] and configs[3] over it: the compact image (index4j's own RRR compression kept),
    # the expanded image, — the default, the rows above — the expanded image with the window directory in the form the default rule
    # gives it (here: the FLAT one, a word per position: every step of a walk one sector), and with the directory in cells.  Every form's
    # results are compared with the default form's (which the oracle checked above).
    if not args.profiling:
        try:
            text_bytes = float(1 << args.text_log2)
            path = index_cache_path(text, args.text_log2, args.sample_rate, args.cache_dir)  # (the headline's index, cached by run_count)
            locate(q, False)
            torch.cuda.synchronize()
            want_locs, want_found = d_locs.cpu().numpy().copy(), d_found.cpu().numpy().copy()
            want_dst, want_len = d_dst.cpu().numpy().copy(), d_len.cpu().numpy().copy()

            def resident(index):
                a, b, c3 = C.c_int64(0), C.c_int64(0), C.c_int64(0)
                check_rc(ia, ia.lib.fmx_resident_bytes(index.handle, C.byref(a), C.byref(b), C.byref(c3)), "fmx_resident_bytes")
                return a.value, b.value, c3.value

            forms = [{"form": "expanded image + window directory (the default rule's form: the rows above)", "configs2_ms": res[-2]["ms"],
                      "configs3_ms": ms3_default, "resident": resident(q), "resident64": resident(fm64)}]
            # (the default rule gives a text of this size the directory's FLAT form — 4 bytes per text byte, every step one sector;
            # window_cells = 1 asks for the cells' form — 1.26 bytes per text byte — by name)
            for form, compact, cells in (("expanded image + directory in CELLS (window_cells 1)", 0, 1), ("expanded image, no directory", 0, 0),
                                         ("compact image (RRR records), no directory", 1, 0)):
                check_rc(ia, ia.lib.fmx_set_option(b"image_compact", compact), "fmx_set_option")
                check_rc(ia, ia.lib.fmx_set_option(b"window_cells", cells), "fmx_set_option")
                try:
                    a32 = ia.FmIndex.read(open(path, "rb").read(), device=ctx.local_rank)
                    a64 = ia.FmIndex.read(open(path64, "rb").read(), device=ctx.local_rank)
                finally:
                    ia.lib.fmx_set_option(b"image_compact", 1 if args.image_compact else 0)
                    ia.lib.fmx_set_option(b"window_cells", env_options(ia).get("window_cells", 2))
                try:
                    d_locs.zero_()
                    locate(a32, False)
                    torch.cuda.synchronize()
                    if not ((d_found.cpu().numpy() == want_found).all() and (d_locs.cpu().numpy() == want_locs).all()):
                        raise RuntimeError("locate over the %s differs from the default form's" % form)
                    ms2_f = timed(lambda: locate(a32, False), 3)
                    d_dst.zero_()
                    check_rc(ia, ia.lib.fmx_extract_boundary_batch_dev(a64.handle, d_from.data_ptr(), K, 10, 0, d_dst.data_ptr(), cap, 0,
                                                                       d_len.data_ptr(), None, d_st.data_ptr(), d_aux.data_ptr(), sp),
                             "fmx_extract_boundary_batch_dev")
                    torch.cuda.synchronize()
                    if not ((d_len.cpu().numpy() == want_len).all() and (d_dst.cpu().numpy() == want_dst).all()):
                        raise RuntimeError("extractUntilBoundary over the %s differs from the default form's" % form)
                    ms3_f = timed(lambda: check_rc(ia, ia.lib.fmx_extract_boundary_batch_dev(
                        a64.handle, d_from.data_ptr(), K, 10, 0, d_dst.data_ptr(), cap, 0, d_len.data_ptr(), None, d_st.data_ptr(),
                        d_aux.data_ptr(), sp), "fmx_extract_boundary_batch_dev"), 3)
                    forms.append({"form": form, "configs2_ms": ms2_f, "configs3_ms": ms3_f, "resident": resident(a32), "resident64": resident(a64)})
                finally:
                    a32.close()
                    a64.close()
            for f in forms:
                f["resident_bytes_per_text_byte"] = sum(f["resident"]) / text_bytes              # the sampleRate-32 index (configs[2])
                f["resident_bytes_per_text_byte_s64"] = sum(f["resident64"]) / text_bytes        # the sampleRate-64 index (configs[3])
                f["image_suffix_table_directory_bytes"] = list(f.pop("resident"))
                f.pop("resident64")
            res.append({"config": "footprint: resident bytes per text byte of four forms of the index and what each costs configs[2] / [3]",
                        "ms": None, "forms": forms,
                        "index4j_serialized_bytes_per_text_byte": os.path.getsize(path) / text_bytes,
                        "checked": "every form's located positions and destination rows equal the default form's (oracle-checked above)"})
        except Exception as e:  # noqa: BLE001 - an extra row: its failure is reported on the row
            log("[bench] footprint rows FAILED: %r" % (e,))
            res.append({"config": "footprint: resident bytes per text byte of four forms of the index", "ms": None, "error": repr(e)[:300]})
    fm64.close()
    # ---- the headline's shape WITHOUT the generator's repetition: 1,048,576 DISTINCT 8-char patterns ----
    # The synthetic log repeats itself (328,091 distinct patterns in the headline batch); equal patterns side by side share
    # their sectors.  This row draws substrings until it has as many DISTINCT ones as the headline batch holds patterns.
    try:
        if args.profiling:
            # (tools/profile.sh averages the headline kernel's counters over its dispatches on the headline's GRID: this row runs
            # the same kernel on the same grid — r05_y's first counter pass read 376 instead of 197 MB per launch because of it)
            raise RuntimeError("not run under --profiling")
        n_d = len(off) - 1
        from index4j_amd import workload

        pool = np.zeros((0, m), np.uint16)
        for seed in range(1000, 1012):  # (draw, keep one of each, until there are enough)
            cand, _o, _p = workload.count_batch_patterns(text, 2 * n_d, m, seed=seed)
            pool = np.concatenate([pool, cand.reshape(-1, m)])
            keys = np.ascontiguousarray(pool).view(np.dtype((np.void, 2 * m))).ravel()
            _u, first = np.unique(keys, return_index=True)
            pool = pool[np.sort(first)]  # first occurrences, in drawing order
            if len(pool) >= n_d:
                break
        seen = min(len(pool), n_d)
        if seen == n_d:
            dpat = np.ascontiguousarray(pool[:n_d]).reshape(-1)
            d_dp = torch.from_numpy(dpat.view(np.int16)).to(dev)
            d_do = torch.from_numpy(np.ascontiguousarray(off)).to(dev)
            d_dc = torch.zeros(n_d, dtype=torch.int32, device=dev)
            d_dl = torch.zeros(n_d, dtype=torch.int32, device=dev)
            d_ds = torch.zeros(n_d, dtype=torch.int32, device=dev)

            def count_distinct(with_lf=False):
                check_rc(ia, ia.lib.fmx_count_batch_dev(q.handle, d_dp.data_ptr(), d_do.data_ptr(), n_d, d_dc.data_ptr(),
                                                        d_dl.data_ptr() if with_lf else None, d_ds.data_ptr() if with_lf else None, sp),
                         "fmx_count_batch_dev")

            ia.lib.fmx_set_option(b"lf_steps_executed_only", 1)
            try:
                count_distinct(True)
                torch.cuda.synchronize()
                lf_exec = int(d_dl.sum(dtype=torch.int64).item())
            finally:
                ia.lib.fmx_set_option(b"lf_steps_executed_only", 0)
            orc.counters_reset()
            oc, ost = ref.count_batch(dpat, off, threads=cores)
            cd = orc.counters()
            if not ((d_dc.cpu().numpy() == oc).all() and int(d_ds.max().item()) == 0):
                raise RuntimeError("count of the distinct batch differs from the oracle")
            ms = timed(count_distinct, 7)
            alg_exec = cd["alg_bytes"] * (lf_exec / float(max(1, cd["lf_steps"])))
            res.append({"config": "configs[1]'s shape with %d DISTINCT patterns (no two alike; the headline batch holds 328,091 distinct ones)" % n_d,
                        "ms": ms, "patterns_per_s": n_d / ms * 1e3, "lf_steps_executed": lf_exec,
                        "roofline": rs.settle_frac({"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                    "achieved": alg_exec / (ms * 1e-3) / 1e9,
                                                    "frac": alg_exec / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                    "kernels": "plan stage + k_count (whole call)",
                                                    "note": "algorithmic bytes of the executed LF-steps (the oracle's bytes per step x the steps "
                                                            "the kernel executed) over the WHOLE call's time, plan stage included"},
                                                   ms, traffic_of("configs[1] distinct", n_d)),
                        "checked_vs_oracle": "all %d counts and statuses" % n_d})
            del d_dp, d_dc, d_dl, d_ds
    except Exception as e:  # noqa: BLE001 - an extra row: its failure is reported on the row
        log("[bench] distinct-pattern row FAILED: %r" % (e,))
        res.append({"config": "configs[1]'s shape with distinct patterns", "ms": None, "error": repr(e)[:300]})
    # ---- the reference's own benchmark shapes on a text with the published data set's alphabet size ----
    # The DEFAULT run carries the published shape's six rows (count, locate 1 / 100, extract 32 at sampleRate 32; count and extract
    # at sampleRate 1: ref_series.DEFAULT_PLAN); --series runs all 18 (sampleRate 1 / 32 / 64 x maxMatches 1 / 10 / 100 / 1000).
    if not args.no_ref_series and not args.profiling:
        name = ("reference_series: FmIndexThroughputBenchmark's count / locate / extract (32 chars), queries of 8..31 chars sampled "
                "from a ~1,100-symbol text (%s)" % ("all 18 rows" if args.series else "the default six rows"))
        try:
            res.append({"config": name,
                        "series": rs.run_series(
                            ia, torch, orc, dev, text_log2=args.text_log2, queries=args.series_queries,
                            plan=rs.full_plan() if args.series else rs.DEFAULT_PLAN,
                            build_device=ctx.local_rank, log=log, traffic_lookup=traffic_of)})
        except Exception as e:  # noqa: BLE001 - a secondary block must not take the headline's line down: the row says what happened
            log("[bench] reference series FAILED: %r" % (e,))
            res.append({"config": name, "ms": None, "error": repr(e)[:300]})
    if args.series_extras:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import series_extras

        res.append({"config": "series_extras: RrrVectorThroughputBenchmark (10 M bits, sampleSize 16/32/64/256, 1 % dense), "
                              "locateAndExtractBenchmark, ingest + serialized size on the 1,099-symbol text",
                    "series": series_extras.run_extras(ia, torch, orc, dev, text_log2=args.text_log2, log=log)})
    return res


# ---------------------------------------------------------------------------------------------------------------
# workload `segments` — BASELINE.json configs[4]
# ---------------------------------------------------------------------------------------------------------------
def run_segments(ctx, args):
    """BASELINE.json configs[4]: every rank holds all segment images, the ONE batch is sharded (strong scaling).
    Returns the JSON line on rank 0 (None elsewhere).  Called as the workload itself (--workload segments) and, at
    N > 1, from the default workload for its `secondary` block, so that one `bench.py --gpus N` yields both figures."""
    from index4j_amd import workload
    from index4j_amd.shard import broadcast_blob, gather_concat, ranks_seen, shard_range

    ia, torch, dist, dev = ctx.ia, ctx.torch, ctx.dist, ctx.dev
    K, m, M, world = args.segments, args.pattern_len, 16, ctx.world
    total = args.patterns_total
    texts = fms = None
    t0 = time.time()
    if ctx.rank == 0:
        texts = workload.segment_texts(K, args.segment_log2)
        fms = [ia.FmIndex(t, args.sample_rate, True, device=None, build_device=None if ctx.dry else ctx.local_rank) for t in texts]
        log("[bench] %d segment indexes (%d chars) built in %.1fs" % (K, sum(len(t) for t in texts), time.time() - t0))
    bases = torch.zeros(K, dtype=torch.int64, device=ctx.cdev)
    if ctx.rank == 0:
        bases.copy_(torch.from_numpy(workload.segment_bases(texts)))
    if dist is not None:
        dist.broadcast(bases, 0)
    segs, bufs, image_bytes = [], [], 0
    t0 = time.time()
    for s in range(K):
        if dist is None:
            if not ctx.dry:
                fms[s].to_device(ctx.local_rank)
            segs.append(fms[s])
            image_bytes += len(fms[s].blob())
        else:
            buf = broadcast_blob(dist, fms[s].blob() if ctx.rank == 0 else None, ctx.cdev, fan_out=getattr(ctx, "fan_out", None))
            if not ctx.dry:
                buf = buf.to(dev)
            bufs.append(buf)
            image_bytes += buf.numel()
            if not ctx.dry:
                segs.append(ia.FmIndex.attach_device_blob(buf.data_ptr(), buf.numel(), ctx.local_rank))
    if ctx.rank == 0 and dist is not None:
        log("[bench] %d segment images (%.2f GB) broadcast and attached on %d rank(s) in %.1fs" % (K, image_bytes / 1e9, world, time.time() - t0))
    sf = None if ctx.dry else ia.SegmentedFmIndex.from_segments(segs, bases.cpu().numpy())
    table_chars = [0] * K if ctx.dry else [f.suffix_table_info()[0] for f in segs]
    table_bytes = 0 if ctx.dry else sum(f.suffix_table_info()[1] for f in segs)
    window_bytes = 0 if ctx.dry else sum(f.window_cells_bytes() for f in segs)
    pat = None
    if ctx.rank == 0:
        pat, _off = workload.segment_patterns(texts, total, m)
    d_pat = hand_out_patterns(ctx, pat, m, total)
    lo, hi = shard_range(total, world, ctx.rank)
    n = hi - lo
    d_off = torch.from_numpy((np.arange(n + 1, dtype=np.int64) * m).astype(np.int32)).to(dev)
    d_cnt = torch.zeros(n, dtype=torch.int64, device=dev)
    d_lf = torch.zeros(n, dtype=torch.int64, device=dev)
    d_st = torch.zeros(n, dtype=torch.int32, device=dev)
    d_tmp = torch.zeros(n * (4 + M), dtype=torch.int32, device=dev)
    d_locs = torch.zeros(n * M, dtype=torch.int64, device=dev)
    d_found = torch.zeros(n, dtype=torch.int32, device=dev)
    stream = None if ctx.dry else torch.cuda.current_stream()
    sp = None if ctx.dry else C.c_void_p(stream.cuda_stream)

    def count_stage(with_lf=False):
        check_rc(ia, ia.lib.fmx_count_segments_dev(sf.handles, K, d_pat.data_ptr(), d_off.data_ptr(), n, d_cnt.data_ptr(),
                                                   d_lf.data_ptr() if with_lf else None, d_st.data_ptr(), d_tmp.data_ptr(), sp),
                 "fmx_count_segments_dev")

    def locate_stage():
        check_rc(ia, ia.lib.fmx_locate_segments_dev(sf.handles, K, sf.base_array.ctypes.data, d_pat.data_ptr(), d_off.data_ptr(), n,
                                                    M, d_locs.data_ptr(), d_found.data_ptr(), d_st.data_ptr(), d_tmp.data_ptr(), sp),
                 "fmx_locate_segments_dev")

    def both_stage(with_lf=False):
        # count() + locate() of the batch in ONE pass over the segments (fmx.h fmx_count_locate_segments_dev): a segment's range
        # search serves both, so the step costs one k_count per segment where the two calls above cost two
        check_rc(ia, ia.lib.fmx_count_locate_segments_dev(sf.handles, K, sf.base_array.ctypes.data, d_pat.data_ptr(), d_off.data_ptr(),
                                                          n, M, d_cnt.data_ptr(), d_lf.data_ptr() if with_lf else None,
                                                          d_locs.data_ptr(), d_found.data_ptr(), d_st.data_ptr(), d_tmp.data_ptr(), sp),
                 "fmx_count_locate_segments_dev")

    def step(with_lf=False):
        if ctx.dry:
            return
        both_stage(with_lf)

    step(True)
    if not ctx.dry:
        torch.cuda.synchronize()
        if int(d_st.max().item()) != 0:
            raise RuntimeError("unexpected per-query status in the benchmark batch")
        # the one-pass call against the two calls it replaces: same counts, LF-steps, hits
        one_pass = (d_cnt.clone(), d_lf.clone(), d_locs.clone(), d_found.clone())
        count_stage(True)
        locate_stage()
        torch.cuda.synchronize()
        live = torch.arange(M, device=dev)[None, :] < d_found[:, None]
        if not (bool((one_pass[0] == d_cnt).all()) and bool((one_pass[1] == d_lf).all()) and bool((one_pass[3] == d_found).all())
                and bool((one_pass[2].view(n, M)[live] == d_locs.view(n, M)[live]).all())):
            raise RuntimeError("fmx_count_locate_segments_dev differs from fmx_count_segments_dev + fmx_locate_segments_dev")
        del one_pass
    lf_local = int(d_lf.sum().item())
    for _ in range(args.warmup):
        step()
    barrier(ctx)
    setup_s = time.time() - getattr(ctx, "t_start", time.time())  # process start -> this workload's timed loop
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier(ctx)
    wall = time.perf_counter() - t0
    # the two stages on their own (this rank's shard), HIP events on the stream the kernels run on
    stage_ms = None
    if not ctx.dry:
        stage_ms = {}
        for name, fn in (("count", count_stage), ("locate", locate_stage), ("count_and_locate_one_pass", both_stage)):
            e0, e1 = hip_events(torch)
            e0.record(stream)
            for _ in range(max(1, args.steps // 4)):
                fn()
            e1.record(stream)
            torch.cuda.synchronize()
            stage_ms[name] = e0.elapsed_time(e1) / max(1, args.steps // 4)
    sums = torch.tensor([int(d_cnt.sum().item()), int(d_found.sum().item()), lf_local], dtype=torch.int64, device=ctx.cdev)
    seen = [[0, ctx.local_rank, ctx.local_rank]]
    head = None
    if dist is not None:
        tw = torch.tensor([wall], dtype=torch.float64, device=ctx.cdev)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())
        dist.all_reduce(sums)
        seen = ranks_seen(dist, ctx.cdev, ctx.local_rank, ctx.local_rank if not ctx.dry else -1)
        sizes = [shard_range(total, world, r)[1] - shard_range(total, world, r)[0] for r in range(world)]
        head = gather_concat(dist, d_cnt, sizes, ctx.cdev, dtype=torch.int64)  # the final gather (counts; hits stay sharded)
    if ctx.rank != 0:
        return None
    if len(seen) != args.gpus or sorted(r[0] for r in seen) != list(range(args.gpus)):
        raise RuntimeError("ranks seen %r do not match --gpus %d" % (seen, args.gpus))
    checked = 0
    roof = base = None
    ms_per_step = wall * 1e3 / args.steps
    if not ctx.dry and not args.no_cpu_baseline:
        # rank 0's oracle sample: the first patterns of the batch against the K oracle indexes — parity of the run, the
        # algorithmic bytes / LF-steps of both stages (counting mode), and the CPU figure
        orc = oracle_module()
        cores = os.cpu_count() or 1
        k = min(n, args.segments_check)
        cnt = d_cnt[:k].cpu().numpy()
        locs = d_locs[: k * M].cpu().numpy().reshape(k, M)
        found = d_found[:k].cpu().numpy()
        off_k = (np.arange(k + 1, dtype=np.int64) * m).astype(np.int32)
        exp_c = np.zeros(k, np.int64)
        exp_l = np.zeros((k, M), np.int64)
        exp_f = np.zeros(k, np.int32)
        bs = bases.cpu().numpy()
        alg = {"count": 0, "locate": 0}
        steps_ref = {"count": 0, "locate": 0}
        table_alg = table_steps = 0
        k1 = min(k, 2000)  # single-thread sample of the CPU figure
        cpu_s = 0.0
        for s in range(K):
            o = orc.OracleFmIndex.read(fms[s].write(False))
            orc.counters_reset()
            oc, _ = o.count_batch(pat[: k * m], off_k, threads=cores)
            c = orc.counters()
            alg["count"] += c["alg_bytes"]
            steps_ref["count"] += c["lf_steps"]
            exp_c += oc
            if table_chars[s] and m >= table_chars[s]:  # what this segment's suffix table answers (both stages' backward search)
                tc = table_chars[s]
                tail = np.ascontiguousarray(pat[: k * m].reshape(k, m)[:, m - tc:]).reshape(-1)
                orc.counters_reset()
                o.count_batch(tail, (np.arange(k + 1, dtype=np.int64) * tc).astype(np.int32), threads=cores)
                ct = orc.counters()
                table_alg += ct["alg_bytes"]
                table_steps += ct["lf_steps"]
            # the caller's loop `n += segment.locate(p, 0, len, locations, maxMatches - n)`: segment s is asked for the hits
            # still missing.  Grouped by that limit so that the counters are those of exactly the work asked for; a pattern
            # that already has all its hits still goes through this segment's backward search on the GPU (k_count runs
            # for the whole batch), which is counted as executed work
            remaining = M - exp_f
            rows = pat[: k * m].reshape(k, m)
            orc.counters_reset()
            for r in np.unique(remaining):
                sel = np.flatnonzero(remaining == r)
                sub = np.ascontiguousarray(rows[sel]).reshape(-1)
                off_sub = (np.arange(len(sel) + 1, dtype=np.int64) * m).astype(np.int32)
                if r <= 0:
                    o.count_batch(sub, off_sub, threads=cores)
                    continue
                ol, of, _ = o.locate_batch(sub, off_sub, int(r), threads=cores)
                for j in range(int(r)):
                    hit = np.flatnonzero(of > j)
                    exp_l[sel[hit], exp_f[sel[hit]]] = ol[hit, j].astype(np.int64) + int(bs[s])
                    exp_f[sel[hit]] += 1
            c = orc.counters()
            alg["locate"] += c["alg_bytes"]
            steps_ref["locate"] += c["lf_steps"]
            t1 = time.perf_counter()
            o.count_batch(pat[: k1 * m], off_k[: k1 + 1], threads=1)
            o.locate_batch(pat[: k1 * m], off_k[: k1 + 1], M, threads=1)
            cpu_s += time.perf_counter() - t1
            del o
        live = np.arange(M)[None, :] < exp_f[:, None]
        if not ((cnt == exp_c).all() and (found == exp_f).all() and (locs[live] == exp_l[live]).all()):
            raise RuntimeError("segment-set results differ from the oracle sample")
        if head is not None and not (head[:k] == exp_c).all():
            raise RuntimeError("gathered counts differ from the oracle sample")
        checked = k
        scale = n / float(k)  # this rank's shard (what the stage times below cover)
        # The step is the ONE-PASS call: each segment's range search runs once and serves count() and locate() alike, so the
        # bytes it executes are those of locate() (search + walks); the two-call form executes the searches twice.
        exec_alg = {"count": (alg["count"] - table_alg) * scale, "locate": (alg["locate"] - table_alg) * scale}
        exec_alg["count_and_locate_one_pass"] = exec_alg["locate"]
        dom = "count_and_locate_one_pass"
        whole = exec_alg[dom] * world  # every rank's shard, per step
        roof = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "achieved": exec_alg[dom] / (stage_ms[dom] * 1e-3) / 1e9, "frac": exec_alg[dom] / (stage_ms[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "traffic": None, "traffic_note": "no PMC pass of this workload",
                "kernel": "one shard, one pass: %d x (k_count + k_segment_add_counts + walk order + k_locate_walk + "
                          "k_segment_append_hits)" % K,
                "stage_ms_this_rank": stage_ms,
                "frac_per_stage": {st: exec_alg[st] / (stage_ms[st] * 1e-3) / 1e9 / HBM_PEAK_GBS for st in stage_ms},
                "frac_whole_step": whole / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS / world,
                "what_frac_means": "algorithmic bytes (oracle counting mode on the %d checked patterns x %d segments, scaled to the "
                                   "shard, minus what the segments' suffix tables answer) of the one-pass call — the range searches "
                                   "once, the walks — / its HIP-event time / peak; stage_ms also holds the two separate calls "
                                   "(fmx_count_segments_dev, fmx_locate_segments_dev), which search every segment twice; "
                                   "frac_whole_step: the same bytes over the contract's step time, per GPU" % (k, K),
                "alg_bytes_executed_per_step_per_gpu": exec_alg[dom],
                "lf_steps_reference_per_pattern": {st: steps_ref[st] / float(k) for st in steps_ref},
                "lf_steps_answered_by_tables_per_pattern": table_steps / float(k),
                "image_bytes_per_text_byte": image_bytes / float(sum(len(t) for t in texts)),
                "window_directory_bytes_per_text_byte": window_bytes / float(sum(len(t) for t in texts)),
                "resident_bytes_per_text_byte": (image_bytes + table_bytes + window_bytes) / float(sum(len(t) for t in texts))}
        base = {"value": k1 / cpu_s, "unit": "patterns/s", "cores": 1, "kind": "port",
                "sample": "count() + locate(maxMatches %d) of the first %d patterns over the %d oracle indexes "
                          "(oracle/index4j_oracle.c, C port of index4j's path), 1 thread, %.1f s" % (M, k1, K, cpu_s)}
    out = {
        "metric": "patterns/sec, count()+locate() of one 8M x 8-char batch over a 2 GiB log text as 8 segment indexes",
        "step_is": "ONE call per step, fmx_count_locate_segments_dev: counts and located hits of the batch from one range search per "
                   "segment (checked against the two separate calls in this run); roofline.stage_ms_this_rank has both forms",
        "value": None if ctx.dry else total * args.steps / wall,
        "unit": "patterns/s",
        "lf_steps_per_sec_count_stage": None if ctx.dry else int(sums[2].item()) * args.steps / wall,
        "n_gpus": world,
        "ranks_seen": seen,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "setup_s": setup_s,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "int32",
        "data": "synthetic",
        "config": {"workload": "count() + locate(maxMatches %d) of ONE batch of %d random %d-char patterns over %d segment "
                               "indexes of <= 2^%d chars (sampleRate %d), images broadcast to every GPU, batch sharded x%d "
                               "(BASELINE.json configs[4])" % (M, total, m, K, args.segment_log2, args.sample_rate, world),
                   "segments": K, "patterns_total": total, "pattern_len": m, "max_matches": M, "image_bytes_per_gpu": image_bytes,
                   "suffix_table_chars_per_segment": table_chars, "suffix_table_bytes_per_gpu": table_bytes,
                   "count_checksum_all_ranks": int(sums[0].item()), "hits_all_ranks": int(sums[1].item()),
                   "patterns_checked_vs_oracle": checked},
        "roofline": roof,
        "cpu_baseline": base,
    }
    if ctx.dry:
        out["dry_run"] = True
    if getattr(ctx, "shared", False):
        out["rehearsal"] = "N ranks sharing ONE GPU, collectives over gloo on host tensors: checks the N > 1 code path end to end, measures nothing"
    for f in segs:
        if not ctx.dry and dist is not None:
            f.close()
    return out


def main():
    # OpenMP teams (the oracle's, torch's) park instead of spinning after a parallel region: a spinning team on every core
    # delays this thread's kernel launches, and HIP-event pairs span those gaps (must be set before libgomp is loaded)
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", choices=["count", "segments"], default="count")
    ap.add_argument("--text-log2", type=int, default=28, help="log2 of the text length in chars (28 = 256 MiB)")
    ap.add_argument("--patterns", type=int, default=1 << 20, help="patterns per GPU and batch (workload count)")
    ap.add_argument("--batches", type=int, default=4, help="distinct batches rotating through the timed loop")
    ap.add_argument("--pattern-len", type=int, default=8)
    ap.add_argument("--sample-rate", type=int, default=32)
    ap.add_argument("--segments", type=int, default=8)
    ap.add_argument("--segment-log2", type=int, default=28)
    ap.add_argument("--patterns-total", type=int, default=1 << 23, help="patterns of the one batch (workload segments)")
    ap.add_argument("--segments-check", type=int, default=20000, help="patterns of rank 0's shard checked against 8 oracle indexes")
    ap.add_argument("--cpu-budget", type=float, default=10.0, help="seconds of single-thread oracle time for cpu_baseline")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip every oracle leg (profiling runs)")
    ap.add_argument("--no-secondary", action="store_true", help="skip configs[2] / [3] and the reference-shaped series")
    ap.add_argument("--series", action="store_true",
                    help="also run the reference-shaped series (count / locate 1..1000 / extract-32 at sampleRate 1, 32, 64 on the "
                         "1,100-symbol text; ~100 s, most of it the oracle's check); off by default so that the default run stays short")
    ap.add_argument("--no-ref-series", action="store_true", help="skip the reference-shaped series (six rows in the default run)")
    ap.add_argument("--no-segments-share", action="store_true",
                    help="skip the configs[4] per-GPU share at N = 1 (the one row of the line whose image set does not fit the Infinity Cache)")
    ap.add_argument("--series-extras", action="store_true",
                    help="also run BASELINE.md's remaining rows (tools/series_extras.py): stand-alone RrrVector.rankOnes at 10 M bits, "
                         "locateAndExtract, ingest time and serialized size on the 1,099-symbol text; each oracle-checked (~60 s)")
    ap.add_argument("--series-queries", type=int, default=1 << 20, help="queries per batch of the reference-shaped series")
    ap.add_argument("--profiling", action="store_true",
                    help="rocprofv3 runs: skip the extra legs that launch the headline kernels in other modes (without the suffix "
                         "table, two batches in flight), so that per-kernel averages and counters describe the timed path only")
    ap.add_argument("--overlap-streams", type=int, default=2,
                    help="streams of the extra `overlapped` measurement (batches in flight); 1 = skip it")
    ap.add_argument("--image-compact", action="store_true",
                    help="run the whole line over COMPACT images (library option image_compact: bit vectors as RRR records, value table "
                         "in LDS; smaller, slower) — the line says so in config.image")
    ap.add_argument("--share-one-gpu", action="store_true",
                    help="REHEARSAL of the N > 1 code path on a one-GPU box: every rank queries on cuda:0, collectives run "
                         "over gloo on host tensors.  The line says so (`rehearsal`); it is not a measurement.")
    ap.add_argument("--single-process", action="store_true",
                    help="ONE host process drives all --gpus N devices through the C ABI's replica calls (fmx_replicate + "
                         "fmx_count_batch_multi_dev) instead of one process per GPU over torch.distributed: the form a Java host uses")
    ap.add_argument("--no-single-process-leg", action="store_true",
                    help="at N > 1 (launcher form): skip rank 0's extra measurement of the single-process form over the same GPUs")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU-only rehearsal of the launch / broadcast / shard / gather plumbing over gloo: no queries, no numbers")
    ap.add_argument("--cache-dir", default=os.environ.get("FMX_CACHE", "/tmp/fmx_cache"))
    ap.add_argument("--time-budget", type=float, default=540.0,
                    help="seconds the whole command may take (the driver's limit is 600): at N > 1 the configs[4] block is skipped, "
                         "and the line says so, once more than half of it is gone")
    args = ap.parse_args()
    t_start = time.time()

    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if args.single_process and not launched:
        run_single_process(args, t_start)
        return
    if args.gpus > 1 and not launched:
        sys.exit(self_launch(args, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log("[bench] WORLD_SIZE=%d but --gpus %d: refusing to run a mislabelled benchmark" % (world, args.gpus))
        sys.exit(2)

    import torch  # before the package: whichever HIP runtime is loaded first serves both (tests/conftest.py)

    import index4j_amd as ia

    if args.image_compact:
        check_rc(ia, ia.lib.fmx_set_option(b"image_compact", 1), "fmx_set_option")
    ctx = Ctx()
    ctx.t_start = t_start
    ctx.ia, ctx.torch, ctx.dry = ia, torch, args.dry_run
    ctx.world, ctx.rank = world, int(os.environ.get("RANK", "0"))
    ctx.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not ctx.dry:
        if not torch.cuda.is_available() or ia.lib.fmx_device_count() < 1:
            raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
        if ctx.local_rank >= torch.cuda.device_count():  # launcher that gives every rank its own visible-device mask
            ctx.local_rank = 0
        if args.share_one_gpu:
            ctx.local_rank = 0
        torch.cuda.set_device(ctx.local_rank)
        ctx.dev = torch.device("cuda", ctx.local_rank)
    else:
        ctx.dev = torch.device("cpu")
    # where collectives' tensors live: the device (RCCL), or the host in the rehearsal forms (gloo)
    ctx.cdev = torch.device("cpu") if (ctx.dry or args.share_one_gpu) else ctx.dev
    ctx.shared = bool(args.share_one_gpu)
    ctx.dist = None
    if launched:  # also at WORLD_SIZE 1: the same RCCL code path (broadcast, scatter, attach, gather) as at 8
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if ctx.dry or args.share_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=ctx.dev)
        ctx.dist = dist
    try:
        out = run_count(ctx, args) if args.workload == "count" else run_segments(ctx, args)
        if ctx.rank == 0:
            emit(out)
    finally:
        if ctx.dist is not None:
            ctx.dist.destroy_process_group()


if __name__ == "__main__":
    main()
