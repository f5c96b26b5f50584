"""The N > 1 form of bench.py with real kernels on ONE GPU: two ranks share cuda:0, collectives run over gloo on host
tensors (`--share-one-gpu`).  Checks what the CPU rehearsal (tests/test_dist_gloo.py, --dry-run) cannot: image broadcast ->
fmx_attach_device_blob, pattern shards, every rank's kernels, max-over-ranks timing, the gathered counts, and the
configs[4] line a multi-GPU launch carries in `secondary`.  The numbers of such a run mean nothing and are not looked at."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_run_the_whole_multi_gpu_path():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-one-gpu", "--steps", "4", "--warmup", "1",
           "--cpu-budget", "0.5", "--text-log2", "24", "--segment-log2", "22", "--patterns-total", "1048576",
           "--segments-check", "2000"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-4000:]
    last = r.stdout.strip().splitlines()[-1]
    contract = json.loads(last)  # what the driver parses: the compact line, last, below 4 KB
    assert len(last) < 4096 and contract["n_gpus"] == 2 and contract["value"] > 0 and "roofline" in contract
    assert [s for s in contract["secondary"] if s["frac"]], contract["secondary"]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("BENCH_DETAIL ")][-1][len("BENCH_DETAIL "):])
    assert line["n_gpus"] == 2 and sorted(x[0] for x in line["ranks_seen"]) == [0, 1] and "rehearsal" in line
    assert line["scaling"] == "weak" and line["value"] > 0
    seg = [s for s in line["secondary"] if s.get("scaling") == "strong"]
    assert len(seg) == 1 and seg[0]["n_gpus"] == 2 and seg[0]["roofline"] and seg[0]["cpu_baseline"]
    assert seg[0]["config"]["patterns_checked_vs_oracle"] > 0
    # rank 0 also ran the single-process form (the C ABI's replica calls) over the same two "GPUs": same batch, same checksum
    sp = line["single_process"]
    assert sp and "error" not in sp and "skipped" not in sp, sp
    assert sp["devices"] == [0, 0] and sp["ms_per_step"] > 0 and sp["count_checksum_all_shards"] == line["config"]["gathered_checksum_all_ranks"]
    assert contract["single_process"]["ms_per_step"] > 0


@pytest.mark.gpu
def test_single_process_form_drives_three_replicas_through_the_c_abi():
    """`bench.py --gpus 3 --single-process`: ONE host process, fmx_replicate + fmx_count_batch_multi_dev (three replicas sharing the
    one GPU of the box), every pattern of batch 0 checked against the oracle, the contract line of the launcher form"""
    env = dict(os.environ)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--single-process", "--share-one-gpu", "--steps", "4",
           "--warmup", "1", "--text-log2", "24", "--patterns", "300001", "--batches", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-4000:]
    last = r.stdout.strip().splitlines()[-1]
    contract = json.loads(last)
    assert len(last) < 4096 and contract["n_gpus"] == 3 and contract["value"] > 0 and contract["scaling"] == "weak"
    assert contract["ranks_seen"] == [[0, 0, 0], [1, 1, 0], [2, 2, 0]] and "single-process" in contract["launch"]
    assert contract["config"]["patterns_checked_vs_oracle"] == 3 * 300001 and contract["config"]["parallelism"] == "dp3"
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("BENCH_DETAIL ")][-1][len("BENCH_DETAIL "):])
    assert "rehearsal" in line and line["host_buffers"]["ms_per_call"] > 0 and line["resident_bytes_per_replica"]["image"] > 0
