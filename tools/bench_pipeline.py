#!/usr/bin/env python3
"""The reference's locateAndExtractBenchmark (FmIndexThroughputBenchmark.java:231-249 with the defaults of
FmIndexThroughputState.java:28-115: sampleRate 32, 20,000 queries of 8..31 chars sampled from the text,
maxMatches 1000, maxExtractionLength 64) as one fused device pipeline on the 256 MiB index, plus the
"grep" form (locate -> extractUntilBoundary('\\n')).  Operands resident in HBM, HIP-event timing, a sample
checked bit-exactly against the oracle, and the oracle timed on the same sample as the CPU figure.
usage: python tools/bench_pipeline.py [--text-log2 28] [--queries 20000] [--out gpurun_out/pipeline.jsonl]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--text-log2", type=int, default=28)
    ap.add_argument("--queries", type=int, default=20000)
    ap.add_argument("--max-matches", type=int, default=1000)
    ap.add_argument("--extract-len", type=int, default=64)
    ap.add_argument("--line-cap", type=int, default=512)
    ap.add_argument("--check", type=int, default=150)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "pipeline.jsonl"))
    args = ap.parse_args()
    import torch

    import bench
    import index4j_amd as ia
    import orc
    from bench_configs import timed

    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream()
    sp = C.c_void_p(stream.cuda_stream)
    text, fm, path = bench.build_or_load_index(ia, args.text_log2, 32, "/tmp/fmx_cache")
    fm.to_device(0)
    o = orc.OracleFmIndex.read(open(path, "rb").read())
    inlen = fm.getInputLength()
    n, mm = args.queries, args.max_matches
    rng = np.random.default_rng(42)
    lens = rng.integers(8, 32, n)
    starts = rng.integers(0, len(text) - 32, n)
    off = np.zeros(n + 1, np.int32)
    off[1:] = np.cumsum(lens)
    pat = np.concatenate([text[s:s + l] for s, l in zip(starts, lens)]).astype(np.uint16)
    d_pat = torch.from_numpy(pat.view(np.int16)).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    slots = n * mm
    d_locs = torch.zeros(slots, dtype=torch.int32, device=dev)
    d_found = torch.zeros(n, dtype=torch.int32, device=dev)
    d_len = torch.zeros(slots, dtype=torch.int32, device=dev)
    d_hst = torch.zeros(slots, dtype=torch.int32, device=dev)
    d_aux = torch.zeros(slots, dtype=torch.int32, device=dev)
    d_lf = torch.zeros(n, dtype=torch.int32, device=dev)
    d_st = torch.zeros(n, dtype=torch.int32, device=dev)
    d_ws = torch.zeros(2 * n, dtype=torch.int32, device=dev)
    out = []

    # ---- locate -> extract(loc, min(inputLength, loc + 64)) ----
    xl = args.extract_len
    d_dst = torch.zeros(slots * xl, dtype=torch.int16, device=dev)

    def run_extract():
        rc = ia.lib.fmx_locate_extract_batch_dev(fm.handle, d_pat.data_ptr(), d_off.data_ptr(), n, mm, xl, d_locs.data_ptr(),
                                                 d_found.data_ptr(), d_dst.data_ptr(), d_len.data_ptr(), d_lf.data_ptr(),
                                                 d_st.data_ptr(), d_hst.data_ptr(), d_ws.data_ptr(), sp)
        assert rc == 0, ia.lib.fmx_last_error()

    run_extract()
    torch.cuda.synchronize()
    found = d_found.cpu().numpy()
    locs = d_locs.cpu().numpy().reshape(n, mm)
    hst = d_hst.cpu().numpy().reshape(n, mm)
    olen = d_len.cpu().numpy().reshape(n, mm)
    hits = int(found.sum())
    assert int(d_st.max().item()) == 0
    t0 = time.perf_counter()
    cpu_hits = 0
    rows = d_dst.view(slots, xl)
    for i in range(args.check):
        k, l = o.locate(pat[off[i]:off[i + 1]], max_matches=mm, cap=mm)
        assert k == found[i] and (l == locs[i, :k]).all(), i
        got = rows[i * mm:i * mm + k].cpu().numpy().view(np.uint16)
        for j in range(k):
            stop = min(inlen, int(l[j]) + xl)
            if stop >= inlen:
                assert hst[i, j] == 3  # "Stop position longer than index string" FM:572-574
                continue
            m, d = o.extract(int(l[j]), stop, dest_len=xl)
            assert hst[i, j] == 0 and olen[i, j] == m and (d == got[j]).all(), (i, j)
        cpu_hits += k
    cpu_s = time.perf_counter() - t0
    ms = timed(run_extract, stream, 3)
    out.append({"config": "locateAndExtractBenchmark: %d queries of 8..31 chars, maxMatches %d, extract %d chars, 2^%d text, sampleRate 32"
                          % (n, mm, xl, args.text_log2),
                "ms": ms, "queries": n, "hits": hits, "chars_extracted": int(olen[olen > 0].sum()),
                "queries_per_s": n / ms * 1e3, "hits_per_s": hits / ms * 1e3,
                "cpu_oracle_1core": {"queries": args.check, "hits": cpu_hits, "seconds": cpu_s,
                                     "hits_per_s": cpu_hits / cpu_s, "note": "includes the Python per-hit call overhead"},
                "checked_vs_oracle_queries": args.check})
    print(json.dumps(out[-1]), flush=True)
    del d_dst, rows

    # ---- grep: locate -> extractUntilBoundary('\n') ----
    cap = args.line_cap
    gm = min(mm, 100)
    gslots = n * gm
    d_dst = torch.zeros(gslots * cap, dtype=torch.int16, device=dev)

    def run_lines():
        rc = ia.lib.fmx_locate_lines_batch_dev(fm.handle, d_pat.data_ptr(), d_off.data_ptr(), n, gm, 10, 0, cap,
                                               d_locs.data_ptr(), d_found.data_ptr(), d_dst.data_ptr(), d_len.data_ptr(),
                                               d_lf.data_ptr(), d_st.data_ptr(), d_hst.data_ptr(), d_aux.data_ptr(),
                                               d_ws.data_ptr(), sp)
        assert rc == 0, ia.lib.fmx_last_error()

    run_lines()
    torch.cuda.synchronize()
    found = d_found.cpu().numpy()
    locs = d_locs.cpu().numpy()[:gslots].reshape(n, gm)
    hst = d_hst.cpu().numpy()[:gslots].reshape(n, gm)
    olen = d_len.cpu().numpy()[:gslots].reshape(n, gm)
    rows = d_dst.view(gslots, cap)
    hits = int(found.sum())
    for i in range(min(args.check, 60)):
        k = int(found[i])
        got = rows[i * gm:i * gm + k].cpu().numpy().view(np.uint16)
        for j in range(k):
            m, d = o.extract_until_boundary(0, int(locs[i, j]), cap, 0, "\n")
            assert hst[i, j] == 0 and olen[i, j] == m and (d == got[j]).all(), (i, j)
    valid = np.arange(gm)[None, :] < found[:, None]
    ms = timed(run_lines, stream, 3)
    out.append({"config": "grep: locate -> extractUntilBoundary('\\n'), %d queries of 8..31 chars, maxMatches %d, line cap %d, 2^%d text, sampleRate 32"
                          % (n, gm, cap, args.text_log2),
                "ms": ms, "queries": n, "hits": hits, "chars_extracted": int(olen[valid].sum()),
                "queries_per_s": n / ms * 1e3, "lines_per_s": hits / ms * 1e3,
                "chars_per_s": int(olen[valid].sum()) / ms * 1e3, "checked_vs_oracle_queries": min(args.check, 60)})
    print(json.dumps(out[-1]), flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        for r in out:
            f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
