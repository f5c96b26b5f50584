#!/usr/bin/env python3
"""ONE row of bench.py's line per process, launched a few times and nothing else — what tools/profile_rows.sh puts under
`rocprofv3 --pmc` so that every dispatch of the query kernels in the process belongs to that row (the counter CSVs carry no
row markers; index construction and the suffix table use other kernels, filtered out by name in tools/summarize_rows.py).
No oracle here: the same calls are checked in bench.py's own run.

    python3 tools/pmc_rows.py --row "configs[3]" [--calls 3]
    python3 tools/pmc_rows.py --list

Row keys = what bench.py's `secondary` rows and profiles/pmc_latest.json `rows` are called (tools/ref_series.py row_key)."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def rows():
    import ref_series as rs

    out = ["configs[2]", "configs[3]", "configs[4] share"]
    for s, wanted in rs.DEFAULT_PLAN:
        out += [rs.row_key(b, mm, s) for b, mm in wanted]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--row")
    ap.add_argument("--calls", type=int, default=3)
    ap.add_argument("--list", action="store_true")
    ap.add_argument("--prepare", action="store_true",
                    help="configs[3]: locate the hit positions and keep them under --cache-dir (run once, NOT under the profiler: "
                         "the locate's own kernels would be counted into the row)")
    ap.add_argument("--text-log2", type=int, default=28)
    ap.add_argument("--series-queries", type=int, default=1 << 20)
    ap.add_argument("--cache-dir", default=os.environ.get("FMX_CACHE", "/tmp/fmx_cache"))
    args = ap.parse_args()
    if args.list:
        print("\n".join(rows()))
        return
    if args.row not in rows():
        raise SystemExit("unknown row %r (see --list)" % args.row)
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    import torch

    import bench
    import index4j_amd as ia
    import ref_series as rs
    from index4j_amd import workload

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    stream = torch.cuda.current_stream()
    sp = C.c_void_p(stream.cuda_stream)

    def check(rc, what):
        if rc != 0:
            raise RuntimeError("%s failed: %s" % (what, ia.lib.fmx_last_error().decode()))

    def t32(a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev)

    if args.row.startswith("series "):
        # "series <bench>[(mm)] s=<rate>"
        name, rate = args.row[len("series "):].rsplit(" s=", 1)
        mm = None
        if "(" in name:
            name, mm = name[:-1].split("(")
            mm = int(mm)
        rs.run_series(ia, torch, None, dev, text_log2=args.text_log2, queries=args.series_queries, plan=((int(rate), ((name, mm),)),),
                      check=False, calls=args.calls, log=lambda *a: print(*a, file=sys.stderr, flush=True))
        print("PMC_ROW_CALLS %d QUERIES %d" % (args.calls, rs.series_queries(args.series_queries, int(rate), name, mm or 0)))
        return
    if args.row == "configs[4] share":
        ctx = bench.Ctx()
        ctx.ia, ctx.torch, ctx.dry, ctx.world, ctx.rank, ctx.local_rank = ia, torch, False, 1, 0, 0
        ctx.dev = ctx.cdev = dev
        ctx.dist, ctx.shared = None, False
        ctx.t_start = __import__('time').time()
        seg = argparse.Namespace(segments=8, pattern_len=8, patterns_total=bench.SHARE_PATTERNS, segment_log2=args.text_log2,
                                 sample_rate=32, steps=args.calls, warmup=0, gpus=1, no_cpu_baseline=True, segments_check=0)
        bench.run_segments(ctx, seg)
        # (run_segments: one step with LF-step outputs, `steps` timed steps, each stage once more for the stage times)
        print("PMC_ROW_CALLS %d QUERIES %d" % (args.calls + 2, bench.SHARE_PATTERNS))
        return
    # configs[2] / [3]: bench.py run_secondary's operands
    K, M, m = 100_000, 16, 8
    text, fm32, _path = bench.build_or_load_index(ia, args.text_log2, 32, args.cache_dir)
    pat, off, _pos = workload.count_batch_patterns(text, K, m, seed=workload.PATTERN_SEED)
    d_pat = torch.from_numpy(np.ascontiguousarray(pat).view(np.int16)).to(dev)
    d_off = t32(off)
    d_locs = torch.zeros(K * M, dtype=torch.int32, device=dev)
    d_found = torch.zeros(K, dtype=torch.int32, device=dev)
    d_st = torch.zeros(K, dtype=torch.int32, device=dev)
    d_rng = torch.zeros(2 * K, dtype=torch.int32, device=dev)

    def locate(index):
        check(ia.lib.fmx_locate_batch_dev(index.handle, d_pat.data_ptr(), d_off.data_ptr(), K, M, d_locs.data_ptr(), M,
                                          d_found.data_ptr(), None, d_st.data_ptr(), d_rng.data_ptr(), sp), "fmx_locate_batch_dev")

    if args.row == "configs[2]":
        fm32.to_device(0)
        for _ in range(args.calls):
            locate(fm32)
        torch.cuda.synchronize()
        print("PMC_ROW_CALLS %d QUERIES %d" % (args.calls, K))
        return
    del fm32
    _t, fm64, _p = bench.build_or_load_index(ia, args.text_log2, 64, args.cache_dir)
    fm64.to_device(0)
    froms_file = os.path.join(args.cache_dir, "pmc_rows_froms_%d.npy" % args.text_log2)
    if args.prepare:
        locate(fm64)
        torch.cuda.synchronize()
        np.save(froms_file, np.ascontiguousarray(d_locs.cpu().numpy().reshape(K, M)[:, 0]))
        return
    froms = np.load(froms_file)
    cap = 1024
    d_from = t32(froms)
    d_dst = torch.zeros(K * cap, dtype=torch.int16, device=dev)
    d_len = torch.zeros(K, dtype=torch.int32, device=dev)
    d_aux = torch.zeros(K, dtype=torch.int32, device=dev)
    for _ in range(args.calls):
        check(ia.lib.fmx_extract_boundary_batch_dev(fm64.handle, d_from.data_ptr(), K, 10, 0, d_dst.data_ptr(), cap, 0, d_len.data_ptr(),
                                                    None, d_st.data_ptr(), d_aux.data_ptr(), sp), "fmx_extract_boundary_batch_dev")
    torch.cuda.synchronize()
    print("PMC_ROW_CALLS %d QUERIES %d" % (args.calls, K))


if __name__ == "__main__":
    main()
