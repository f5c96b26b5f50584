#!/usr/bin/env python3
"""BASELINE configs[4] on ONE MI355X: a 2 GiB text as 8 segment indexes of <= 2^28 chars (cut at '\\n'; a Java
int cannot address 2^31 chars), all resident on the GPU (scheme (i) of SURVEY §8e), and the
per-GPU share of the 8M-pattern batch (1,048,576 patterns of 8 chars): count() summed over the segments and
locate() with base-shifted hits.  HIP-event timing with operands in HBM, a sample checked against 8 oracle
indexes.  usage: python tools/bench_segments.py [--segments 8] [--segment-log2 28] [--patterns 1048576]"""
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
    ap.add_argument("--segments", type=int, default=8)
    ap.add_argument("--segment-log2", type=int, default=28)
    ap.add_argument("--patterns", type=int, default=1 << 20)
    ap.add_argument("--max-matches", type=int, default=16)
    ap.add_argument("--check", type=int, default=300)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "segments.jsonl"))
    args = ap.parse_args()
    import torch

    import index4j_amd as ia
    import orc
    from bench_configs import timed

    dev = torch.device("cuda", 0)
    K, n_seg = args.segments, 1 << args.segment_log2
    texts, fms = [None] * K, [None] * K

    def build(s):
        t = ia.synth_log(n_seg, seed=42 + s)
        t = t[: int(np.flatnonzero(t == 10)[-1]) + 1]  # the piece ends with its last complete line
        texts[s] = t
        fms[s] = ia.FmIndex(t, 32, True, device=None, build_device=0)  # suffix-array stage on the GPU

    t0 = time.time()
    for s in range(K):
        build(s)
    build_s = time.time() - t0
    bases = np.concatenate([[0], np.cumsum([len(t) for t in texts])[:-1]]).astype(np.int64)
    total_chars = int(sum(len(t) for t in texts))
    for f in fms:
        f.to_device(0)
    sf = ia.SegmentedFmIndex.from_segments(fms, bases)
    blob_bytes = sum(f.device_blob()[1] for f in fms)
    print("[segments] %d segments, %d chars, built in %.1fs, %.2f GB of index in HBM" % (K, total_chars, build_s, blob_bytes / 1e9),
          file=sys.stderr, flush=True)

    n, m = args.patterns, 8
    rng = np.random.default_rng(43)
    seg_of = rng.integers(0, K, n)
    pat = np.empty((n, m), np.uint16)
    for s in range(K):
        sel = np.flatnonzero(seg_of == s)
        p = rng.integers(0, len(texts[s]) - m, len(sel))
        pat[sel] = texts[s][p[:, None] + np.arange(m)[None, :]]
    off = (np.arange(n + 1) * m).astype(np.int32)
    stream = torch.cuda.current_stream()
    sp = C.c_void_p(stream.cuda_stream)
    d_pat = torch.from_numpy(pat.reshape(-1).view(np.int16)).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    d_cnt = torch.zeros(n, dtype=torch.int64, device=dev)
    d_lf = torch.zeros(n, dtype=torch.int64, device=dev)
    d_st = torch.zeros(n, dtype=torch.int32, device=dev)
    M = args.max_matches
    d_tmp = torch.zeros(n * (4 + M), dtype=torch.int32, device=dev)
    d_locs = torch.zeros(n * M, dtype=torch.int64, device=dev)
    d_found = torch.zeros(n, dtype=torch.int32, device=dev)

    def count(with_lf=True):
        rc = ia.lib.fmx_count_segments_dev(sf.handles, K, d_pat.data_ptr(), d_off.data_ptr(), n, d_cnt.data_ptr(),
                                           d_lf.data_ptr() if with_lf else None, d_st.data_ptr(), d_tmp.data_ptr(), sp)
        assert rc == 0, ia.lib.fmx_last_error()

    def locate():
        rc = ia.lib.fmx_locate_segments_dev(sf.handles, K, sf.base_array.ctypes.data, d_pat.data_ptr(), d_off.data_ptr(), n, M,
                                            d_locs.data_ptr(), d_found.data_ptr(), d_st.data_ptr(), d_tmp.data_ptr(), sp)
        assert rc == 0, ia.lib.fmx_last_error()

    count()
    torch.cuda.synchronize()
    cnt = d_cnt.cpu().numpy()
    lf_total = int(d_lf.sum().item())
    assert int(d_st.max().item()) == 0 and (cnt >= 1).all()
    locate()
    torch.cuda.synchronize()
    found = d_found.cpu().numpy()
    locs = d_locs.cpu().numpy().reshape(n, M)
    assert (found == np.minimum(cnt, M)).all()
    # every located position holds the pattern (whole batch), and a sample equals the oracle bit for bit
    seg_idx = np.searchsorted(bases, locs, side="right") - 1
    for k in range(M):
        sel = np.flatnonzero(found > k)
        for s in range(K):
            ss = sel[seg_idx[sel, k] == s]
            loc = locs[ss, k] - bases[s]
            assert (texts[s][loc[:, None] + np.arange(m)[None, :]] == pat[ss]).all()
    oracles = [orc.OracleFmIndex.read(f.write(False)) for f in fms]
    for i in range(args.check):
        exp_c, exp_l = 0, []
        for o, b in zip(oracles, bases):
            exp_c += o.count(pat[i])
            k, l = o.locate(pat[i], max_matches=M, cap=M)
            exp_l.extend(int(x) + int(b) for x in l)
        assert cnt[i] == exp_c and list(locs[i, :found[i]]) == exp_l[:M], i
    del oracles
    ms_c = timed(count, stream, 10)
    ms_l = timed(locate, stream, 5)
    out = [{"config": "configs[4] per-GPU share: count() of %d x %d-char patterns over %d segments (%d chars, %.2f GB index in HBM), sampleRate 32"
                      % (n, m, K, total_chars, blob_bytes / 1e9),
            "ms": ms_c, "patterns_per_s": n / ms_c * 1e3, "lf_steps": lf_total, "lf_steps_per_s": lf_total / ms_c * 1e3,
            "build_seconds_all_segments": build_s, "checked_vs_oracle": args.check},
           {"config": "configs[4] per-GPU share: locate() maxMatches %d, same batch and segments" % M, "ms": ms_l,
            "patterns_per_s": n / ms_l * 1e3, "hits": int(found.sum()), "hits_per_s": int(found.sum()) / ms_l * 1e3,
            "checked_vs_oracle": args.check}]
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        for r in out:
            print(json.dumps(r), flush=True)
            f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
