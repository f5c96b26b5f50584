#!/usr/bin/env python3
"""The same index as an EXPANDED image (default) and as a COMPACT one (option image_compact: RRR records + offsets streams,
value table in LDS): resident bytes per text byte and the time of configs[1] (count), configs[2] (locate) and configs[3]
(extractUntilBoundary) over each, results compared with each other and counts with the oracle.  GPU box only.
usage: python tools/compact_vs_expanded.py [--text-log2 28] [--segments]"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--text-log2", type=int, default=28)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "compact_vs_expanded.json"))
    args = ap.parse_args()
    import torch

    import bench
    import index4j_amd as ia
    import orc

    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream()
    sp = C.c_void_p(stream.cuda_stream)
    n_text = 1 << args.text_log2
    text, fm32, path32 = bench.build_or_load_index(ia, args.text_log2, 32, "/tmp/fmx_cache")
    _t, fm64, path64 = bench.build_or_load_index(ia, args.text_log2, 64, "/tmp/fmx_cache")
    ser32, ser64 = fm32.write(False), fm64.write(False)
    fm32.close()
    fm64.close()
    ref = orc.OracleFmIndex.read(ser32)
    n, K, M, cap = 1 << 20, 100_000, 16, 1024
    pat, off, _ = ia.synth_patterns(text, 8, n, seed=43)
    oc, _ost = ref.count_batch(pat, off, threads=os.cpu_count() or 1)
    d_pat = torch.from_numpy(pat.view(np.int16)).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    d_cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    d_locs = torch.zeros(K * M, dtype=torch.int32, device=dev)
    d_found = torch.zeros(K, dtype=torch.int32, device=dev)
    d_st = torch.zeros(n, dtype=torch.int32, device=dev)
    d_rng = torch.zeros(2 * K, dtype=torch.int32, device=dev)
    d_dst = torch.zeros(K * cap, dtype=torch.int16, device=dev)
    d_len = torch.zeros(K, dtype=torch.int32, device=dev)
    d_aux = torch.zeros(K, dtype=torch.int32, device=dev)

    def mean_ms(fn, reps):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    rows, keep = [], {}
    for compact in (0, 1):
        assert ia.lib.fmx_set_option(b"image_compact", compact) == 0
        a = ia.FmIndex.read(ser32, device=0)
        b = ia.FmIndex.read(ser64, device=0)
        img, tbl = a.device_blob()[1], a.suffix_table_info()

        def count():
            assert ia.lib.fmx_count_batch_dev(a.handle, d_pat.data_ptr(), d_off.data_ptr(), n, d_cnt.data_ptr(), None, None, sp) == 0

        def locate(ix=a):
            assert ia.lib.fmx_locate_batch_dev(ix.handle, d_pat.data_ptr(), d_off.data_ptr(), K, M, d_locs.data_ptr(), M,
                                               d_found.data_ptr(), None, d_st.data_ptr(), d_rng.data_ptr(), sp) == 0

        t_count = mean_ms(count, 20)
        if not (d_cnt.cpu().numpy() == oc).all():
            raise RuntimeError("counts differ from the oracle (compact %d)" % compact)
        t_locate = mean_ms(locate, 10)
        locs32 = d_locs.cpu().numpy().copy()
        found32 = d_found.cpu().numpy().copy()
        locate(b)
        torch.cuda.synchronize()
        froms = np.ascontiguousarray(d_locs.cpu().numpy().reshape(K, M)[:, 0])
        d_from = torch.from_numpy(froms).to(dev)

        def boundary():
            assert ia.lib.fmx_extract_boundary_batch_dev(b.handle, d_from.data_ptr(), K, 10, 0, d_dst.data_ptr(), cap, 0,
                                                         d_len.data_ptr(), None, d_st.data_ptr(), d_aux.data_ptr(), sp) == 0

        t_boundary = mean_ms(boundary, 5)
        res = (locs32, found32, d_dst.cpu().numpy().copy(), d_len.cpu().numpy().copy())
        if compact == 0:
            keep = res
        else:
            live = np.arange(M)[None, :] < res[1][:, None]
            same = (res[1] == keep[1]).all() and (res[0].reshape(K, M)[live] == keep[0].reshape(K, M)[live]).all() and \
                (res[2] == keep[2]).all() and (res[3] == keep[3]).all()
            if not same:
                raise RuntimeError("compact and expanded images disagree")
        rows.append({"image": "compact" if compact else "expanded", "image_bytes_per_text_byte_s32": img / n_text,
                     "image_bytes_per_text_byte_s64": b.device_blob()[1] / n_text,
                     "suffix_table_chars": tbl[0], "resident_bytes_per_text_byte_s32": (img + tbl[1]) / n_text,
                     "serialized_bytes_per_text_byte_s32": len(ser32) / n_text,
                     "configs1_count_ms": t_count, "configs2_locate_ms": t_locate, "configs3_boundary_ms": t_boundary})
        print(json.dumps(rows[-1]), flush=True)
        a.close()
        b.close()
    ia.lib.fmx_set_option(b"image_compact", 0)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(rows, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
