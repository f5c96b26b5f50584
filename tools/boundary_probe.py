#!/usr/bin/env python3
"""configs[3] (100,000 hit locations, sampleRate 64, extractUntilBoundary) under option settings given on the command line, each
checked against the oracle: python tools/boundary_probe.py "boundary_group=4" "boundary_group=2" "boundary_narrow=1" ...
(an option set is a comma-separated list; options stay set for the sets that follow unless they set them again)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    import torch

    import bench
    import index4j_amd as ia
    import orc
    from bench_configs import timed
    from index4j_amd import workload

    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream()
    sp = C.c_void_p(stream.cuda_stream)
    K, cap = int(os.environ.get("PROBE_K", "100000")), 1024
    text, fm64, path64 = bench.build_or_load_index(ia, 28, 64, "/tmp/fmx_cache")
    fm64.to_device(0)
    o64 = orc.OracleFmIndex.read(open(path64, "rb").read())
    pat, off, _pos = workload.count_batch_patterns(text, K, 8)
    locs, found, st = fm64.locate_batch(pat, off, 1, 1)
    froms = np.ascontiguousarray(locs[:, 0]).astype(np.int32)
    d_from = torch.from_numpy(froms).to(dev)
    d_dst = torch.zeros(K * cap, dtype=torch.int16, device=dev)
    d_len = torch.zeros(K, dtype=torch.int32, device=dev)
    d_lf = torch.zeros(K, dtype=torch.int32, device=dev)
    d_st = torch.zeros(K, dtype=torch.int32, device=dev)
    d_aux = torch.zeros(K, dtype=torch.int32, device=dev)
    expect = {}
    for opts in sys.argv[1:] or [""]:
        for kv in filter(None, opts.split(",")):
            k, _, v = kv.partition("=")
            assert ia.lib.fmx_set_option(k.strip().encode(), int(v)) == 0, kv
        for mode in (0, 1, 2):
            def call(with_lf=False):
                rc = ia.lib.fmx_extract_boundary_batch_dev(fm64.handle, d_from.data_ptr(), K, 10, mode, d_dst.data_ptr(), cap, 0, d_len.data_ptr(),
                                                           d_lf.data_ptr() if with_lf else None, d_st.data_ptr(), d_aux.data_ptr(), sp)
                assert rc == 0, ia.lib.fmx_last_error()

            d_dst.zero_()
            call(True)
            torch.cuda.synchronize()
            if mode not in expect:
                expect[mode] = o64.extract_until_boundary_batch(mode, froms, "\n", cap, threads=os.cpu_count() or 1)
            odst, olen, ost, oaux = expect[mode]
            dst = d_dst.cpu().numpy().view(np.uint16).reshape(K, cap)
            ok = (d_len.cpu().numpy() == olen).all() and (d_st.cpu().numpy() == ost).all() and (dst == odst).all()
            steps = int(d_lf.sum(dtype=torch.int64).item())
            print("[%s] mode %d: %.3f ms, %.1f M LF-steps walked, %s" % (opts, mode, timed(lambda: call(False), stream, 7), steps / 1e6,
                                                                          "every row = the oracle's" if ok else "ROWS DIFFER FROM THE ORACLE"), flush=True)


if __name__ == "__main__":
    main()
