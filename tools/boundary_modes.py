#!/usr/bin/env python3
"""configs[3]'s 100,000 hit locations through extractUntilBoundary / ...Left / ...Right (modes 0 / 1 / 2): HIP-event time per
mode, every row checked against the oracle.  usage: python tools/boundary_modes.py"""
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
    K, cap = 100_000, 1024
    text, fm64, path64 = bench.build_or_load_index(ia, 28, 64, "/tmp/fmx_cache")
    fm64.to_device(0)
    o64 = orc.OracleFmIndex.read(open(path64, "rb").read())
    pat, off, _pos = workload.count_batch_patterns(text, K, 8)
    locs, found, st = fm64.locate_batch(pat, off, 1, 1)
    froms = np.ascontiguousarray(locs[:, 0]).astype(np.int32)
    d_from = torch.from_numpy(froms).to(dev)
    d_dst = torch.zeros(K * cap, dtype=torch.int16, device=dev)
    d_len = torch.zeros(K, dtype=torch.int32, device=dev)
    d_st = torch.zeros(K, dtype=torch.int32, device=dev)
    d_aux = torch.zeros(K, dtype=torch.int32, device=dev)
    for mode, name in ((0, "extractUntilBoundary"), (1, "extractUntilBoundaryLeft"), (2, "extractUntilBoundaryRight")):
        def call():
            rc = ia.lib.fmx_extract_boundary_batch_dev(fm64.handle, d_from.data_ptr(), K, 10, mode, d_dst.data_ptr(), cap, 0, d_len.data_ptr(),
                                                       None, d_st.data_ptr(), d_aux.data_ptr(), sp)
            assert rc == 0, ia.lib.fmx_last_error()

        d_dst.zero_()
        call()
        torch.cuda.synchronize()
        odst, olen, ost, oaux = o64.extract_until_boundary_batch(mode, froms, "\n", cap, threads=os.cpu_count() or 1)
        dst = d_dst.cpu().numpy().view(np.uint16).reshape(K, cap)
        assert (d_len.cpu().numpy() == olen).all() and (d_st.cpu().numpy() == ost).all() and (dst == odst).all(), name
        print("%-28s %d queries: %.3f ms (every row = the oracle's)" % (name, K, timed(call, stream, 5)), flush=True)


if __name__ == "__main__":
    main()
