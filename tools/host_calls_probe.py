#!/usr/bin/env python3
"""The host-buffer entry points a JNI binding calls, wall time per call with pageable and with registered arrays:
fmx_locate_batch (configs[2]'s shape) and fmx_extract_boundary_batch (configs[3]'s), beside fmx_count_batch.  GPU box only."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import index4j_amd as ia  # noqa: E402

text, fm, _ = bench.build_or_load_index(ia, 28, 32, "/tmp/fmx_cache")
fm.to_device(0)
K, M, m = 100_000, 16, 8
pat, off, _ = ia.synth_patterns(text, m, K, seed=43)
locs = np.zeros((K, M), np.int32)
found = np.zeros(K, np.int32)
st = np.zeros(K, np.int32)


def timed(fn, reps=12):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.mean(ts)), float(np.median(ts)), min(ts)


def locate():
    assert ia.lib.fmx_locate_batch(fm.handle, pat.ctypes.data, off.ctypes.data, K, M, locs.ctypes.data, M, found.ctypes.data, None, st.ctypes.data) == 0


print("fmx_locate_batch 100,000 x <= 16 hits, pageable:   mean %.3f median %.3f min %.3f ms" % timed(locate), flush=True)
ref = locs.copy()
arrays = (pat, off, locs, found, st)
for a in arrays:
    assert ia.lib.fmx_host_register(a.ctypes.data, a.nbytes) == 0
locs[:] = 0
print("fmx_locate_batch 100,000 x <= 16 hits, registered: mean %.3f median %.3f min %.3f ms" % timed(locate), flush=True)
assert (locs == ref).all()
for a in arrays:
    ia.lib.fmx_host_unregister(a.ctypes.data)

fm64_text, fm64, _ = bench.build_or_load_index(ia, 28, 64, "/tmp/fmx_cache")
fm64.to_device(0)
l64, f64, _ = fm64.locate_batch(pat, off, 1)
frm = np.ascontiguousarray(l64[:, 0])
cap = 1024
dst = np.zeros((K, cap), np.uint16)
ol = np.zeros(K, np.int32)
st2 = np.zeros(K, np.int32)
aux = np.zeros(K, np.int32)


def boundary():
    assert ia.lib.fmx_extract_boundary_batch(fm64.handle, frm.ctypes.data, K, 10, 0, dst.ctypes.data, cap, 0, ol.ctypes.data, None, st2.ctypes.data,
                                             aux.ctypes.data) == 0


print("fmx_extract_boundary_batch 100,000 lines (dest rows of %d chars = %d MB), pageable:   mean %.3f median %.3f min %.3f ms"
      % ((cap, dst.nbytes >> 20) + timed(boundary, 6)), flush=True)
ref = dst.copy()
arrays = (frm, dst, ol, st2, aux)
for a in arrays:
    assert ia.lib.fmx_host_register(a.ctypes.data, a.nbytes) == 0
dst[:] = 0
print("fmx_extract_boundary_batch 100,000 lines, registered: mean %.3f median %.3f min %.3f ms" % timed(boundary, 6), flush=True)
assert (dst == ref).all()
for a in arrays:
    ia.lib.fmx_host_unregister(a.ctypes.data)

# the fused pipeline of a "grep": locate + the line of every hit (fmx_locate_lines_batch), 20,000 patterns x <= 8 hits, rows of 256 chars
K2, M2, cap2 = 20_000, 8, 256
pat2, off2 = pat[: K2 * m], off[: K2 + 1]
locs2 = np.zeros((K2, M2), np.int32)
found2 = np.zeros(K2, np.int32)
dst2 = np.zeros((K2 * M2, cap2), np.uint16)
ol2 = np.zeros(K2 * M2, np.int32)
st3 = np.zeros(K2, np.int32)
hst = np.zeros(K2 * M2, np.int32)
haux = np.zeros(K2 * M2, np.int32)


def lines():
    assert ia.lib.fmx_locate_lines_batch(fm.handle, pat2.ctypes.data, off2.ctypes.data, K2, M2, 10, 0, cap2, locs2.ctypes.data, found2.ctypes.data,
                                         dst2.ctypes.data, ol2.ctypes.data, None, st3.ctypes.data, hst.ctypes.data, haux.ctypes.data) == 0


print("fmx_locate_lines_batch 20,000 patterns x <= 8 hits, rows of 256 chars (%d MB): mean %.3f median %.3f min %.3f ms"
      % ((dst2.nbytes >> 20,) + timed(lines, 8)), flush=True)
