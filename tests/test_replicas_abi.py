"""The replica calls of the C ABI without a GPU (include/fmx.h "replicas"): the shard arithmetic every fmx_*_multi call cuts a
batch with — it must be shard.py's (the one-process-per-GPU form) for any n and replica count, n not divisible included — and
the argument / residency checks, which come before any HIP call."""
import ctypes as C

import numpy as np

import index4j_amd as ia
from index4j_amd.shard import shard_range as dist_shard_range


def test_shard_range_is_contiguous_balanced_and_the_distributed_form():
    for n in [0, 1, 2, 7, 8, 9, 63, 64, 65, 100_000, 1_048_576, 8_388_608, (1 << 31) - 1, 1_000_003]:
        for parts in [1, 2, 3, 4, 5, 7, 8, 16]:
            got = [ia.shard_range(n, parts, p) for p in range(parts)]
            assert got == [dist_shard_range(n, parts, p) for p in range(parts)], (n, parts)
            assert got[0][0] == 0 and got[-1][1] == n
            assert all(got[p][1] == got[p + 1][0] for p in range(parts - 1))
            sizes = [hi - lo for lo, hi in got]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)  # the first n % parts parts hold the extra item
    # outside the domain: an empty range, never garbage
    for n, parts, part in [(10, 0, 0), (10, 3, 3), (10, 3, -1), (-5, 2, 0)]:
        assert ia.shard_range(n, parts, part) == (0, 0)
    ia.lib.fmx_shard_range(10, 2, 1, None, None)  # null outputs are allowed


def test_replica_calls_check_arguments_and_residency_before_any_device_work():
    L = ia.lib
    fm = ia.FmIndex("a small text\nwith two lines\n", 4, True, device=None)
    assert L.fmx_device_of(fm.handle) == -1 and L.fmx_device_of(None) == -1
    a, b, c = C.c_int64(-1), C.c_int64(-1), C.c_int64(-1)
    assert L.fmx_resident_bytes(fm.handle, C.byref(a), C.byref(b), C.byref(c)) == 0 and (a.value, b.value, c.value) == (0, 0, 0)
    assert L.fmx_resident_bytes(None, None, None, None) == ia._lib.E_ARG
    out = (C.c_void_p * 2)()
    devs = np.array([0, 0], np.int32)
    assert L.fmx_replicate(None, devs.ctypes.data, 2, out) == ia._lib.E_ARG
    assert L.fmx_replicate(fm.handle, None, 2, out) == ia._lib.E_ARG
    assert L.fmx_replicate(fm.handle, devs.ctypes.data, 0, out) == ia._lib.E_ARG
    if L.fmx_device_count() == 0:
        assert L.fmx_replicate(fm.handle, devs.ctypes.data, 2, out) == ia._lib.E_NO_DEVICE
        assert out[0] is None and out[1] is None
    ch, off = ia.pack_patterns(["text", "two"])
    z = np.zeros(2, np.int32)
    z64 = np.zeros(2, np.int64)
    hs = (C.c_void_p * 2)(fm.handle, fm.handle)
    E_ND, E_ARG = ia._lib.E_NO_DEVICE, ia._lib.E_ARG
    assert L.fmx_count_batch_multi(hs, 2, ch.ctypes.data, off.ctypes.data, 2, z.ctypes.data, None, None) == E_ND
    assert b"not resident" in L.fmx_last_error()
    assert L.fmx_locate_batch_multi(hs, 2, ch.ctypes.data, off.ctypes.data, 2, 4, z.ctypes.data, 1, z.ctypes.data, None, None) == E_ND
    assert L.fmx_extract_batch_multi(hs, 2, z.ctypes.data, z.ctypes.data, 2, None, 0, 0, z.ctypes.data, None, None) == E_ND
    assert L.fmx_extract_boundary_batch_multi(hs, 2, z.ctypes.data, 2, 10, 0, None, 0, 0, z.ctypes.data, None, None, None) == E_ND
    assert L.fmx_extract_boundary_batch_multi(hs, 2, z.ctypes.data, 2, 10, 3, None, 0, 0, z.ctypes.data, None, None, None) == E_ARG
    assert L.fmx_count_locate_segments_multi(hs, 2, 1, z64.ctypes.data, ch.ctypes.data, off.ctypes.data, 2, 1, z64.ctypes.data, None,
                                             z64.ctypes.data, z.ctypes.data, None) == E_ND
    assert L.fmx_count_locate_segments_multi(hs, 2, 0, z64.ctypes.data, ch.ctypes.data, off.ctypes.data, 2, 1, z64.ctypes.data, None,
                                             z64.ctypes.data, z.ctypes.data, None) == E_ARG
    assert L.fmx_count_batch_multi(None, 2, ch.ctypes.data, off.ctypes.data, 2, z.ctypes.data, None, None) == E_ARG
    assert L.fmx_count_batch_multi(hs, 0, ch.ctypes.data, off.ctypes.data, 2, z.ctypes.data, None, None) == E_ARG
    assert L.fmx_multi_synchronize(hs, 2, None) == E_ND
    ns = (C.c_int32 * 2)(1, 1)
    assert L.fmx_count_batch_multi_dev(hs, 2, hs, hs, ns, hs, None, None, None) == E_ND
    fm.close()
