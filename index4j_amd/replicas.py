"""One immutable index on several GPUs of a node, one host process (include/fmx.h "replicas").

FmIndex is @ThreadSafe and immutable (FM:82); the reference's own throughput benchmark gives every thread an index of its
own (FmIndexThroughputState.java:30).  Here the image is replicated on the devices named (``fmx_replicate``: peer copies out
of the source's HBM, all destinations at once) and a batch is cut into contiguous shards, one per replica, each run by a
host thread of the library and storing into its own slice of the caller's arrays (``fmx_*_multi``): no collective on the
query path.  This is the C-ABI form of what ``shard.py`` does with one process per GPU over torch.distributed."""
import ctypes as C

import numpy as np

from ._lib import check, lib
from .fmindex import FmIndex


def shard_range(n, parts, part):
    """[lo, hi) of part `part` of `parts` — the library's own arithmetic (fmx_shard_range)"""
    lo, hi = C.c_int64(0), C.c_int64(0)
    lib.fmx_shard_range(int(n), int(parts), int(part), C.byref(lo), C.byref(hi))
    return lo.value, hi.value


class ReplicaSet:
    """replicas of ONE FmIndex on `devices` (a device may be named twice: two replicas then share it)"""

    def __init__(self, source, devices):
        devs = np.ascontiguousarray(devices, dtype=np.int32)
        out = (C.c_void_p * len(devs))()
        check(lib.fmx_replicate(source.handle, devs.ctypes.data, len(devs), out), "fmx_replicate")
        self.replicas = [FmIndex(device=None, _handle=C.c_void_p(h)) for h in out]
        self.devices = [lib.fmx_device_of(r.handle) for r in self.replicas]
        self._handles = (C.c_void_p * len(devs))(*[r.handle for r in self.replicas])

    def __len__(self):
        return len(self.replicas)

    @property
    def handles(self):
        return self._handles

    def close(self):
        for r in self.replicas:
            r.close()
        self.replicas = []

    def resident_bytes(self):
        """[(image, suffix table, window directory)] per replica — fmx_resident_bytes"""
        out = []
        for r in self.replicas:
            a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
            check(lib.fmx_resident_bytes(r.handle, C.byref(a), C.byref(b), C.byref(c)), "fmx_resident_bytes")
            out.append((a.value, b.value, c.value))
        return out

    # ---- the sharded batch calls: same arguments and results as FmIndex.*_batch ----
    def count_batch(self, chars, offsets, want_steps=False):
        chars = np.ascontiguousarray(chars, dtype=np.uint16)
        offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        n = len(offsets) - 1
        counts, steps, status = (np.zeros(n, dtype=np.int32) for _ in range(3))
        check(lib.fmx_count_batch_multi(self._handles, len(self), chars.ctypes.data, offsets.ctypes.data, n, counts.ctypes.data,
                                        steps.ctypes.data, status.ctypes.data), "fmx_count_batch_multi")
        return (counts, status, steps) if want_steps else (counts, status)

    def locate_batch(self, chars, offsets, max_matches, loc_cap=None, want_steps=False, locs=None):
        chars = np.ascontiguousarray(chars, dtype=np.uint16)
        offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        n = len(offsets) - 1
        if loc_cap is None:
            loc_cap = max_matches
        if locs is None:
            locs = np.zeros((n, max(loc_cap, 0)), dtype=np.int32)
        found, steps, status = (np.zeros(n, dtype=np.int32) for _ in range(3))
        check(lib.fmx_locate_batch_multi(self._handles, len(self), chars.ctypes.data, offsets.ctypes.data, n, int(max_matches),
                                         locs.ctypes.data, int(loc_cap), found.ctypes.data, steps.ctypes.data,
                                         status.ctypes.data), "fmx_locate_batch_multi")
        return (locs, found, status, steps) if want_steps else (locs, found, status)

    def extract_batch(self, starts, stops, dst_len, offset=0, dst=None, want_steps=False):
        starts = np.ascontiguousarray(starts, dtype=np.int32)
        stops = np.ascontiguousarray(stops, dtype=np.int32)
        n = len(starts)
        if dst is None:
            dst = np.zeros((n, dst_len), dtype=np.uint16)
        out_len, steps, status = (np.zeros(n, dtype=np.int32) for _ in range(3))
        check(lib.fmx_extract_batch_multi(self._handles, len(self), starts.ctypes.data, stops.ctypes.data, n, dst.ctypes.data,
                                          int(dst_len), int(offset), out_len.ctypes.data, steps.ctypes.data,
                                          status.ctypes.data), "fmx_extract_batch_multi")
        return (dst, out_len, status, steps) if want_steps else (dst, out_len, status)

    def extract_boundary_batch(self, froms, boundary, mode, dst_len, offset=0, dst=None, want_steps=False):
        froms = np.ascontiguousarray(froms, dtype=np.int32)
        n = len(froms)
        if dst is None:
            dst = np.zeros((n, dst_len), dtype=np.uint16)
        out_len, steps, status, aux = (np.zeros(n, dtype=np.int32) for _ in range(4))
        b = boundary if isinstance(boundary, (int, np.integer)) else ord(boundary)
        check(lib.fmx_extract_boundary_batch_multi(self._handles, len(self), froms.ctypes.data, n, int(b), int(mode),
                                                   dst.ctypes.data, int(dst_len), int(offset), out_len.ctypes.data,
                                                   steps.ctypes.data, status.ctypes.data, aux.ctypes.data),
              "fmx_extract_boundary_batch_multi")
        return (dst, out_len, status, aux, steps) if want_steps else (dst, out_len, status, aux)


class SegmentReplicaSet:
    """BASELINE configs[4] behind the C ABI: every segment index of a SegmentedFmIndex replicated on `devices`; a batch is
    sharded over the devices, each shard summed / located over all segments on its device (fmx_count_locate_segments_multi)."""

    def __init__(self, segmented, devices):
        self.devices = [int(d) for d in devices]
        self.n_segs = len(segmented)
        self.sets = [ReplicaSet(seg, self.devices) for seg in segmented.segments]  # sets[s].replicas[r]
        flat = [self.sets[s].replicas[r].handle for r in range(len(self.devices)) for s in range(self.n_segs)]
        self._handles = (C.c_void_p * len(flat))(*flat)  # replica-major
        self._bases = np.asarray(segmented.bases, dtype=np.int64)

    @property
    def handles(self):
        return self._handles

    @property
    def base_array(self):
        return self._bases

    def close(self):
        for s in self.sets:
            s.close()
        self.sets = []

    def count_locate_batch(self, chars, offsets, max_matches, fill=-1):
        chars = np.ascontiguousarray(chars, dtype=np.uint16)
        offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        n = len(offsets) - 1
        counts = np.zeros(n, dtype=np.int64)
        steps = np.zeros(n, dtype=np.int64)
        locs = np.full((n, int(max_matches)), fill, dtype=np.int64)
        found = np.zeros(n, dtype=np.int32)
        status = np.zeros(n, dtype=np.int32)
        check(lib.fmx_count_locate_segments_multi(self._handles, len(self.devices), self.n_segs, self._bases.ctypes.data,
                                                  chars.ctypes.data, offsets.ctypes.data, n, int(max_matches), counts.ctypes.data,
                                                  steps.ctypes.data, locs.ctypes.data, found.ctypes.data, status.ctypes.data),
              "fmx_count_locate_segments_multi")
        return counts, locs, found, status, steps
