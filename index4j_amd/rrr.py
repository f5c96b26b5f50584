"""Host-side mirror of index4j's RrrVector (sdsl/RrrVector.java) as a stand-alone structure on the GPU: the
compressed form (15-bit blocks as class + offset, RRR:225-286) with rankOnes / rankZeroes / access answered by
HIP kernels that stage the value-of-offset table in LDS.  Inside an FM-index image the bit vectors are expanded
instead (csrc/fmx_blob.hpp); this class is the reference's public RrrVector for callers that use it directly."""
import ctypes as C

import numpy as np

from ._lib import check, lib


class RrrVector:
    def __init__(self, bits, sampleSize=32, device=0):
        """bits: iterable of 0/1 (the reference takes a BitVector)"""
        a = np.ascontiguousarray(bits, dtype=np.uint8)
        self._n = len(a)
        self._h = C.c_void_p()
        check(lib.fmx_rrr_build(a.ctypes.data, len(a), int(sampleSize), C.byref(self._h)), "fmx_rrr_build")
        if device is not None:
            check(lib.fmx_to_device(self._h, int(device)), "fmx_to_device")

    def close(self):
        if self._h:
            lib.fmx_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    def __len__(self):
        return self._n

    # ---- batched ----
    def rank_ones_batch(self, positions):
        p = np.ascontiguousarray(positions, dtype=np.int32)
        out = np.zeros(len(p), dtype=np.int32)
        check(lib.fmx_rrr_rank_ones_batch(self._h, p.ctypes.data, len(p), out.ctypes.data), "fmx_rrr_rank_ones_batch")
        return out

    def access_batch(self, positions):
        p = np.ascontiguousarray(positions, dtype=np.int32)
        out = np.zeros(len(p), dtype=np.uint8)
        status = np.zeros(len(p), dtype=np.int32)
        check(lib.fmx_rrr_access_batch(self._h, p.ctypes.data, len(p), out.ctypes.data, status.ctypes.data),
              "fmx_rrr_access_batch")
        return out, status

    # ---- the reference's scalar methods ----
    def rankOnes(self, position):  # RRR:358-396
        return int(self.rank_ones_batch([position])[0])

    def rankZeroes(self, position):  # RRR:398-409
        return 0 if position < 0 else position - self.rankOnes(position)

    def access(self, position):  # RRR:314-349
        bit, status = self.access_batch([position])
        if status[0]:
            raise ValueError("Out of range access. Requested %d" % position)
        return bool(bit[0])
