"""Host-side mirror of com.dynatrace.wavelet.WaveletFixedBlockBoosting (constructor WFBB:130-154, rank
WFBB:1010-1285, inverseSelect WFBB:1305-1537) over the C ABI.  Queries run on the GPU."""
import ctypes as C

import numpy as np

from ._lib import check, lib
from .fmindex import raise_for_status


class WaveletFixedBlockBoosting:
    def __init__(self, text, samplingRate=64, device=0):
        """`text`: symbols mapped to small non-negative integers (the short[] of WFBB:130), or a str whose
        UTF-16 code units are used as symbols (the char[] constructor, WFBB:176-211)"""
        if isinstance(text, str):
            text = np.frombuffer(text.encode("utf-16-le"), dtype=np.uint16)
        seq = np.ascontiguousarray(text, dtype=np.int16)
        if len(seq) == 0:
            raise ValueError("Input length must be > 0")  # WFBB:178-180
        h = C.c_void_p()
        check(lib.fmx_wavelet_build(seq.ctypes.data, len(seq), int(samplingRate), C.byref(h)), "fmx_wavelet_build")
        self._h = h
        self.size = len(seq)
        if device is not None:
            check(lib.fmx_to_device(self._h, int(device)), "fmx_to_device")

    def blob(self):
        p, n = C.c_void_p(), C.c_size_t()
        check(lib.fmx_blob(self._h, C.byref(p), C.byref(n)), "fmx_blob")
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n.value,))

    def rank_batch(self, positions, symbols):
        pos = np.ascontiguousarray(positions, dtype=np.int64)
        sym = np.ascontiguousarray(symbols, dtype=np.int32)
        out = np.zeros(len(pos), dtype=np.int64)
        st = np.zeros(len(pos), dtype=np.int32)
        check(lib.fmx_wavelet_rank_batch(self._h, pos.ctypes.data, sym.ctypes.data, len(pos), out.ctypes.data, st.ctypes.data),
              "fmx_wavelet_rank_batch")
        return out, st

    def inverse_select_batch(self, positions):
        pos = np.ascontiguousarray(positions, dtype=np.int64)
        out = np.zeros(len(pos), dtype=np.int64)
        st = np.zeros(len(pos), dtype=np.int32)
        check(lib.fmx_wavelet_inverse_select_batch(self._h, pos.ctypes.data, len(pos), out.ctypes.data, st.ctypes.data),
              "fmx_wavelet_inverse_select_batch")
        return out, st

    def rank(self, position, symbol):  # WFBB:1010
        symbol = ord(symbol) if isinstance(symbol, str) else int(symbol)
        if symbol > 32767:
            symbol -= 65536  # (short) symbol, WFBB:1294-1296
        out, st = self.rank_batch([position], [symbol])
        raise_for_status(st[0])
        return int(out[0])

    def inverseSelect(self, position):  # WFBB:1305
        out, st = self.inverse_select_batch([position])
        raise_for_status(st[0])
        return int(out[0])

    def close(self):
        if self._h is not None:
            lib.fmx_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
