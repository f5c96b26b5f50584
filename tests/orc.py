"""ctypes binding of oracle/liboracle.so — test infrastructure only (never imported by index4j_amd)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = None


class Counters(C.Structure):
    _fields_ = [
        ("lf_steps", C.c_uint64),
        ("alg_bytes", C.c_uint64),
        ("wt_levels", C.c_uint64),
        ("quirk_runblock_right", C.c_uint64),
        ("quirk_clamped_right", C.c_uint64),
        ("rank_calls", C.c_uint64),
        ("absent_superblock", C.c_uint64),
        ("absent_block", C.c_uint64),
        ("absent_scan_steps", C.c_uint64),
        ("run_block", C.c_uint64),
    ]


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    src = os.path.join(ORACLE_DIR, "index4j_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    L = C.CDLL(so)
    vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
    p = C.POINTER
    L.orc_rrr_from_bits.restype = vp
    L.orc_rrr_from_bits.argtypes = [vp, i64, i32]
    L.orc_rrr_from_ints.restype = vp
    L.orc_rrr_from_ints.argtypes = [vp, i32, i32]
    L.orc_rrr_access.argtypes = [vp, i32, p(i32)]
    L.orc_rrr_rank_ones.argtypes = [vp, i32]
    L.orc_rrr_rank_zeroes.argtypes = [vp, i32]
    L.orc_rrr_rank_ones_batch.argtypes = [vp, vp, i32, vp, i32]
    L.orc_rrr_rank_ones_batch.restype = None
    L.orc_rrr_estimated_memory.argtypes = [vp]
    L.orc_rrr_free.argtypes = [vp]
    for n in ("offset_of_value", "value_of_offset", "cardinality_offsets"):
        getattr(L, "orc_rrr_table_" + n).restype = p(C.c_uint16)
    L.orc_rrr_table_bits_needed.restype = p(i32)
    L.orc_wfbb_build.restype = vp
    L.orc_wfbb_build.argtypes = [vp, i64, i32]
    L.orc_wfbb_rank.restype = i64
    L.orc_wfbb_rank.argtypes = [vp, i64, C.c_int16, p(i32)]
    L.orc_wfbb_inverse_select.restype = i64
    L.orc_wfbb_inverse_select.argtypes = [vp, i64]
    L.orc_wfbb_block_size_log.argtypes = [vp, i32]
    L.orc_wfbb_free.argtypes = [vp]
    L.orc_fm_build.restype = vp
    L.orc_fm_build.argtypes = [vp, i32, i32, i32, p(i32)]
    L.orc_fm_free.argtypes = [vp]
    L.orc_fm_input_length.argtypes = [vp]
    L.orc_fm_alphabet_length.argtypes = [vp]
    L.orc_fm_sample_rate.argtypes = [vp]
    L.orc_fm_wavelet.restype = vp
    L.orc_fm_wavelet.argtypes = [vp]
    L.orc_fm_count.argtypes = [vp, vp, i32, i32, p(i32)]
    L.orc_fm_locate.argtypes = [vp, vp, i32, i32, vp, i32, i32, p(i32)]
    L.orc_fm_extract.argtypes = [vp, i32, i32, vp, i32, i32, p(i32)]
    L.orc_fm_extract_until_boundary.argtypes = [vp, i32, i32, vp, i32, i32, C.c_uint16, p(i32), p(i32)]
    L.orc_fm_count_batch.argtypes = [vp, vp, vp, i32, vp, vp, i32]
    L.orc_fm_count_batch.restype = None
    L.orc_fm_locate_batch.argtypes = [vp, vp, vp, i32, i32, vp, i32, vp, vp, i32]
    L.orc_fm_locate_batch.restype = None
    L.orc_fm_extract_batch.argtypes = [vp, vp, vp, i32, vp, i32, i32, vp, vp, i32]
    L.orc_fm_extract_batch.restype = None
    L.orc_fm_extract_until_boundary_batch.argtypes = [vp, i32, vp, i32, C.c_uint16, vp, i32, i32, vp, vp, vp, i32]
    L.orc_fm_extract_until_boundary_batch.restype = None
    L.orc_fm_write.argtypes = [vp, i32, p(vp), p(C.c_size_t)]
    L.orc_fm_read.restype = vp
    L.orc_fm_read.argtypes = [vp, C.c_size_t, p(i32)]
    L.orc_free_buffer.argtypes = [vp]
    L.orc_convert_byte_pattern.argtypes = [vp, i32, i32, vp, p(i32)]
    L.orc_counters_get.argtypes = [p(Counters)]
    _LIB = L
    return L


def u16(text):
    """str / bytes-like / array -> contiguous uint16 array of UTF-16 code units (Java char[])."""
    if isinstance(text, str):
        return np.frombuffer(text.encode("utf-16-le"), dtype=np.uint16).copy()
    return np.ascontiguousarray(text, dtype=np.uint16)


MESSAGES = {
    1: (RuntimeError, "Text recovery not enabled at build time"),
    2: (RuntimeError, "Requested position less than 0"),
    3: (RuntimeError, "Stop position longer than index string"),
    4: (RuntimeError, "Supplied destination is not large enough"),
    5: (RuntimeError, "Requested position longer than index string"),
    6: (ValueError, "Supplied destination for extraction has size zero"),
    7: (ValueError, "Boundary does not exist"),
    8: (RuntimeError, "Extraction does not fit in the supplied destination. Currently extracted: %d"),
    9: (IndexError, "ArrayIndexOutOfBoundsException"),
}


def raise_status(st, aux=0):
    exc, msg = MESSAGES[st]
    raise exc(msg % aux if "%d" in msg else msg)


class Rrr:
    def __init__(self, bits=None, ints=None, sample=32):
        L = lib()
        if ints is not None:
            a = np.ascontiguousarray(ints, dtype=np.int32)
            self.h = L.orc_rrr_from_ints(a.ctypes.data, len(a), sample)
        else:
            a = np.ascontiguousarray(bits, dtype=np.uint8)
            self.h = L.orc_rrr_from_bits(a.ctypes.data, len(a), sample)

    def access(self, pos):
        st = C.c_int(0)
        v = lib().orc_rrr_access(self.h, pos, C.byref(st))
        if st.value:
            raise ValueError("Out of range access. Requested %d" % pos)
        return bool(v)

    def rank_ones(self, pos):
        return lib().orc_rrr_rank_ones(self.h, pos)

    def rank_zeroes(self, pos):
        return lib().orc_rrr_rank_zeroes(self.h, pos)

    def rank_ones_batch(self, positions, threads=1):
        p = np.ascontiguousarray(positions, dtype=np.int32)
        out = np.zeros(len(p), np.int32)
        lib().orc_rrr_rank_ones_batch(self.h, p.ctypes.data, len(p), out.ctypes.data, int(threads))
        return out

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_rrr_free(self.h)
            self.h = None


class Wfbb:
    def __init__(self, seq, sampling_rate=64):
        a = np.ascontiguousarray(seq, dtype=np.int16)
        self.n = len(a)
        self.h = lib().orc_wfbb_build(a.ctypes.data, len(a), sampling_rate)

    def rank(self, pos, sym):
        st = C.c_int(0)
        v = lib().orc_wfbb_rank(self.h, pos, sym, C.byref(st))
        if st.value:
            raise_status(st.value)
        return v

    def inverse_select(self, pos):
        return lib().orc_wfbb_inverse_select(self.h, pos)

    def block_size_log(self, sb):
        return lib().orc_wfbb_block_size_log(self.h, sb)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_wfbb_free(self.h)
            self.h = None


class OracleFmIndex:
    """Mirrors com.dynatrace.fm.FmIndex's public surface over the C oracle."""

    def __init__(self, text=None, sample_rate=32, enable_extract=True, handle=None):
        L = lib()
        if handle is not None:
            self.h = handle
            return
        a = u16(text)
        st = C.c_int(0)
        self.h = L.orc_fm_build(a.ctypes.data, len(a), sample_rate, int(enable_extract), C.byref(st))
        if not self.h:
            raise ValueError("Input has more than 32767 different symbols")

    @classmethod
    def read(cls, data):
        buf = np.frombuffer(bytes(data), dtype=np.uint8)
        st = C.c_int(0)
        h = lib().orc_fm_read(buf.ctypes.data, len(buf), C.byref(st))
        if not h:
            if st.value == 2:
                raise IOError("Incompatible serial versions!")
            raise IOError("malformed stream (%d)" % st.value)
        return cls(handle=h)

    def write(self, framed=True):
        buf = C.c_void_p()
        n = C.c_size_t()
        lib().orc_fm_write(self.h, int(framed), C.byref(buf), C.byref(n))
        out = bytes((C.c_ubyte * n.value).from_address(buf.value))  # (string_at takes a C int; streams may pass 2 GiB)
        lib().orc_free_buffer(buf)
        return out

    def getInputLength(self):
        return lib().orc_fm_input_length(self.h)

    def getAlphabetLength(self):
        return lib().orc_fm_alphabet_length(self.h)

    def wavelet_handle(self):
        return lib().orc_fm_wavelet(self.h)

    def count(self, pattern, offset=0, length=None):
        p = u16(pattern)
        if length is None:
            length = len(p)
        st = C.c_int(0)
        v = lib().orc_fm_count(self.h, p.ctypes.data, offset, length, C.byref(st))
        if st.value:
            raise_status(st.value)
        return v

    def locate(self, pattern, offset=0, length=None, max_matches=-1, cap=None):
        p = u16(pattern)
        if length is None:
            length = len(p)
        if cap is None:
            cap = max_matches if max_matches > 0 else 1 << 20
        locs = np.zeros(max(cap, 1), dtype=np.int32)
        st = C.c_int(0)
        n = lib().orc_fm_locate(self.h, p.ctypes.data, offset, length, locs.ctypes.data, cap, max_matches, C.byref(st))
        if st.value:
            raise_status(st.value)
        return n, locs[:n].copy()

    def extract(self, start, stop, dest_len=None, offset=0, dest=None):
        if dest is None:
            dest = np.zeros(dest_len if dest_len is not None else max(stop - start + offset, 0), dtype=np.uint16)
        st = C.c_int(0)
        n = lib().orc_fm_extract(self.h, start, stop, dest.ctypes.data, len(dest), offset, C.byref(st))
        if st.value:
            raise_status(st.value)
        return n, dest

    def extract_until_boundary(self, mode, frm, dest_len, offset, boundary, dest=None):
        if dest is None:
            dest = np.zeros(dest_len, dtype=np.uint16)
        st = C.c_int(0)
        aux = C.c_int(0)
        b = boundary if isinstance(boundary, int) else ord(boundary)
        n = lib().orc_fm_extract_until_boundary(self.h, mode, frm, dest.ctypes.data, len(dest), offset, b, C.byref(st), C.byref(aux))
        if st.value:
            raise_status(st.value, aux.value)
        return n, dest

    def count_batch(self, pat, pat_off, threads=1):
        pat = np.ascontiguousarray(pat, dtype=np.uint16)
        pat_off = np.ascontiguousarray(pat_off, dtype=np.int32)
        n = len(pat_off) - 1
        counts = np.zeros(n, dtype=np.int32)
        status = np.zeros(n, dtype=np.int32)
        lib().orc_fm_count_batch(self.h, pat.ctypes.data, pat_off.ctypes.data, n, counts.ctypes.data, status.ctypes.data, threads)
        return counts, status

    def locate_batch(self, pat, pat_off, max_matches, loc_cap=None, threads=1, fill=0):
        """FM:504-552 looped in C (OpenMP over patterns): (locs[n, loc_cap], found, status)"""
        pat = np.ascontiguousarray(pat, dtype=np.uint16)
        pat_off = np.ascontiguousarray(pat_off, dtype=np.int32)
        n = len(pat_off) - 1
        if loc_cap is None:
            loc_cap = max_matches
        locs = np.full((n, max(loc_cap, 0)), fill, dtype=np.int32)
        found = np.zeros(n, dtype=np.int32)
        status = np.zeros(n, dtype=np.int32)
        lib().orc_fm_locate_batch(self.h, pat.ctypes.data, pat_off.ctypes.data, n, int(max_matches), locs.ctypes.data,
                                  int(loc_cap), found.ctypes.data, status.ctypes.data, threads)
        return locs, found, status

    def extract_batch(self, starts, stops, dst_len, offset=0, threads=1, fill=0):
        """FM:564-608 looped in C: (dst[n, dst_len], out_len, status)"""
        starts = np.ascontiguousarray(starts, dtype=np.int32)
        stops = np.ascontiguousarray(stops, dtype=np.int32)
        n = len(starts)
        dst = np.full((n, dst_len), fill, dtype=np.uint16)
        out_len = np.zeros(n, dtype=np.int32)
        status = np.zeros(n, dtype=np.int32)
        lib().orc_fm_extract_batch(self.h, starts.ctypes.data, stops.ctypes.data, n, dst.ctypes.data, int(dst_len),
                                   int(offset), out_len.ctypes.data, status.ctypes.data, threads)
        return dst, out_len, status

    def extract_until_boundary_batch(self, mode, froms, boundary, dst_len, offset=0, threads=1, fill=0):
        """FM:640-922 looped in C: (dst[n, dst_len], out_len, status, aux)"""
        froms = np.ascontiguousarray(froms, dtype=np.int32)
        n = len(froms)
        dst = np.full((n, dst_len), fill, dtype=np.uint16)
        out_len = np.zeros(n, dtype=np.int32)
        status = np.zeros(n, dtype=np.int32)
        aux = np.zeros(n, dtype=np.int32)
        b = boundary if isinstance(boundary, (int, np.integer)) else ord(boundary)
        lib().orc_fm_extract_until_boundary_batch(self.h, int(mode), froms.ctypes.data, n, int(b), dst.ctypes.data,
                                                  int(dst_len), int(offset), out_len.ctypes.data, status.ctypes.data,
                                                  aux.ctypes.data, threads)
        return dst, out_len, status, aux

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_fm_free(self.h)
            self.h = None


def counters_reset():
    lib().orc_counters_reset()


def counters():
    c = Counters()
    lib().orc_counters_get(C.byref(c))
    return {k: getattr(c, k) for k, _ in Counters._fields_}
