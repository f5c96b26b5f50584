"""Host-side mirror of index4j's public API for the backward-search path, over the C ABI.

``FmIndexBuilder`` mirrors fm/FmIndexBuilder.java:21-62 and ``FmIndex`` mirrors the query surface of
fm/FmIndex.java:443-941 (same method names, argument meaning and exceptions, with Java's
RuntimeException -> RuntimeError, IllegalArgumentException -> ValueError,
ArrayIndexOutOfBoundsException -> IndexError, IOException -> IOError).  Scalar calls are batches of
one; the ``*_batch`` methods are what a service would use.  All queries run on the GPU through
libfmx.so — there is no CPU query path.

In production the host language is Java (see INTEGRATION.md and bindings/java); this module is the
same binding written in Python for the test-suite and the benchmark.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib

_EXC = {0: RuntimeError, 1: ValueError, 2: IndexError}


def as_chars(text):
    """str / array -> contiguous uint16 array of UTF-16 code units (a Java char[])."""
    if isinstance(text, str):
        return np.frombuffer(text.encode("utf-16-le"), dtype=np.uint16).copy()
    if isinstance(text, (bytes, bytearray)):
        raise TypeError("pass str or a uint16 array; use FmIndex.convertBytePatternToCharPattern for UTF-8 bytes")
    return np.ascontiguousarray(text, dtype=np.uint16)


def chars_to_str(arr):
    return np.ascontiguousarray(arr, dtype=np.uint16).tobytes().decode("utf-16-le", errors="surrogatepass")


def raise_for_status(status, aux=0):
    """Re-throw the reference's exception for a per-query status code (same type, same message)."""
    if status == _lib.ST_OK:
        return
    msg = lib.fmx_status_message(int(status)).decode()
    if "%d" in msg:
        msg = msg % aux
    raise _EXC[lib.fmx_status_kind(int(status))](msg)


def pack_patterns(patterns):
    """list of str / uint16 arrays -> (chars, offsets) in the layout fmx_count_batch takes."""
    arrs = [as_chars(p) for p in patterns]
    off = np.zeros(len(arrs) + 1, dtype=np.int32)
    if arrs:
        np.cumsum([len(a) for a in arrs], out=off[1:])
    chars = np.concatenate(arrs) if arrs else np.zeros(0, dtype=np.uint16)
    return np.ascontiguousarray(chars, dtype=np.uint16), off


class FmIndexBuilder:
    """fm/FmIndexBuilder.java: defaults sampleRate=32, enableExtraction=true (FMB:21-22)."""

    def __init__(self):
        self._sample_rate = 32
        self._enable_extraction = True
        self._build_device = None

    def setBuildDevice(self, device):
        """extension: run the suffix-array stage of the constructor on this GPU (same index, byte for byte)"""
        self._build_device = device
        return self

    def setSampleRate(self, sample_rate):  # FMB:34-37
        self._sample_rate = int(sample_rate)
        return self

    def setEnableExtraction(self, enable):  # FMB:46-49
        self._enable_extraction = bool(enable)
        return self

    def build(self, text, device=0):  # FMB:59-61
        return FmIndex(text, self._sample_rate, self._enable_extraction, device=device, build_device=self._build_device)


class FmIndex:
    """fm/FmIndex.java query surface.  `device=None` keeps the index on the host (build / save /
    load only); any query then fails loudly."""

    def __init__(self, text=None, sampleRate=32, enableExtract=True, device=0, _handle=None, build_device=None):
        self._h = None
        self.build_stats = None
        if _handle is not None:
            self._h = _handle
        else:
            a = as_chars(text)
            h = C.c_void_p()
            if build_device is None:
                check(lib.fmx_build(a.ctypes.data, len(a), int(sampleRate), int(bool(enableExtract)), C.byref(h)), "fmx_build")
            else:  # suffix array, BWT and samples computed in HBM (fmx_sa_gpu.hip); same index
                rounds, rows, secs = C.c_int32(0), C.c_int64(0), C.c_double(0)
                check(lib.fmx_build_on_device(a.ctypes.data, len(a), int(sampleRate), int(bool(enableExtract)),
                                              int(build_device), C.byref(h), C.byref(rounds), C.byref(rows),
                                              C.byref(secs)), "fmx_build_on_device")
                self.build_stats = {"doubling_rounds": rounds.value, "rows_sorted": rows.value,
                                    "device_stage_seconds": secs.value,
                                    # 0.0: the wavelet tree was encoded on the host (alphabet above 1,024 codes, or option)
                                    "wavelet_device_seconds": lib.fmx_build_wavelet_seconds(h)}
            self._h = h
        if device is not None:
            self.to_device(device)

    # ---- persistence (FM:948-1025 through SER:67-100) ----
    @classmethod
    def read(cls, data, device=0):
        buf = np.frombuffer(bytes(data), dtype=np.uint8)
        h = C.c_void_p()
        check(lib.fmx_load(buf.ctypes.data, len(buf), C.byref(h)), "fmx_load")
        return cls(device=device, _handle=h)

    def write(self, framed=True):
        buf, n = C.c_void_p(), C.c_size_t()
        check(lib.fmx_save(self._h, int(framed), C.byref(buf), C.byref(n)), "fmx_save")
        try:
            # (ctypes.string_at takes a C int: a sampleRate-1 index of 256 MiB serializes to more than 2 GiB)
            return bytes((C.c_ubyte * n.value).from_address(buf.value))
        finally:
            lib.fmx_free_buffer(buf)

    def serialized_key_order_is_modelled(self):
        """False: a JVM's HashMap would have made one of the character map's buckets a tree bin, whose iteration order write()
        does not model (fmx.h fmx_save_key_order_modelled): the stream is valid, its key order in that bucket unverified"""
        rc = lib.fmx_save_key_order_modelled(self._h)
        if rc < 0:
            check(rc, "fmx_save_key_order_modelled")
        return rc == 1

    @classmethod
    def attach_device_blob(cls, device_ptr, nbytes, device):
        """adopt a blob already in HBM (e.g. received through an RCCL broadcast)"""
        h = C.c_void_p()
        check(lib.fmx_attach_device_blob(C.c_void_p(device_ptr), nbytes, device, C.byref(h)), "fmx_attach_device_blob")
        return cls(device=None, _handle=h)

    def to_device(self, device=0):
        check(lib.fmx_to_device(self._h, int(device)), "fmx_to_device")
        return self

    def blob(self):
        p, n = C.c_void_p(), C.c_size_t()
        check(lib.fmx_blob(self._h, C.byref(p), C.byref(n)), "fmx_blob")
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n.value,))

    def device_blob(self):
        n = C.c_size_t()
        p = lib.fmx_device_blob(self._h, C.byref(n))
        return p, n.value

    @property
    def handle(self):
        return self._h

    def close(self):
        if self._h is not None:
            lib.fmx_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- convenience (FM:929-941, 1044-1046) ----
    def getInputLength(self):
        return lib.fmx_input_length(self._h)

    def getAlphabetLength(self):
        return lib.fmx_alphabet_length(self._h)

    def __str__(self):
        return "FMIndex-sampleRate:%d-extract:%s" % (
            lib.fmx_sample_rate(self._h),
            "true" if lib.fmx_extract_enabled(self._h) else "false",
        )

    toString = __str__

    @staticmethod
    def convertBytePatternToCharPattern(pattern, offset, length, destination):  # FM:239-298
        src = np.frombuffer(bytes(pattern), dtype=np.uint8)
        bad = C.c_int32(0)
        n = lib.fmx_convert_byte_pattern(src.ctypes.data, offset, length, destination.ctypes.data, C.byref(bad))
        if n < 0:
            raise RuntimeError("Found a character that exceeds (32767): it was %d" % bad.value)
        return n

    def window_cells_bytes(self):
        """bytes of the resident index's window directory (0: none) — fmx_window_cells_info"""
        nbytes = C.c_int64(0)
        check(lib.fmx_window_cells_info(self._h, C.byref(nbytes)), "fmx_window_cells_info")
        return nbytes.value

    def suffix_table_info(self):
        """(characters, bytes) of the resident index's suffix table (0, 0: none) — fmx_suffix_table_info"""
        chars, nbytes = C.c_int32(0), C.c_int64(0)
        check(lib.fmx_suffix_table_info(self._h, C.byref(chars), C.byref(nbytes)), "fmx_suffix_table_info")
        return chars.value, nbytes.value

    # ---- batched queries ----
    def count_batch(self, chars, offsets, want_steps=False):
        chars = np.ascontiguousarray(chars, dtype=np.uint16)
        offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        n = len(offsets) - 1
        counts = np.zeros(n, dtype=np.int32)
        steps = np.zeros(n, dtype=np.int32)
        status = np.zeros(n, dtype=np.int32)
        check(lib.fmx_count_batch(self._h, chars.ctypes.data, offsets.ctypes.data, n, counts.ctypes.data,
                                  steps.ctypes.data, status.ctypes.data), "fmx_count_batch")
        return (counts, status, steps) if want_steps else (counts, status)

    def locate_batch(self, chars, offsets, max_matches, loc_cap=None, want_steps=False, locs=None):
        chars = np.ascontiguousarray(chars, dtype=np.uint16)
        offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        n = len(offsets) - 1
        if loc_cap is None:
            loc_cap = max_matches
        if locs is None:
            locs = np.zeros((n, max(loc_cap, 0)), dtype=np.int32)
        found = np.zeros(n, dtype=np.int32)
        steps = np.zeros(n, dtype=np.int32)
        status = np.zeros(n, dtype=np.int32)
        check(lib.fmx_locate_batch(self._h, chars.ctypes.data, offsets.ctypes.data, n, int(max_matches),
                                   locs.ctypes.data, int(loc_cap), found.ctypes.data, steps.ctypes.data,
                                   status.ctypes.data), "fmx_locate_batch")
        return (locs, found, status, steps) if want_steps else (locs, found, status)

    def extract_batch(self, starts, stops, dst_len, offset=0, dst=None, want_steps=False):
        starts = np.ascontiguousarray(starts, dtype=np.int32)
        stops = np.ascontiguousarray(stops, dtype=np.int32)
        n = len(starts)
        if dst is None:
            dst = np.zeros((n, dst_len), dtype=np.uint16)
        out_len = np.zeros(n, dtype=np.int32)
        steps = np.zeros(n, dtype=np.int32)
        status = np.zeros(n, dtype=np.int32)
        check(lib.fmx_extract_batch(self._h, starts.ctypes.data, stops.ctypes.data, n, dst.ctypes.data, int(dst_len),
                                    int(offset), out_len.ctypes.data, steps.ctypes.data, status.ctypes.data),
              "fmx_extract_batch")
        return (dst, out_len, status, steps) if want_steps else (dst, out_len, status)

    def extract_boundary_batch(self, froms, boundary, mode, dst_len, offset=0, dst=None, want_steps=False):
        froms = np.ascontiguousarray(froms, dtype=np.int32)
        n = len(froms)
        if dst is None:
            dst = np.zeros((n, dst_len), dtype=np.uint16)
        out_len = np.zeros(n, dtype=np.int32)
        steps = np.zeros(n, dtype=np.int32)
        status = np.zeros(n, dtype=np.int32)
        aux = np.zeros(n, dtype=np.int32)
        b = boundary if isinstance(boundary, (int, np.integer)) else ord(boundary)
        check(lib.fmx_extract_boundary_batch(self._h, froms.ctypes.data, n, int(b), int(mode), dst.ctypes.data,
                                             int(dst_len), int(offset), out_len.ctypes.data, steps.ctypes.data,
                                             status.ctypes.data, aux.ctypes.data), "fmx_extract_boundary_batch")
        return (dst, out_len, status, aux, steps) if want_steps else (dst, out_len, status, aux)

    # ---- locate -> extract pipelines (hits stay in HBM between the stages) ----
    def _pipeline(self, chars, offsets, max_matches, row_len, boundary, mode, fill):
        chars = np.ascontiguousarray(chars, dtype=np.uint16)
        offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        n = len(offsets) - 1
        mm = int(max_matches)
        out = {
            "locs": np.full((n, mm), -1, dtype=np.int32), "found": np.zeros(n, dtype=np.int32),
            "dst": np.full((n, mm, row_len), fill, dtype=np.uint16), "out_len": np.full((n, mm), -1, dtype=np.int32),
            "steps": np.zeros(n, dtype=np.int32), "status": np.zeros(n, dtype=np.int32),
            "hit_status": np.zeros((n, mm), dtype=np.int32), "hit_aux": np.zeros((n, mm), dtype=np.int32),
        }
        o = out
        if mode < 0:
            check(lib.fmx_locate_extract_batch(self._h, chars.ctypes.data, offsets.ctypes.data, n, mm, int(row_len),
                                               o["locs"].ctypes.data, o["found"].ctypes.data, o["dst"].ctypes.data,
                                               o["out_len"].ctypes.data, o["steps"].ctypes.data, o["status"].ctypes.data,
                                               o["hit_status"].ctypes.data), "fmx_locate_extract_batch")
        else:
            b = boundary if isinstance(boundary, (int, np.integer)) else ord(boundary)
            check(lib.fmx_locate_lines_batch(self._h, chars.ctypes.data, offsets.ctypes.data, n, mm, int(b), int(mode),
                                             int(row_len), o["locs"].ctypes.data, o["found"].ctypes.data,
                                             o["dst"].ctypes.data, o["out_len"].ctypes.data, o["steps"].ctypes.data,
                                             o["status"].ctypes.data, o["hit_status"].ctypes.data,
                                             o["hit_aux"].ctypes.data), "fmx_locate_lines_batch")
        return out

    def locate_extract_batch(self, chars, offsets, max_matches, extract_len, fill=0):
        """locate, then extract(loc, min(getInputLength(), loc + extract_len), row, 0) per hit — the reference's
        locateAndExtractBenchmark (FmIndexThroughputBenchmark.java:231-249).  Returns a dict of arrays; slots
        k >= found[i] keep their initial values (locs/out_len -1, rows `fill`)."""
        return self._pipeline(chars, offsets, max_matches, extract_len, 0, -1, fill)

    def locate_lines_batch(self, chars, offsets, max_matches, boundary, dst_len, mode=0, fill=0):
        """locate, then extractUntilBoundary{,Left,Right}(loc, row, 0, boundary) per hit (FM:640-922)"""
        return self._pipeline(chars, offsets, max_matches, dst_len, boundary, mode, fill)

    # ---- scalar API, as in the reference ----
    def count(self, pattern, offset=0, length=None):  # FM:443-474
        p = as_chars(pattern)
        if length is None:
            length = len(p)
        if length <= 0 or offset < 0 or offset + length > len(p):
            raise IndexError("ArrayIndexOutOfBoundsException")  # pattern[i] out of range, FM:456-457
        sub = p[offset:offset + length]
        counts, status = self.count_batch(sub, np.array([0, len(sub)], dtype=np.int32))
        raise_for_status(status[0])
        return int(counts[0])

    def locate(self, pattern, locations, offset=0, length=None, maxMatches=-1):  # FM:487-552
        """`locations` is the caller's int32 array (written in place); returns the number located."""
        p = as_chars(pattern)
        if length is None:
            length = len(p)
        if length <= 0 or offset < 0 or offset + length > len(p):
            raise IndexError("ArrayIndexOutOfBoundsException")
        sub = p[offset:offset + length]
        cap = len(locations)
        rows = np.ascontiguousarray(locations, dtype=np.int32).reshape(1, cap)
        locs, found, status = self.locate_batch(sub, np.array([0, len(sub)], dtype=np.int32), maxMatches, cap, locs=rows)
        locations[:] = locs[0]
        raise_for_status(status[0])
        return int(found[0])

    def extract(self, start, stop, destination, offset=0):  # FM:564-608
        dst = np.ascontiguousarray(destination, dtype=np.uint16).reshape(1, len(destination))
        dst, out_len, status = self.extract_batch([start], [stop], len(destination), offset, dst=dst)
        destination[:] = dst[0]
        raise_for_status(status[0])
        return int(out_len[0])

    def _boundary(self, mode, frm, destination, offset, boundary):
        dst = np.ascontiguousarray(destination, dtype=np.uint16).reshape(1, len(destination))
        dst, out_len, status, aux = self.extract_boundary_batch([frm], boundary, mode, len(destination), offset, dst=dst)
        destination[:] = dst[0]
        raise_for_status(status[0], int(aux[0]))
        return int(out_len[0])

    def extractUntilBoundary(self, frm, destination, offset, boundary):  # FM:640-759
        return self._boundary(0, frm, destination, offset, boundary)

    def extractUntilBoundaryLeft(self, frm, destination, offset, boundary):  # FM:772-831
        return self._boundary(1, frm, destination, offset, boundary)

    def extractUntilBoundaryRight(self, frm, destination, offset, boundary):  # FM:844-922
        return self._boundary(2, frm, destination, offset, boundary)


def synth_log(n, seed=42):
    """deterministic synthetic ASCII log text of exactly n chars"""
    out = np.zeros(n, dtype=np.uint16)
    check(lib.fmx_synth_log(seed, n, out.ctypes.data), "fmx_synth_log")
    return out


def synth_log_multichar(n, symbols=1100, seed=42):
    """the same log with runs of multi-byte characters at word boundaries, `symbols` distinct characters: the shape of the
    reference's fixture HDFS_2k_multichar.log and of the > 1,000-symbol data set its numbers are quoted on"""
    out = np.zeros(n, dtype=np.uint16)
    check(lib.fmx_synth_log_multichar(seed, n, symbols, out.ctypes.data), "fmx_synth_log_multichar")
    return out


def synth_patterns(text, m, count, seed=43):
    """`count` substrings of length m at SplitMix64(seed) % (n - m); returns (chars, offsets, positions)"""
    text = np.ascontiguousarray(text, dtype=np.uint16)
    pat = np.zeros(count * m, dtype=np.uint16)
    off = np.zeros(count + 1, dtype=np.int32)
    pos = np.zeros(count, dtype=np.int32)
    check(lib.fmx_synth_patterns(seed, text.ctypes.data, len(text), m, count, pat.ctypes.data, off.ctypes.data,
                                 pos.ctypes.data), "fmx_synth_patterns")
    return pat, off, pos
