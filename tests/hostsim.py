"""ctypes wrapper of tests/libhostsim.so: the device functions of index4j_amd/csrc/fmx_device.hpp compiled
for the host (test-only; see hostsim.cpp).  Lets the CPU suite check the exact device source against
the oracle.  Not a product path."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def lib(compact=False):
    """compact = the device header compiled with -DFMX_COMPACT=1: the bv_* functions decode RRR records (what the kernels of
    namespace fmxc run over COMPACT images, option image_compact)"""
    if compact not in _LIBS:
        so = os.path.join(_HERE, "libhostsim_compact.so" if compact else "libhostsim.so")
        srcs = [os.path.join(_HERE, "hostsim.cpp"),
                os.path.join(_HERE, "..", "index4j_amd", "csrc", "fmx_device.hpp"),
                os.path.join(_HERE, "..", "index4j_amd", "csrc", "fmx_blob.hpp")]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.check_call(["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared"] + (["-DFMX_COMPACT=1"] if compact else []) +
                                  ["-o", so, srcs[0]])
        L = C.CDLL(so)
        L.sim_wt_rank.argtypes = [C.c_void_p, C.c_uint32, C.c_int32, C.c_void_p]
        L.sim_wt_inverse_select.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.sim_win_attach.argtypes = [C.c_void_p, C.c_void_p]
        L.sim_win_attach.restype = C.c_int64
        L.sim_win_detach.argtypes = [C.c_void_p]
        L.sim_set_pack.argtypes = [C.c_int]
        L.sim_set_entry_bytes.argtypes = [C.c_int]
        _LIBS[compact] = L
    return _LIBS[compact]


def reference_route_index(text, sample_rate, enable_extract=True):
    """an index whose image has NO fast mapping entries: every present entry says 'take the reference's own route',
    so rank() reads the block header and the leaf entry, fixes clamped entries up and rebuilds the canonical code
    as WFBB:1119-1156 do, and NO inverseSelect node records: every block's InvHdr says the same, so inverseSelect
    walks block header, level table and cumulative counts as WFBB:1305-1537 do.
    (The image is flattened under options map_fast = 0, inv_fast = 0.)"""
    import index4j_amd as ia

    assert ia.lib.fmx_set_option(b"map_fast", 0) == 0
    assert ia.lib.fmx_set_option(b"inv_fast", 0) == 0
    try:
        f = ia.FmIndex(text, sample_rate, enable_extract, device=None)
        f.blob()  # flatten now, under the options
    finally:
        ia.lib.fmx_set_option(b"map_fast", 1)
        ia.lib.fmx_set_option(b"inv_fast", 1)
    return f


def inverse_select_block_kinds(blob):
    """(blocks with node records, run blocks, blocks on the reference's route) of an FM-index image — read from
    the InvHdr arrays (index4j_amd/csrc/fmx_blob.hpp)"""
    b = np.frombuffer(blob, np.uint8) if not isinstance(blob, np.ndarray) else blob
    u32 = lambda off: int(b[off:off + 4].view(np.uint32)[0])
    i32 = lambda off: int(b[off:off + 4].view(np.int32)[0])
    n_sb = i32(8 + 8 + 9 * 4)            # BlobHeader.n_sb
    off_sbdesc = u32(8 + 8 + 12 * 4 + 8 + 6 * 4) << 3
    tree = run = slow = 0
    for s in range(n_sb):
        d = off_sbdesc + 64 * s
        n_blocks = i32(d + 16)
        inv = u32(d + 32 + 4) << 3       # SbDesc.rrr.off_bits
        x = b[inv:inv + 16 * n_blocks].view(np.uint32)[0::4]
        run += int(((x & 0x80000000) != 0).sum())
        slow += int(((x & 0x20000000) != 0).sum())
        tree += int(((x & 0xA0000000) == 0).sum())
    return tree, run, slow


class HostSim:
    """runs the device code over the host blob of an index4j_amd.FmIndex"""

    def __init__(self, fm_index):
        self.fm = fm_index  # keeps the blob alive
        self.blob = fm_index.blob()
        self.p = self.blob.ctypes.data
        # BlobHeader.compact (byte 152): the image's bit vectors are RRR records — the simulation compiled for that form
        self.compact = bool(int(np.frombuffer(self.blob, np.uint8)[152:156].view(np.int32)[0]))
        self.L = lib(self.compact)
        self.windows = None

    def attach_windows(self, entry_bytes=0):
        """grows the window directory (fmx_device.hpp: win_build_cell, what k_win_build runs when an index becomes resident) and
        makes every later call of this simulation take it first, as the kernels do; returns (positions with a class, positions,
        classes in use, positions with an entry, entries that carry a status or `suspect`).  entry_bytes: 0 = the form fmx_to_device
        picks by the alphabet (four-byte entries where cumulativeCounts fit LDS), 4 / 6 = that form"""
        stats = np.zeros(6, np.int64)
        self.L.sim_set_entry_bytes(int(entry_bytes))
        try:
            self.L.sim_win_attach(C.c_void_p(self.p), C.c_void_p(stats.ctypes.data))
        finally:
            self.L.sim_set_entry_bytes(0)
        self.windows = tuple(int(v) for v in stats[:5])
        self.window_slots = int(stats[5])  # four-byte entries: how many point at an eight-byte slot (-1: six-byte entries)
        return self.windows

    def detach_windows(self):
        self.L.sim_win_detach(C.c_void_p(self.p))
        self.windows = None

    def __del__(self):
        try:
            if self.windows is not None:
                self.L.sim_win_detach(C.c_void_p(self.p))
        except Exception:
            pass

    def wt_rank_batch(self, positions, symbols):
        st = np.zeros(1, np.int32)
        out = np.zeros(len(positions), np.int64)
        sts = np.zeros(len(positions), np.int32)
        for i, (p_, s_) in enumerate(zip(positions, symbols)):
            st[0] = 0
            out[i] = self.L.sim_wt_rank(self.p, int(p_), int(s_), st.ctypes.data)
            sts[i] = st[0]
        return out, sts

    def wt_rank(self, pos, sym):
        st = np.zeros(1, np.int32)
        return self.L.sim_wt_rank(self.p, pos, sym, st.ctypes.data), int(st[0])

    def wt_inverse_select(self, pos):
        r = np.zeros(1, np.int32)
        c = self.L.sim_wt_inverse_select(self.p, pos, r.ctypes.data)
        return c, int(r[0])

    def lf_step_both(self, row):
        out = np.zeros(6, np.int32)
        self.L.sim_lf_step_both(C.c_void_p(self.p), int(row), C.c_void_p(out.ctypes.data))
        return out

    def count_batch(self, chars, offsets):
        chars = np.ascontiguousarray(chars, np.uint16)
        offsets = np.ascontiguousarray(offsets, np.int32)
        n = len(offsets) - 1
        counts, lf, st, rng = (np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(2 * n, np.int32))
        self.L.sim_count(C.c_void_p(self.p), C.c_void_p(chars.ctypes.data), C.c_void_p(offsets.ctypes.data), n,
                        C.c_void_p(counts.ctypes.data), C.c_void_p(lf.ctypes.data), C.c_void_p(st.ctypes.data),
                        C.c_void_p(rng.ctypes.data))
        return counts, st, lf, rng

    def count_batch_with_table(self, table_chars, chars, offsets):
        """count() started from a suffix table of `table_chars` characters (filled and consulted by the device header's
        own functions); returns (counts, status, lf_steps per pattern, LF-steps the table answered, entries)"""
        chars = np.ascontiguousarray(chars, np.uint16)
        offsets = np.ascontiguousarray(offsets, np.int32)
        n = len(offsets) - 1
        counts, lf, st = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
        answered = np.zeros(1, np.int64)
        self.L.sim_count_table.restype = C.c_int64
        entries = self.L.sim_count_table(C.c_void_p(self.p), int(table_chars), C.c_void_p(chars.ctypes.data),
                                        C.c_void_p(offsets.ctypes.data), n, C.c_void_p(counts.ctypes.data),
                                        C.c_void_p(lf.ctypes.data), C.c_void_p(st.ctypes.data), C.c_void_p(answered.ctypes.data))
        return counts, st, lf, int(answered[0]), int(entries)

    def locate_batch(self, chars, offsets, max_matches, loc_cap):
        counts, st, lf, rng = self.count_batch(chars, offsets)
        n = len(counts)
        locs = np.zeros((n, max(loc_cap, 0)), np.int32)
        found = np.zeros(n, np.int32)
        self.L.sim_locate_walk(C.c_void_p(self.p), C.c_void_p(rng.ctypes.data), n, max_matches,
                              C.c_void_p(locs.ctypes.data), loc_cap, C.c_void_p(found.ctypes.data),
                              C.c_void_p(lf.ctypes.data), C.c_void_p(st.ctypes.data))
        return locs, found, st, lf

    def extract_batch(self, starts, stops, dst_len, offset=0, dst=None):
        starts = np.ascontiguousarray(starts, np.int32)
        stops = np.ascontiguousarray(stops, np.int32)
        n = len(starts)
        if dst is None:
            dst = np.zeros((n, dst_len), np.uint16)
        out_len, lf, st = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
        self.L.sim_extract(C.c_void_p(self.p), C.c_void_p(starts.ctypes.data), C.c_void_p(stops.ctypes.data), n,
                          C.c_void_p(dst.ctypes.data), dst_len, offset, C.c_void_p(out_len.ctypes.data),
                          C.c_void_p(lf.ctypes.data), C.c_void_p(st.ctypes.data))
        return dst, out_len, st, lf

    def extract_boundary_batch(self, froms, boundary, mode, dst_len, offset=0, dst=None, accelerate=1):
        froms = np.ascontiguousarray(froms, np.int32)
        n = len(froms)
        if dst is None:
            dst = np.zeros((n, dst_len), np.uint16)
        out_len, lf, st, aux = (np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32))
        b = boundary if isinstance(boundary, (int, np.integer)) else ord(boundary)
        self.L.sim_extract_boundary(C.c_void_p(self.p), C.c_void_p(froms.ctypes.data), n, C.c_uint16(b), mode,
                                   C.c_void_p(dst.ctypes.data), dst_len, offset, C.c_void_p(out_len.ctypes.data),
                                   C.c_void_p(lf.ctypes.data), C.c_void_p(st.ctypes.data), C.c_void_p(aux.ctypes.data), accelerate)
        return dst, out_len, st, aux, lf
