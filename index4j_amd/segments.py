"""A text beyond one FmIndex as a set of segment indexes.

FmIndex addresses its text with Java ints (``length`` FM:131, RrrVector positions RRR:358), so a text of
2^31 chars or more (BASELINE configs[4]: 2 GiB) cannot be one index.  What a user of the reference does is
what this class does: cut the text at record boundaries into pieces of at most ``segment_chars``, build
one FmIndex per piece, sum ``count`` over the pieces and add each piece's start to its ``locate`` results.
All pieces live on one GPU and are queried with one call (``fmx_count_segments`` / ``fmx_locate_segments``).
Occurrences that span a cut are not occurrences in any piece — exactly as with K Java objects."""
import ctypes as C

import numpy as np

from ._lib import check, lib
from .fmindex import FmIndex, as_chars


def cut_points(text, segment_chars, boundary):
    """starts of the pieces: each piece ends right after the last `boundary` char that keeps it within
    segment_chars (or at segment_chars if it holds no boundary at all)"""
    t = as_chars(text)
    n = len(t)
    b = boundary if isinstance(boundary, (int, np.integer)) else ord(boundary)
    starts = [0]
    while n - starts[-1] > segment_chars:
        lo = starts[-1]
        window = t[lo:lo + segment_chars]
        hits = np.flatnonzero(window == b)
        starts.append(lo + (int(hits[-1]) + 1 if len(hits) else segment_chars))
    return starts


class SegmentedFmIndex:
    def __init__(self, text=None, sampleRate=32, enableExtract=True, device=0, segment_chars=1 << 28, boundary="\n",
                 _segments=None, _bases=None):
        if _segments is not None:
            self.segments, self.bases = list(_segments), [int(b) for b in _bases]
        else:
            t = as_chars(text)
            starts = cut_points(t, segment_chars, boundary)
            ends = starts[1:] + [len(t)]
            self.bases = starts
            self.segments = [FmIndex(t[a:b], sampleRate, enableExtract, device=device) for a, b in zip(starts, ends)]
        self._handles = (C.c_void_p * len(self.segments))(*[s.handle for s in self.segments])
        self._bases = np.asarray(self.bases, dtype=np.int64)

    @classmethod
    def from_segments(cls, segments, bases):
        """existing FmIndex objects (e.g. FmIndex.read of K serialized indexes) + the text offset of each"""
        return cls(_segments=segments, _bases=bases)

    def __len__(self):
        return len(self.segments)

    @property
    def handles(self):
        return self._handles

    @property
    def base_array(self):
        return self._bases

    def getInputLength(self):
        """sum of the pieces' lengths, each including its own sentinel (FM:929)"""
        return sum(s.getInputLength() for s in self.segments)

    def count_batch(self, chars, offsets, want_steps=False):
        chars = np.ascontiguousarray(chars, dtype=np.uint16)
        offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        n = len(offsets) - 1
        counts = np.zeros(n, dtype=np.int64)
        steps = np.zeros(n, dtype=np.int64)
        status = np.zeros(n, dtype=np.int32)
        check(lib.fmx_count_segments(self._handles, len(self.segments), chars.ctypes.data, offsets.ctypes.data, n,
                                     counts.ctypes.data, steps.ctypes.data, status.ctypes.data), "fmx_count_segments")
        return (counts, status, steps) if want_steps else (counts, status)

    def locate_batch(self, chars, offsets, max_matches, fill=-1):
        chars = np.ascontiguousarray(chars, dtype=np.uint16)
        offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        n = len(offsets) - 1
        locs = np.full((n, int(max_matches)), fill, dtype=np.int64)
        found = np.zeros(n, dtype=np.int32)
        status = np.zeros(n, dtype=np.int32)
        check(lib.fmx_locate_segments(self._handles, len(self.segments), self._bases.ctypes.data, chars.ctypes.data,
                                      offsets.ctypes.data, n, int(max_matches), locs.ctypes.data, found.ctypes.data,
                                      status.ctypes.data), "fmx_locate_segments")
        return locs, found, status
