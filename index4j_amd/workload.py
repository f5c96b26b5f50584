"""Synthetic workloads of BASELINE.json's configs, shared by bench.py, the tools and the
full-size parity tests, so that every one of them talks about the same inputs.

  configs[1]  text = synth_log(2^28, seed 42); batch = 8-char substrings at SplitMix64(43) positions
  configs[2]  the first 100,000 patterns of that batch, locate(maxMatches = 16)
  configs[3]  sampleRate-64 index of the same text; from_j = first located position of pattern j
  configs[4]  2 GiB as 8 segment texts synth_log(2^28, seed 42 + s), each cut after its last complete line
              (a Java int cannot address 2^31 chars); batch = substrings drawn from all segments
"""
import numpy as np

from .fmindex import FmIndex, synth_log, synth_patterns
from .segments import SegmentedFmIndex

TEXT_SEED = 42
PATTERN_SEED = 43


def log_text(text_log2, seed=TEXT_SEED):
    return synth_log(1 << text_log2, seed=seed)


def count_batch_patterns(text, n, m=8, seed=PATTERN_SEED):
    """(chars, offsets, positions) of configs[1]'s batch: n substrings of m chars"""
    return synth_patterns(text, m, n, seed=seed)


def segment_texts(n_segments=8, segment_log2=28):
    """the pieces of configs[4]'s text; piece s ends with its last complete line"""
    out = []
    for s in range(n_segments):
        t = synth_log(1 << segment_log2, seed=TEXT_SEED + s)
        out.append(t[: int(np.flatnonzero(t == 10)[-1]) + 1])
    return out


def segment_bases(texts):
    return np.concatenate([[0], np.cumsum([len(t) for t in texts])[:-1]]).astype(np.int64)


def build_segment_set(texts, sample_rate=32, device=0, build_device=0):
    """one FmIndex per piece (suffix-array stage on GPU `build_device`), all resident on `device`"""
    fms = [FmIndex(t, sample_rate, True, device=None, build_device=build_device) for t in texts]
    if device is not None:
        for f in fms:
            f.to_device(device)
    return SegmentedFmIndex.from_segments(fms, segment_bases(texts))


def segment_patterns(texts, n, m=8, seed=PATTERN_SEED):
    """n substrings of m chars, each drawn from a random piece (never across a cut); (chars, offsets)"""
    rng = np.random.default_rng(seed)
    seg_of = rng.integers(0, len(texts), n)
    pat = np.empty((n, m), np.uint16)
    for s, t in enumerate(texts):
        sel = np.flatnonzero(seg_of == s)
        p = rng.integers(0, len(t) - m, len(sel))
        pat[sel] = t[p[:, None] + np.arange(m)[None, :]]
    off = (np.arange(n + 1, dtype=np.int64) * m).astype(np.int32)
    return np.ascontiguousarray(pat.reshape(-1)), off


# ---- the reference's own benchmark shapes -------------------------------------------------------
# FmIndexThroughputState.java:76-83: start uniform in [0, len - maxQueryLength), size uniform in [minQueryLength,
# maxQueryLength) = 8..31 chars, queries are substrings of the indexed text; extractBenchmark extracts
# [start, start + maxQueryLength) (FmIndexThroughputBenchmark.java:222-229).  The data set behind the published numbers
# (loghub Android.log, > 1,000 distinct symbols, README.md:291-292) is not in the repository: the text here is the
# synthetic log with runs of multi-byte characters (fmx_synth_log_multichar), ~1,100 symbols.
REFERENCE_SYMBOLS = 1100
MIN_QUERY, MAX_QUERY = 8, 32


def reference_text(text_log2, symbols=REFERENCE_SYMBOLS, seed=TEXT_SEED):
    from .fmindex import synth_log_multichar

    return synth_log_multichar(1 << text_log2, symbols, seed=seed)


def reference_queries(text, n, seed=42, min_len=MIN_QUERY, max_len=MAX_QUERY):
    """(chars, offsets, starts): n substrings of min_len..max_len-1 chars, the JMH state's query shape"""
    rng = np.random.default_rng(seed)
    starts = rng.integers(0, len(text) - max_len, n).astype(np.int64)
    lens = rng.integers(min_len, max_len, n).astype(np.int64)
    off = np.zeros(n + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    idx = np.repeat(starts - off[:-1], lens) + np.arange(off[-1])
    return np.ascontiguousarray(text[idx], dtype=np.uint16), off.astype(np.int32), starts.astype(np.int32)
