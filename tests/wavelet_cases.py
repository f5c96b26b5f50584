"""Sequences for the stand-alone wavelet tests, shared by the CPU (host simulation) and GPU suites."""
import numpy as np


def quirk_sequence():
    """128 x ([4096 mixed symbols without 50] + [4096 x 50]): the block-size search picks 2^12 blocks, so every
    rank(pos, 50) with pos inside a mixed block takes the next-block path onto a RUN block — the reference then
    reads (treeHeight-1)*4 = -4 bytes early (WFBB:1081) and returns garbage, which must be reproduced."""
    rng = np.random.default_rng(1)
    parts = []
    for _ in range(128):
        parts.append(rng.integers(1, 40, 4096))
        parts.append(np.full(4096, 50))
    return np.concatenate(parts).astype(np.int16)


def probes(seq, rng, n=3000):
    L = len(seq)
    pos = np.concatenate([rng.integers(0, L + 1, n), [0, 1, L - 1, L, L + 7, 4095, 4096, 4097, 8191, 8192]])
    sym = np.concatenate([seq[rng.integers(0, L, n)], [50, 50, 50, 50, 50, 50, 50, 50, 3, 3]])
    return pos.astype(np.int64), sym.astype(np.int32)
