"""shared helpers for the test-suite: fixture text, brute-force definitional oracles (the reference's
own test oracles, util/Util.java:111-258, re-implemented independently), java.util.Random."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")


def hdfs_text():
    """the reference's own test fixture (indices/src/test/resources/HDFS_2k_multichar.log): data, Apache-2.0"""
    return open(os.path.join(GOLDEN, "HDFS_2k_multichar.log"), "rb").read().decode("utf-8")


def occurrences(text, pat):
    """all overlapping occurrence positions, sorted (Util.findExpectedLocationsWithOverlap, T-UTIL:129-139)"""
    out, i = [], text.find(pat)
    while i >= 0:
        out.append(i)
        i = text.find(pat, i + 1)
    return out


def until_boundary(text, seed, b):  # T-UTIL:167-196
    if text[seed] == b:
        return ""
    return until_boundary_left(text, seed, b) + until_boundary_right(text, seed, b)


def until_boundary_left(text, seed, b):  # T-UTIL:209-226
    if text[seed] == b:
        return ""
    i = seed
    while i >= 0 and text[i] != b:
        i -= 1
    return text[i + 1: seed + 1]


def until_boundary_right(text, seed, b):  # T-UTIL:240-258
    if text[seed] == b:
        return ""
    i = seed + 1
    while i < len(text) and text[i] != b:
        i += 1
    return text[seed + 1: i]


def naive_sa_order_hits(text16, pattern16, code_of, limit):
    """positions of the first `limit` occurrences in SUFFIX-ARRAY order under the index's alphabet
    (codes by first appearance, FM:396-435) — pins which hits a truncated locate returns."""
    n = len(text16)
    m = len(pattern16)
    occ = [i for i in range(n - m + 1) if (text16[i:i + m] == pattern16).all()]
    key = lambda i: tuple(code_of[c] for c in text16[i:]) + (0,)
    occ.sort(key=key)
    return occ[:limit]


class JavaRandom:
    """java.util.Random (LCG) with nextInt(bound) / nextInt(origin, bound) as in JDK 17, so the
    reference's seeded test loops (Random(42), T-FM:40) can be replayed"""

    def __init__(self, seed):
        self.seed = (seed ^ 0x5DEECE66D) & ((1 << 48) - 1)

    def next(self, bits):
        self.seed = (self.seed * 0x5DEECE66D + 0xB) & ((1 << 48) - 1)
        v = self.seed >> (48 - bits)
        return v - (1 << bits) if v >= (1 << (bits - 1)) else v

    def next_int(self, a=None, b=None):
        if a is None:
            return self.next(32)
        if b is None:
            bound = a
            r = self.next(31)
            m = bound - 1
            if (bound & m) == 0:
                return (bound * r) >> 31
            u = r
            while True:
                r = u % bound
                if u - r + m < (1 << 31):
                    return r
                u = self.next(31)
        origin, bound = a, b
        r = self.next(32)
        n = bound - origin
        m = n - 1
        if (n & m) == 0:
            return (r & m) + origin
        u = (r & 0xFFFFFFFF) >> 1
        while True:
            r = u % n
            if u + m - r < (1 << 31):
                return r + origin
            u = (self.next(32) & 0xFFFFFFFF) >> 1
