#!/usr/bin/env python3
"""Differential fuzzing of the GPU path against the oracle: random texts (alphabet 2..3000 symbols, runs, embedded
sentinels, lengths that cross the 2^20 superblock boundary now and then), random sample rates, all query kinds
through tests/parity_checks.check_all.  usage: python tools/fuzz_gpu.py [--seconds 240] [--seed 1]"""
import argparse
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def random_text(rnd, big=False):
    kind = rnd.randrange(6)
    sigma = rnd.choice([2, 3, 5, 17, 64, 200, 257, 700, 3000])
    n = rnd.choice([1, 7, 100, 5000, 40_000, 150_000, 300_000]) if rnd.random() < 0.9 else (1 << 20) + rnd.randrange(-3, 70_000)
    if big:  # two to three superblocks; alphabets that change along the text (symbols absent from whole superblocks)
        n = rnd.choice([(1 << 20) - 1, 1 << 20, (1 << 20) + 1, (1 << 21) - 1, (1 << 21) + rnd.randrange(0, 70_000), (1 << 20) + 300_000])
        kind = rnd.choice([0, 2, 4, 4, 5])
    rng = np.random.default_rng(rnd.randrange(1 << 30))
    if kind == 0:
        a = rng.integers(0, sigma, n)
    elif kind == 1:  # skewed
        a = np.minimum((rng.exponential(sigma / 8.0, n)).astype(np.int64), sigma - 1)
    elif kind == 2:  # long runs
        a = np.repeat(rng.integers(0, sigma, n // 50 + 1), rng.integers(1, 100, n // 50 + 1))[:n]
    elif kind == 3:  # periodic
        p = rng.integers(0, sigma, rnd.randrange(1, 40))
        a = np.tile(p, n // len(p) + 1)[:n]
    elif kind == 4:  # blocks with different alphabets
        seg = 3000 if not big else rnd.choice([3000, 200_000, 700_000])
        parts = [rng.integers(lo, lo + max(2, sigma // 8), seg) for lo in rng.integers(0, max(1, sigma - sigma // 8), n // seg + 1)]
        a = np.concatenate(parts)[:n]
    else:  # text-like with line breaks
        a = rng.integers(0, sigma, n)
        a[rng.random(n) < 0.02] = 0
    if len(a) == 0:
        a = np.array([32], np.int64)
    a = (a.astype(np.int64) + 33).astype(np.uint16)
    n = len(a)
    a[a == 43] = 10  # some newlines
    if rnd.random() < 0.2 and n > 3:
        a[rng.integers(0, n, max(1, n // 500))] = 0  # embedded sentinels
    return a


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--big", action="store_true", help="texts of two to three superblocks with shifting alphabets")
    ap.add_argument("--only-case", type=int, default=-1,
                    help="replay ONE case of a seed: the cases before it only draw their random numbers (a FAILED line names seed and case)")
    args = ap.parse_args()
    import index4j_amd as ia
    from parity_checks import GpuEngine, check_all

    rnd = random.Random(args.seed)
    t0 = time.time()
    cases = 0
    while time.time() - t0 < args.seconds:
        text = random_text(rnd, args.big)
        sr = rnd.choice([1, 2, 3, 8, 16, 32, 64, 100])
        layout = rnd.choice([-1, 0, 1])
        cache = rnd.choice([320, 0])
        opts = {"map_by_symbol": layout, "sb_cache_limit": cache,
                "map_fast": rnd.choice([1, 1, 1, 0]),  # 0: every mapping entry on the reference's route
                "inv_fast": rnd.choice([1, 1, 1, 0]),  # 0: inverseSelect on the reference's route
                "suffix_table_mb": rnd.choice([256, 256, 1, 0]),  # budget of the suffix table (0: none)
                "suffix_table_chars": rnd.choice([4, 4, 2, 3, 6, 8]),  # its depth
                "suffix_table_image_fraction": rnd.choice([8, 0, 0, 2]),  # ... and its size against the image's
                "cells_split_blocks": rnd.choice([1 << 20, 64, 256]),  # chunked decoding of the bit vectors
                "boundary_group": rnd.choice([0, 1, 2, 4, 8]),
                "image_compact": rnd.choice([0, 0, 1]),  # bit vectors as RRR records, kernels of namespace fmxc
                "plan_min_per_string": rnd.choice([0, 0, 16]),  # 0: every batch of sort_min patterns is planned; 16: the policy
                "plan_sa_min": rnd.choice([0, 0, 786432]),  # ... with the SA-row order: planned from this many patterns on
                "plan_sa_key": rnd.choice([2, 2, 1, 0]),
                "walk_order_min": rnd.choice([1, 1, 0, 32768]),  # locate: hits walked by the first row of the ranges from this batch size on
                "walk_fine": rnd.choice([1, 1, 0]),
                "boundary_order_min": rnd.choice([1, 1, 0, 32768]),  # extractUntilBoundary: queries by text position from this batch size on
                "plan_fused": rnd.choice([0, 0, 1]),  # the plan stage as one launch (round 5) ...
                "plan_spin_limit": rnd.choice([4096, 4096, 0]),  # ... whose barrier gives up at once: records in the caller's order
                "boundary_first_fill": rnd.choice([2, 2, 0]),  # extractUntilBoundary: a lane's two walks interleaved / one after the other
                "regroup_by_length": rnd.choice([1, 1, 0]),  # k_count: workgroups with mixed pattern lengths regroup by length
                # round 6
                "code_bits_12": rnd.choice([1, 1, 0]),  # alphabets of 257..4,096 codes: five codes per plan word / 16-bit codes
                "count_lean": rnd.choice([0, 0, 1]),  # k_count_lean + the list pass (fast routes inlined, everything else on a redo list)
                "host_small_max": rnd.choice([2048, 2048, 0, 40]),  # host-array calls through one mapped pinned block up to this many queries
                "window_entry_bytes": rnd.choice([0, 0, 4, 6]),  # the directory's entries: by the alphabet / the row alone / row + symbol
                "window_cells": rnd.choice([2, 2, 1, 0, 3]),  # ... 3: the flat form (a word per position); 2: the rule picks (texts of this size: flat)
                "window_flat_fraction": rnd.choice([128, 0]),  # ... unless the rule may not (0): the cells' form  # the window directory: by the memory rule / always / never
                "walk_queue": rnd.choice([8, 8, 4, 0]),  # locate: tickets per lane of the per-wave queue (0: the packed form)
                "walk_queue_min_slots": rnd.choice([32, 1, 1, 8]),  # ... from this many hit slots per pattern on
                "walk_burst": rnd.choice([0, 0, 1, 3, 8]),  # ... steps between two hand-outs (0: sampleRate / 4)
                "boundary_rounds": rnd.choice([1, 1, 0]),  # extractUntilBoundary: the four intervals next to `from` first, the rest for who needs them
                "boundary_narrow": rnd.choice([0, 0, 1]),  # extractUntilBoundary: a narrow first round + the wide form over the list
                "boundary_narrow_min": rnd.choice([4096, 1, 1])}
        check_seed = rnd.randrange(1 << 30)
        if args.only_case >= 0 and cases != args.only_case:
            if cases % 3 == 1:
                rnd.random()  # (the draw of the device-construction check below)
            cases += 1
            if cases > args.only_case:
                break
            continue
        for k, v in opts.items():
            assert ia.lib.fmx_set_option(k.encode(), v) == 0, (k, v)
        try:
            check_all(lambda t, s: GpuEngine(t, s), text, sr, random.Random(check_seed), n_q=60)
            if cases % 3 == 1:  # the builder with its suffix-array stage on the GPU gives the same bytes
                extract = rnd.random() < 0.7
                a = ia.FmIndex(text, sr, extract, device=None).write(False)
                b = ia.FmIndex(text, sr, extract, device=None, build_device=0).write(False)
                assert a == b, "device construction differs"
            if cases % 6 == 0 and len(text) >= 2000:  # a batch large enough for the planned path of count / locate
                import orc

                fm = ia.FmIndex(text, sr, True, device=0)
                o = orc.OracleFmIndex(text, sr, True)
                t16 = ia.as_chars(text)
                r2 = random.Random(cases)
                pats = [t16[p_:p_ + r2.randrange(1, 22)] for p_ in (r2.randrange(len(t16) - 1) for _ in range(20000))]
                for k in range(0, 20000, 97):
                    pats[k] = pats[k].copy()
                    pats[k][r2.randrange(len(pats[k]))] = 7  # absent character
                ch, off = ia.pack_patterns(pats)
                cnt, st, lf = fm.count_batch(ch, off, want_steps=True)
                oc, ost = o.count_batch(ch, off, threads=8)
                if not ((cnt == oc).all() and (st == ost).all()):
                    bad = np.flatnonzero((cnt != oc) | (st != ost))
                    print("planned count batch: %d of %d patterns differ; first: %r" % (
                        len(bad), len(cnt), [(int(i), pats[i].tolist(), int(cnt[i]), int(oc[i]), int(st[i]), int(ost[i])) for i in bad[:5]]),
                        "suffix table", fm.suffix_table_info(), flush=True)
                assert (cnt == oc).all() and (st == ost).all(), "planned count batch"
                locs, found, st2 = fm.locate_batch(ch, off, 3)
                for k in range(0, 20000, 211):
                    try:
                        kk, ll = o.locate(pats[k], max_matches=3, cap=3)
                        assert st2[k] == 0 and found[k] == kk and (locs[k, :kk] == ll).all(), "planned locate batch"
                    except IndexError:  # rank(size) with size % 2^20 == 0: the JVM raises AIOOBE (Q3)
                        assert st2[k] == 9, "planned locate batch: AIOOBE expected"
        except Exception:
            print("FAILED: seed %d case %d len %d sr %d options %r" % (args.seed, cases, len(text), sr, opts), flush=True)
            raise
        cases += 1
        if cases % 25 == 0:
            print("... %d cases, %.0f s" % (cases, time.time() - t0), flush=True)
    print("fuzz ok: %d cases in %.0f s (seed %d)" % (cases, time.time() - t0, args.seed))


if __name__ == "__main__":
    main()
