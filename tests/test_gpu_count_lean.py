"""k_count_lean + k_count's list mode (option count_lean = 1, default off: profiles/r06_experiments.txt 1) against k_count and the
oracle.  The lean kernel carries the fast routes of WFBB.rank only (absent from the superblock WFBB:1040-1042, run block
WFBB:1141-1146, next block to the right WFBB:1048-1110 — landing on a run block through the word the flattener left in the entry,
blob v15 — and the walk over cells and path records); a pattern that meets anything else goes on a redo list that k_count's list
mode works off over every route.  Cases: 8-bit and 16-bit code words, patterns longer than a code word (chunks), mixed lengths
(regrouping), no suffix table, every entry forced onto the reference's own route (map_fast = 0: the whole batch takes the list),
both layouts of the mapping rows.  Run with `-m gpu` on an MI355X."""
import random

import numpy as np
import pytest

import index4j_amd as ia
import orc
from common import hdfs_text

pytestmark = pytest.mark.gpu
HD = hdfs_text()


def _batch(t16, rnd, n, lo, hi):
    L = len(t16)
    pats = []
    for i in range(n):
        s = rnd.randrange(L - hi - 1)
        p = t16[s:s + rnd.randrange(lo, hi + 1)].copy()
        if i % 11 == 0:
            p[0] = 0x7a7a if i % 22 else p[0]  # a character the text does not hold, in FRONT (met last)
        pats.append(p)
    ch, off = ia.pack_patterns(pats)
    return ch, np.concatenate([off, [off[-1]]]).astype(np.int32)  # + an empty pattern


@pytest.mark.parametrize("case", ["ascii-short", "ascii-mixed", "multichar", "slow-routes", "by-code", "no-table"])
def test_lean_kernel_equals_k_count_and_the_oracle(case):
    rnd = random.Random(len(case))
    L = ia.lib
    opts = {}
    if case in ("ascii-short", "ascii-mixed", "by-code", "no-table"):
        text = ia.synth_log(1 << 21)
        sr = 16
    else:
        text = ia.as_chars(HD)
        sr = 8
    lo, hi = (8, 8) if case == "ascii-short" else (1, 30)
    if case == "slow-routes":
        opts = {b"map_fast": (0, 1)}
    if case == "by-code":
        opts = {b"map_by_symbol": (0, -1)}
    if case == "multichar":
        opts = {b"map_by_symbol": (1, -1)}
    if case == "no-table":
        opts = {b"suffix_table_mb": (0, 256)}
    try:
        for k, (v, _) in opts.items():
            assert L.fmx_set_option(k, v) == 0
        fm = ia.FmIndex(text, sr, True, device=0)
    finally:
        for k, (_, back) in opts.items():
            L.fmx_set_option(k, back)
    o = orc.OracleFmIndex.read(fm.write(False))
    t16 = ia.as_chars(text)
    ch, off = _batch(t16, rnd, 40_000, lo, hi)
    assert L.fmx_count_batch_is_planned(fm.handle, len(off) - 1) == 1
    orc.counters_reset()
    oc, ost = o.count_batch(ch, off, threads=8)
    steps = orc.counters()["lf_steps"]
    ol, of, ost2 = o.locate_batch(ch, off, 5, threads=8)
    got = {}
    try:
        for lean in (1, 0):
            assert L.fmx_set_option(b"count_lean", lean) == 0
            for rep in range(2):  # twice: the list pass must leave its counters zero
                c, st, lf = fm.count_batch(ch, off, want_steps=True)
                assert (c == oc).all() and (st == ost).all() and st[-1] == 9, (case, lean, rep)
                assert int(lf.astype(np.int64).sum()) == steps, (case, lean)
            locs, found, st2 = fm.locate_batch(ch, off, 5)  # the ranges k_count hands to the walk
            live = np.arange(5)[None, :] < of[:, None]
            assert (found == of).all() and (st2 == ost2).all() and (locs[live] == ol[live]).all(), (case, lean)
            got[lean] = (c, st, lf)
        for a, b in zip(got[1], got[0]):
            assert (a == b).all()
    finally:
        L.fmx_set_option(b"count_lean", 0)
        fm.close()
