"""fm_lf_step skips the reference's second call (rank(row, c) after inverseSelect(row-1), FM:532-535) only
when that call provably returns rank_before + 1.  This test evaluates BOTH forms for every row of several
indexes — run blocks, absent symbols, alphabets above 256 codes (the 8-bit mask quirk, WFBB:1332), block
and superblock boundaries — on the host simulation of the device source (CPU only)."""
import numpy as np
import pytest

import hostsim
import index4j_amd as ia
from common import hdfs_text


def texts():
    rng = np.random.default_rng(9)
    out = {}
    parts = []
    for i in range(30):
        parts.append("".join(chr(97 + int(x)) for x in rng.integers(0, 6 + i, 2500)))
        parts.append("zq" * 3000)
    out["runs_small_alphabet"] = ia.as_chars("".join(parts))
    # > 256 codes, then long runs of symbols whose code is >= 256 (run blocks whose symbol gets masked)
    big = np.concatenate([np.arange(300, 900, dtype=np.uint16), np.full(70_000, 880, np.uint16),
                          rng.integers(300, 900, 20_000).astype(np.uint16), np.full(70_000, 650, np.uint16),
                          np.tile(np.array([880, 650], np.uint16), 30_000)])
    out["runs_big_alphabet"] = big
    out["fixture_head"] = ia.as_chars(hdfs_text()[:60_000])
    out["two_superblocks"] = np.concatenate([rng.integers(65, 91, (1 << 20) - 7).astype(np.uint16),
                                             np.full(40_000, 66, np.uint16)])
    return out


@pytest.mark.parametrize("name", ["runs_small_alphabet", "runs_big_alphabet", "fixture_head", "two_superblocks"])
def test_fused_equals_two_call_form_on_every_row(name):
    t = texts()[name]
    fm = ia.FmIndex(t, 16, True, device=None)
    h = hostsim.HostSim(fm)
    L = fm.getInputLength()
    step = 1 if L < 400_000 else 7
    rows = np.unique(np.concatenate([np.arange(1, L + 1, step), [L, L - 1, 1 << 20, (1 << 20) + 1, (1 << 20) - 1]]))
    rows = rows[(rows >= 1) & (rows <= L)]
    for r in rows:
        o = h.lf_step_both(int(r))
        assert o[0] == o[2] and o[1] == o[3] and o[4] == o[5], (name, int(r), o.tolist())
