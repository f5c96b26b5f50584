"""The oracle's batch helpers (OpenMP loops used by the full-size parity tests and bench.py's secondary
checks) against its scalar calls — same answers, same counter totals, any thread count."""
import random

import numpy as np

import orc
from common import hdfs_text

HD = hdfs_text()


def test_batch_helpers_equal_the_scalar_calls():
    rnd = random.Random(5)
    o = orc.OracleFmIndex(HD[:60_000], 8, True)
    t16 = orc.u16(HD[:60_000])
    L = len(t16)
    pats = [t16[s:s + rnd.randrange(1, 12)] for s in (rnd.randrange(L - 12) for _ in range(300))]
    off = np.zeros(len(pats) + 2, np.int32)
    off[1:-1] = np.cumsum([len(p) for p in pats])
    off[-1] = off[-2]  # one empty pattern: AIOOBE
    ch = np.concatenate(pats)
    for mm, cap in ((16, 16), (-1, 40), (8, 3)):
        orc.counters_reset()
        exp = []
        for i in range(len(off) - 1):
            try:
                exp.append((0,) + o.locate(ch[off[i]:off[i + 1]], max_matches=mm, cap=cap))
            except IndexError:
                exp.append((9, None, None))
        c1 = orc.counters()
        for threads in (1, 3):
            orc.counters_reset()
            locs, found, st = o.locate_batch(ch, off, mm, cap, threads=threads, fill=-7)
            c2 = orc.counters()
            assert c2["lf_steps"] == c1["lf_steps"] and c2["alg_bytes"] == c1["alg_bytes"]
            for i, (est, en, el) in enumerate(exp):
                assert st[i] == est
                if est == 0:
                    assert found[i] == en and (locs[i, :en] == el).all() and (locs[i, en:] == -7).all()
    fr = np.array([rnd.randrange(L) for _ in range(200)] + [-1, L + 5], np.int32)
    for mode in (0, 1, 2):
        for cap, offs in ((512, 0), (30, 0), (90, 5)):
            for threads in (1, 4):
                dst, ol, st, aux = o.extract_until_boundary_batch(mode, fr, "\n", cap, offs, threads=threads)
                for i in range(len(fr)):
                    try:
                        n, d = o.extract_until_boundary(mode, int(fr[i]), cap, offs, "\n")
                        assert st[i] == 0 and ol[i] == n and (dst[i] == d).all()
                    except RuntimeError as e:
                        assert st[i] in (2, 5, 8)
                        if st[i] == 8:
                            assert str(e).endswith(": %d" % aux[i])
                    except IndexError:
                        assert st[i] == 9
    a = np.array([rnd.randrange(L) for _ in range(100)] + [-3, 5], np.int32)
    b = np.minimum(a + 40, L).astype(np.int32)
    b[-1] = L + 9
    dst, ol, st = o.extract_batch(a, b, 48, 3, threads=2)
    for i in range(len(a)):
        try:
            n, d = o.extract(int(a[i]), int(b[i]), dest_len=48, offset=3)
            assert st[i] == 0 and ol[i] == n and (dst[i] == d).all()
        except RuntimeError:
            assert st[i] in (2, 3)
