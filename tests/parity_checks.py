"""Parity check routines shared by test_hostsim_parity.py (device source simulated on the host) and
test_gpu_parity.py (the HIP kernels through the C ABI).  `make_engine(text, sample_rate)` returns an
object with count_batch / locate_batch / extract_batch / extract_boundary_batch in HostSim's shapes."""
import numpy as np

import index4j_amd as ia
import orc


class GpuEngine:
    """adapter: index4j_amd.FmIndex (C ABI, HIP kernels) in the call shapes of hostsim.HostSim"""

    def __init__(self, text, sr, extract=True):
        self.fm = ia.FmIndex(text, sr, extract, device=0)

    def count_batch(self, ch, off):
        c, st, lf = self.fm.count_batch(ch, off, want_steps=True)
        return c, st, lf, None

    def locate_batch(self, ch, off, mm, cap):
        locs, found, st, lf = self.fm.locate_batch(ch, off, mm, cap, want_steps=True)
        return locs, found, st, lf

    def extract_batch(self, a, b, dst_len, offset=0, dst=None):
        dst, ol, st, lf = self.fm.extract_batch(a, b, dst_len, offset, dst=dst, want_steps=True)
        return dst, ol, st, lf

    def extract_boundary_batch(self, fr, boundary, mode, dst_len, offset=0, dst=None):
        dst, ol, st, aux, lf = self.fm.extract_boundary_batch(fr, boundary, mode, dst_len, offset, dst=dst, want_steps=True)
        return dst, ol, st, aux, lf


def check_all(make_engine, text, sr, rnd, n_q=120):
    """every query kind, incl. error statuses and the full destination buffers, engine vs oracle"""
    h = make_engine(text, sr)
    o = orc.OracleFmIndex(text, sr, True)
    assert h.fm.write(False) == o.write(False)
    t16 = ia.as_chars(text)
    L = len(t16)
    pats = [t16[s:s + rnd.randrange(1, 24)] for s in (rnd.randrange(max(1, L - 24)) for _ in range(n_q))]
    pats += [ia.as_chars("zzzzqq"), t16[:1], ia.as_chars("\0")]
    ch, off = ia.pack_patterns(pats)
    off = np.concatenate([off, [off[-1]]]).astype(np.int32)  # plus one EMPTY pattern (FM:456-457 -> AIOOBE)
    cnt, st, lf, rng = h.count_batch(ch, off)
    oc, ost = o.count_batch(ch, off)
    assert (cnt == oc).all() and (st == ost).all()
    assert st[-1] == 9
    for mm, cap in ((16, 16), (-1, 64), (3, 8), (8, 3)):
        locs, found, st2, lf2 = h.locate_batch(ch, off, mm, cap)
        for i, p in enumerate(pats):
            try:
                n, l = o.locate(p, max_matches=mm, cap=cap)
                assert st2[i] == 0 and n == found[i] and (l == locs[i, :n]).all(), (sr, i, mm, cap)
            except IndexError:
                assert st2[i] == 9
    a = np.array([rnd.randrange(L) for _ in range(n_q)], np.int32)
    b = np.minimum(a + np.array([rnd.randrange(60) for _ in range(n_q)], np.int32), L)
    a[:3] = (-5, 3, 10)
    b[:3] = (10, L + 7, 5)
    dst, ol, st3, lf3 = h.extract_batch(a, b, 50, 2)
    for i in range(n_q):
        try:
            n, d = o.extract(int(a[i]), int(b[i]), dest_len=50, offset=2)
            assert st3[i] == 0 and n == ol[i] and (d == dst[i]).all()
        except (RuntimeError, IndexError) as e:
            assert st3[i] != 0, (i, e)
    fr = np.array([rnd.randrange(L) for _ in range(n_q)], np.int32)
    fr[:2] = (-1, L + 3)
    if isinstance(text, str):
        bch = "\n" if "\n" in text else text[len(text) // 2]
    else:  # uint16 array
        bch = 10 if (t16 == 10).any() else int(t16[L // 2])
    for mode in (0, 1, 2):
        for cap, offs in ((1 << 12, 0), (40, 0), (90, 5), (0, 0)):
            results = [h.extract_boundary_batch(fr, bch, mode, cap, offs)]
            if hasattr(h, "blob"):  # host simulation: also the literal +4-chunk form (accelerate=0)
                results.append(h.extract_boundary_batch(fr, bch, mode, cap, offs, accelerate=0))
                results.append(h.extract_boundary_batch(fr, bch, mode, cap, offs, accelerate=2))  # group form, G = 1
                results.append(h.extract_boundary_batch(fr, bch, mode, cap, offs, accelerate=3))  # + interleaved first walks (fm_lf_step2)
            for dst, ol, st4, aux, lf4 in results:
                for i in range(n_q):
                    try:
                        n, d = o.extract_until_boundary(mode, int(fr[i]), cap, offs, bch)
                        assert st4[i] == 0 and n == ol[i] and (d == dst[i]).all(), (sr, mode, cap, offs, i)
                    except RuntimeError as e:
                        if "Currently extracted" in str(e):
                            assert st4[i] == 8 and str(e).endswith(": %d" % aux[i])
                        else:
                            assert st4[i] in (1, 2, 5)
                    except ValueError:
                        assert st4[i] in (6, 7)
                    except IndexError:
                        assert st4[i] == 9
        dst, ol, st5, aux, _ = h.extract_boundary_batch(fr[2:10], "이", mode, 64, 0)
        assert (st5 == 7).all()
