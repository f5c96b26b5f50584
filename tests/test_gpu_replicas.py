"""One immutable index on several GPUs behind the C ABI (include/fmx.h "replicas": fmx_replicate, fmx_*_multi, fmx_*_multi_dev).

What makes replication legal in the reference: FmIndex is @ThreadSafe and immutable (FM:82) and its own throughput benchmark
gives every thread an index of its own (FmIndexThroughputState.java:30).  A GPU box of this pool has ONE device, so the
replica sets here name device 0 two or three times: two replicas on one GPU run the very code path of two GPUs — their own
images, suffix tables, window directories, worker threads, streams — and must answer like the single index and like the
oracle, entry by entry, for batch sizes that the replica count does not divide.  The full-size forms (configs[1] and the
configs[4] share over replicas) are in test_gpu_configs_fullsize.py.  Run with `-m gpu` on an MI355X."""
import ctypes as C
import random

import numpy as np
import pytest

import index4j_amd as ia
import orc
from common import hdfs_text

pytestmark = pytest.mark.gpu
HD = hdfs_text()


def _queries(t16, rnd, n):
    L = len(t16)
    pats = [t16[s:s + rnd.randrange(1, 24)] for s in (rnd.randrange(max(1, L - 24)) for _ in range(n - 4))]
    pats += [ia.as_chars("zzzzqq"), t16[:1], ia.as_chars("INFO"), ia.as_chars("\n")]
    ch, off = ia.pack_patterns(pats)
    off = np.concatenate([off, [off[-1]]]).astype(np.int32)  # + an EMPTY pattern (FM:456-457: AIOOBE) in the last shard
    return ch, off


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0], [0]])
def test_every_query_kind_over_replicas_equals_single_index_and_oracle(devices):
    rnd = random.Random(61 + len(devices))
    text = HD[:180_001]
    t16 = ia.as_chars(text)
    fm = ia.FmIndex(text, 16, True, device=0)
    o = orc.OracleFmIndex(text, 16, True)
    rs = ia.ReplicaSet(fm, devices)
    try:
        assert rs.devices == devices and len(rs) == len(devices)
        for r in rs.replicas:  # a replica answers the accessors from its image's header; it keeps no host model
            assert r.getInputLength() == fm.getInputLength() and r.getAlphabetLength() == fm.getAlphabetLength() and str(r) == str(fm)
            with pytest.raises(ia.FmxError):
                r.write(False)
        res = rs.resident_bytes()
        assert all(b[0] == fm.device_blob()[1] for b in res) and all(b[1] == res[0][1] and b[2] == res[0][2] for b in res)
        ch, off = _queries(t16, rnd, 1001)  # 1,002 patterns: not a multiple of 3
        n = len(off) - 1
        c1, s1, lf1 = fm.count_batch(ch, off, want_steps=True)
        c2, s2, lf2 = rs.count_batch(ch, off, want_steps=True)
        oc, ost = o.count_batch(ch, off)
        assert (c2 == c1).all() and (s2 == s1).all() and (lf2 == lf1).all()
        assert (c2 == oc).all() and (s2 == ost).all() and s2[-1] == 9
        for mm, cap in ((16, 16), (-1, 40), (8, 3)):
            pre = np.full((n, cap), -7, np.int32)
            l1, f1, st1, w1 = fm.locate_batch(ch, off, mm, cap, want_steps=True, locs=pre.copy())
            l2, f2, st2, w2 = rs.locate_batch(ch, off, mm, cap, want_steps=True, locs=pre.copy())
            assert (l2 == l1).all() and (f2 == f1).all() and (st2 == st1).all() and (w2 == w1).all()  # whole rows: untouched slots too
            ol, of, ost2 = o.locate_batch(ch, off, mm, cap, fill=-7)
            ok = ost2 == 0  # (a row the reference abandons with an AIOOBE is compared between the two GPU forms above)
            assert (st2 == ost2).all() and (f2[ok] == of[ok]).all() and (l2[ok] == ol[ok]).all()
        L = len(t16) + 1
        a = np.array([rnd.randrange(L) for _ in range(500)], np.int32)
        b = np.minimum(a + np.array([rnd.randrange(60) for _ in range(500)], np.int32), L)
        a[:3], b[:3] = (-5, 3, 10), (10, L + 7, 5)
        d1, n1, e1 = fm.extract_batch(a, b, 70, 2, dst=np.full((500, 70), 9, np.uint16))
        d2, n2, e2 = rs.extract_batch(a, b, 70, 2, dst=np.full((500, 70), 9, np.uint16))
        od, on, oe = o.extract_batch(a, b, 70, 2, fill=9)
        assert (d2 == d1).all() and (n2 == n1).all() and (e2 == e1).all()
        assert (e2 == oe).all() and (d2[oe == 0] == od[oe == 0]).all() and (n2[oe == 0] == on[oe == 0]).all()
        fr = np.array([rnd.randrange(L) for _ in range(401)], np.int32)
        fr[:2] = (-1, L + 3)
        for mode in (0, 1, 2):
            for cap in (1 << 10, 40):
                x1 = fm.extract_boundary_batch(fr, "\n", mode, cap, dst=np.full((401, cap), 5, np.uint16))
                x2 = rs.extract_boundary_batch(fr, "\n", mode, cap, dst=np.full((401, cap), 5, np.uint16))
                ox = o.extract_until_boundary_batch(mode, fr, "\n", cap, fill=5)
                for u, v in zip(x1, x2):
                    assert (u == v).all(), (mode, cap)
                assert (x2[0] == ox[0]).all() and (x2[2] == ox[2]).all() and (x2[1][ox[2] == 0] == ox[1][ox[2] == 0]).all()
                assert (x2[3][ox[2] == 8] == ox[3][ox[2] == 8]).all()
        # fewer queries than replicas: the empty shards are skipped
        c3, s3 = rs.count_batch(ch[: off[1]], off[:2])
        assert c3[0] == c1[0] and s3[0] == 0
        c4, s4 = rs.count_batch(ch[:0], off[:1])
        assert len(c4) == 0
    finally:
        rs.close()
        fm.close()


def test_large_batches_take_the_pipelined_host_path_in_every_shard():
    """shards of >= 131,072 patterns go through the chunk pipeline of fmx_count_batch, each shard shipping only its own characters
    (pat_off + lo starts above 0); registered arrays take the mapped form"""
    t = ia.synth_log(1 << 22)
    fm = ia.FmIndex(t, 32, True, device=0)
    o = orc.OracleFmIndex.read(fm.write(False))
    rs = ia.ReplicaSet(fm, [0, 0])
    try:
        n = 300_001
        pat, off, _ = ia.synth_patterns(t, 8, n, seed=5)
        oc, ost = o.count_batch(pat, off, threads=8)
        c, s, lf = rs.count_batch(pat, off, want_steps=True)
        c1, s1, lf1 = fm.count_batch(pat, off, want_steps=True)
        assert (c == oc).all() and (s == ost).all() and (lf == lf1).all()
        arrays = [pat, off, c, lf, s]
        for a in arrays:
            assert ia.lib.fmx_host_register(a.ctypes.data, a.nbytes) == 0
        try:
            c[:] = -1
            assert ia.lib.fmx_count_batch_multi(rs.handles, 2, pat.ctypes.data, off.ctypes.data, n, c.ctypes.data, lf.ctypes.data,
                                                s.ctypes.data) == 0, ia.lib.fmx_last_error()
            assert (c == oc).all() and (lf == lf1).all() and (s == 0).all()
        finally:
            for a in arrays:
                ia.lib.fmx_host_unregister(a.ctypes.data)
        locs, found, st = rs.locate_batch(pat, off, 4)
        l1, f1, _ = fm.locate_batch(pat, off, 4)
        assert (locs == l1).all() and (found == f1).all() and (st == 0).all()
    finally:
        rs.close()
        fm.close()


def test_device_resident_shards_launched_from_the_workers():
    """fmx_count_batch_multi_dev / fmx_multi_synchronize: every replica's shard resident on its device, one stream per replica, the
    launches issued by the replicas' worker threads at once (what bench.py --single-process times)"""
    import torch

    t = ia.synth_log(1 << 22)
    fm = ia.FmIndex(t, 32, True, device=0)
    o = orc.OracleFmIndex.read(fm.write(False))
    R = 3
    rs = ia.ReplicaSet(fm, [0] * R)
    try:
        dev = torch.device("cuda", 0)
        sizes = [70_000, 0, 40_001]  # an idle replica in the middle
        streams = [torch.cuda.Stream(device=dev) for _ in range(R)]
        bufs, exp = [], []
        for r, n in enumerate(sizes):
            pat, off, _ = ia.synth_patterns(t, 8, max(n, 1), seed=50 + r)
            pat, off = pat[: n * 8], off[: n + 1]
            exp.append(o.count_batch(pat, off, threads=8)[0] if n else np.zeros(0, np.int32))
            bufs.append((torch.from_numpy(pat.view(np.int16).copy()).to(dev), torch.from_numpy(off.copy()).to(dev),
                         torch.full((max(n, 1),), -1, dtype=torch.int32, device=dev), torch.zeros(max(n, 1), dtype=torch.int32, device=dev),
                         torch.zeros(max(n, 1), dtype=torch.int32, device=dev)))
        torch.cuda.synchronize()
        vp = C.c_void_p * R
        arr = lambda k: vp(*[b[k].data_ptr() for b in bufs])  # noqa: E731
        ns = (C.c_int32 * R)(*sizes)
        sp = vp(*[s.cuda_stream for s in streams])
        for _ in range(3):
            rc = ia.lib.fmx_count_batch_multi_dev(rs.handles, R, arr(0), arr(1), ns, arr(2), arr(3), arr(4), sp)
            assert rc == 0, ia.lib.fmx_last_error()
        assert ia.lib.fmx_multi_synchronize(rs.handles, R, sp) == 0
        for r, n in enumerate(sizes):
            if n:
                assert (bufs[r][2].cpu().numpy()[:n] == exp[r]).all() and int(bufs[r][4].max().item()) == 0
            else:
                assert int(bufs[r][2][0].item()) == -1  # nothing was launched for it
    finally:
        rs.close()
        fm.close()


def test_segment_set_over_replicas_configs4_shape():
    """configs[4] behind the C ABI at a small size: 5 segment indexes x 2 replicas; counts summed over the segments and base-shifted
    hits of every shard against the single-device segment set and the per-segment oracles"""
    rnd = random.Random(99)
    text = HD[:150_000]
    t16 = ia.as_chars(text)
    sf = ia.SegmentedFmIndex(text, 16, True, device=0, segment_chars=32_000)
    assert len(sf) >= 5
    srs = ia.SegmentReplicaSet(sf, [0, 0])
    try:
        ch, off = _queries(t16, rnd, 700)
        mm = 6
        cnt, locs, found, st, lf = srs.count_locate_batch(ch, off, mm)
        c1, s1, lf1 = sf.count_batch(ch, off, want_steps=True)
        l1, f1, s2 = sf.locate_batch(ch, off, mm)
        assert (cnt == c1).all() and (lf == lf1).all() and (st == s1).all() and st[-1] == 9
        live = np.arange(mm)[None, :] < f1[:, None]
        assert (found == f1).all() and (locs[live] == l1[live]).all() and (locs[~live] == -1).all()
        ends = sf.bases[1:] + [len(t16)]
        oracles = [orc.OracleFmIndex(t16[a:b], 16, True) for a, b in zip(sf.bases, ends)]
        exp = np.zeros(len(off) - 1, np.int64)
        for oo in oracles:
            exp += oo.count_batch(ch, off)[0]
        assert (cnt == exp).all()
        # the host form on one device is the same call with one replica
        one = (C.c_void_p * len(sf))(*[s.handle for s in sf.segments])
        n = len(off) - 1
        cnt2 = np.zeros(n, np.int64)
        locs2 = np.full((n, mm), -1, np.int64)
        found2 = np.zeros(n, np.int32)
        assert ia.lib.fmx_count_locate_segments(one, len(sf), sf.base_array.ctypes.data, ch.ctypes.data, off.ctypes.data, n, mm,
                                                cnt2.ctypes.data, None, locs2.ctypes.data, found2.ctypes.data, None) == 0
        assert (cnt2 == cnt).all() and (found2 == found).all() and (locs2[live] == locs[live]).all()
    finally:
        srs.close()
        for f in sf.segments:
            f.close()


def test_an_index_moves_between_devices_and_back_cold_routes_included():
    """fmx_to_device on a resident index: everything the handle holds on the old device — image, suffix table, window
    directory, DevIndex.self (what the cold routes of rank() / inverseSelect() dereference), per-stream scratch — is released there
    and made again on the new one.  With window_cells = 0 and map_fast / inv_fast = 0 every LF-step takes the reference's own (cold)
    route through DevIndex.self.  On a box with one GPU the moves are 0 -> 0 -> 0; with two, 0 -> 1 -> 0."""
    rnd = random.Random(3)
    text = HD[:120_000]
    t16 = ia.as_chars(text)
    o = orc.OracleFmIndex(text, 8, True)
    n_dev = ia.lib.fmx_device_count()
    L = ia.lib
    assert L.fmx_set_option(b"window_cells", 0) == 0 and L.fmx_set_option(b"map_fast", 0) == 0 and L.fmx_set_option(b"inv_fast", 0) == 0
    try:
        fm = ia.FmIndex(text, 8, True, device=0)
        ch, off = _queries(t16, rnd, 400)
        fr = np.array([rnd.randrange(len(t16)) for _ in range(300)], np.int32)
        for device in ([1, 0] if n_dev > 1 else [0, 0]):
            fm.to_device(device)
            assert L.fmx_device_of(fm.handle) == device and fm.window_cells_bytes() == 0
            locs, found, st = fm.locate_batch(ch, off, 10)
            ol, of, ost = o.locate_batch(ch, off, 10)
            assert (found == of).all() and (st == ost).all() and (locs == ol).all()
            dst, ln, st2, aux = fm.extract_boundary_batch(fr, "\n", 0, 700)
            od, oln, ost2, oaux = o.extract_until_boundary_batch(0, fr, "\n", 700)
            assert (st2 == ost2).all() and (dst == od).all() and (ln[ost2 == 0] == oln[ost2 == 0]).all()
            d3, n3, s3 = fm.extract_batch(fr, np.minimum(fr + 40, len(t16) + 1), 48)
            od3, on3, os3 = o.extract_batch(fr, np.minimum(fr + 40, len(t16) + 1), 48)
            assert (s3 == os3).all() and (d3 == od3).all()
        fm.close()
    finally:
        L.fmx_set_option(b"window_cells", 2)
        L.fmx_set_option(b"map_fast", 1)
        L.fmx_set_option(b"inv_fast", 1)


def test_errors_of_the_replica_calls():
    L = ia.lib
    fm = ia.FmIndex(HD[:20_000], 16, True, device=None)  # host only
    out = (C.c_void_p * 2)()
    devs = np.array([0, 99], np.int32)
    assert L.fmx_replicate(fm.handle, devs.ctypes.data, 2, out) == ia._lib.E_ARG  # a device that does not exist: nothing made
    assert L.fmx_replicate(fm.handle, devs.ctypes.data, 0, out) == ia._lib.E_ARG
    assert L.fmx_device_of(fm.handle) == -1
    # a source that is not resident replicates from its host image
    rs = ia.ReplicaSet(fm, [0, 0])
    ch, off = ia.pack_patterns(["INFO", "blk_"])
    c, s = rs.count_batch(ch, off)
    assert c[0] > 0 and c[1] > 0 and (s == 0).all()
    z = np.zeros(2, np.int32)
    hs = (C.c_void_p * 2)(rs.replicas[0].handle, fm.handle)  # the second one is not resident
    assert L.fmx_count_batch_multi(hs, 2, ch.ctypes.data, off.ctypes.data, 2, z.ctypes.data, None, None) == ia._lib.E_NO_DEVICE
    assert L.fmx_count_batch_multi(None, 2, ch.ctypes.data, off.ctypes.data, 2, z.ctypes.data, None, None) == ia._lib.E_ARG
    assert L.fmx_count_batch_multi(rs.handles, 0, ch.ctypes.data, off.ctypes.data, 2, z.ctypes.data, None, None) == ia._lib.E_ARG
    # a failing shard reports which one it was; the other shard's results are still stored
    bad = off.copy()
    bad[2] = bad[1] - 1  # offsets that decrease, in shard 1
    rc = L.fmx_count_batch_multi(rs.handles, 2, ch.ctypes.data, bad.ctypes.data, 2, z.ctypes.data, None, None)
    assert rc == ia._lib.E_ARG and b"shard 1" in L.fmx_last_error()
    assert z[0] == c[0]
    rs.close()
    fm.close()


@pytest.mark.parametrize("n_segs", [3, 4])
def test_segment_set_side_stream_and_direct_stores_toggled(n_segs):
    """the side-stream overlap of a segment set's range searches (two alternating {found, status, range, count} sets,
    2 * n_segs + 1 events) only engaged from 262,144 patterns on, which no test reached.  Option segments_overlap_min = 0 engages
    it for any batch: odd and even segment counts, hits stored directly / staged, with and without counts, against the two
    separate calls."""
    import torch

    rnd = random.Random(17 + n_segs)
    text = HD[: 40_000 * n_segs]
    t16 = ia.as_chars(text)
    sf = ia.SegmentedFmIndex(text, 16, True, device=0, segment_chars=len(t16) // n_segs + 2_000)
    K = len(sf)
    assert K >= n_segs - 1
    dev = torch.device("cuda", 0)
    ch, off = _queries(t16, rnd, 900)
    n, mm = len(off) - 1, 7
    d_pat = torch.from_numpy(np.ascontiguousarray(ch).view(np.int16)).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L = ia.lib
    c0, s0, lf0 = sf.count_batch(ch, off, want_steps=True)
    l0, f0, st0 = sf.locate_batch(ch, off, mm)
    live = np.arange(mm)[None, :] < f0[:, None]
    try:
        for overlap in (1, 0):
            for direct in (1, 0):
                for with_counts in (True, False):
                    assert L.fmx_set_option(b"segments_overlap", overlap) == 0 and L.fmx_set_option(b"segments_direct", direct) == 0
                    assert L.fmx_set_option(b"segments_overlap_min", 0) == 0
                    d_cnt = torch.zeros(n, dtype=torch.int64, device=dev)
                    d_lf = torch.zeros(n, dtype=torch.int64, device=dev)
                    d_locs = torch.full((n * mm,), -1, dtype=torch.int64, device=dev)
                    d_found = torch.zeros(n, dtype=torch.int32, device=dev)
                    d_st = torch.zeros(n, dtype=torch.int32, device=dev)
                    d_tmp = torch.zeros(n * (4 + mm), dtype=torch.int32, device=dev)
                    for _ in range(2):  # twice: the lane's events and buffers are reused
                        if with_counts:
                            rc = L.fmx_count_locate_segments_dev(sf.handles, K, sf.base_array.ctypes.data, d_pat.data_ptr(), d_off.data_ptr(),
                                                                 n, mm, d_cnt.data_ptr(), d_lf.data_ptr(), d_locs.data_ptr(), d_found.data_ptr(),
                                                                 d_st.data_ptr(), d_tmp.data_ptr(), sp)
                        else:
                            rc = L.fmx_locate_segments_dev(sf.handles, K, sf.base_array.ctypes.data, d_pat.data_ptr(), d_off.data_ptr(), n,
                                                           mm, d_locs.data_ptr(), d_found.data_ptr(), d_st.data_ptr(), d_tmp.data_ptr(), sp)
                        assert rc == 0, L.fmx_last_error()
                    torch.cuda.synchronize()
                    key = (overlap, direct, with_counts)
                    got = d_locs.cpu().numpy().reshape(n, mm)
                    assert (d_found.cpu().numpy() == f0).all() and (got[live] == l0[live]).all(), key
                    assert (d_st.cpu().numpy() == st0).all(), key
                    if with_counts:
                        assert (d_cnt.cpu().numpy() == c0).all() and (d_lf.cpu().numpy() == lf0).all(), key
    finally:
        L.fmx_set_option(b"segments_overlap", 1)
        L.fmx_set_option(b"segments_direct", 1)
        L.fmx_set_option(b"segments_overlap_min", 262144)
        for f in sf.segments:
            f.close()


def test_several_host_threads_share_one_replica_set():
    """the sharded calls from four host threads at once on ONE replica set (a JVM's request threads): the library's workers are per
    (device, replica slot), so concurrent calls queue behind each other on a worker — every call still gets its own results"""
    import threading

    t = ia.synth_log(1 << 21)
    fm = ia.FmIndex(t, 16, True, device=0)
    o = orc.OracleFmIndex.read(fm.write(False))
    rs = ia.ReplicaSet(fm, [0, 0, 0])
    work = []
    for k, n in enumerate([50_001, 7, 20_000, 140_000]):
        pat, off, pos = ia.synth_patterns(t, 8, n, seed=300 + k)
        oc, ost = o.count_batch(pat, off, threads=8)
        ol, of, _ = o.locate_batch(pat[: min(n, 3000) * 8], off[: min(n, 3000) + 1], 8, threads=8)
        work.append((pat, off, oc, ost, ol, of))
    errors = []
    gate = threading.Barrier(len(work))

    def run(k):
        try:
            pat, off, oc, ost, ol, of = work[k]
            m = len(of)
            gate.wait()
            for _ in range(5):
                c, s = rs.count_batch(pat, off)
                assert (c == oc).all() and (s == ost).all(), "count, thread %d" % k
                locs, found, st = rs.locate_batch(pat[: m * 8], off[: m + 1], 8)
                live = np.arange(8)[None, :] < found[:, None]
                assert (found == of).all() and (locs[live] == ol[live]).all(), "locate, thread %d" % k
        except BaseException as e:  # noqa: BLE001 - reported to the main thread
            errors.append(e)

    threads = [threading.Thread(target=run, args=(k,)) for k in range(len(work))]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    rs.close()
    fm.close()
    assert not errors, errors[0]
