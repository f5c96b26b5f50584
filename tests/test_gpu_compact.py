"""COMPACT images (option image_compact, BlobHeader.compact): the FM path over the compressed RrrVector form — 16-block
records + offsets stream, the value-of-offset table staged in LDS (kernels of namespace fmxc) — instead of the expanded
96-bit cells.  Same answers, statuses, LF-step counts and located positions as the oracle; a smaller resident image; the
same index queried through both images agrees with itself; images travel (fmx_attach_device_blob validates records)."""
import random

import numpy as np
import pytest

import index4j_amd as ia
import orc
from common import hdfs_text
from index4j_amd import workload
from parity_checks import GpuEngine, check_all

pytestmark = pytest.mark.gpu
HD = hdfs_text()


@pytest.fixture
def compact():
    assert ia.lib.fmx_set_option(b"image_compact", 1) == 0
    yield
    assert ia.lib.fmx_set_option(b"image_compact", 0) == 0


def test_every_query_kind_on_compact_images(compact):
    """the shared parity routine (count, locate with every cap form, extract, the three extractUntilBoundary modes, every
    status, full destination rows) over compact images: the fixture at sampleRate 1 / 7 / 32 / 64, embedded sentinels"""
    for sr, n in ((32, 150_000), (64, 80_000), (7, 60_000), (1, 30_000)):
        check_all(lambda t, s: GpuEngine(t, s), HD[:n], sr, random.Random(1000 + sr), n_q=100)
    rnd = random.Random(5)
    check_all(lambda t, s: GpuEngine(t, s), "ab\0cd\0\0ef" * 2000 + "tail", 4, rnd, n_q=60)
    check_all(lambda t, s: GpuEngine(t, s), "What a string!\nNow this is long, indeed\nBut others could be longer.", 2, rnd, n_q=40)


def test_compact_and_expanded_images_of_one_index_agree_and_the_compact_one_is_smaller(compact):
    """planned batches (suffix table grown by the compact kernels too), large alphabet (16-bit code words), long patterns,
    locate and extractUntilBoundary at batch size: compact == expanded == oracle; resident bytes per character"""
    t = workload.reference_text(22)
    o = orc.OracleFmIndex(t, 32, True)
    ser = o.write(False)
    fc = ia.FmIndex.read(ser, device=0)
    assert ia.lib.fmx_set_option(b"image_compact", 0) == 0
    fe = ia.FmIndex.read(ser, device=0)
    assert ia.lib.fmx_set_option(b"image_compact", 1) == 0
    nc, ne = fc.device_blob()[1], fe.device_blob()[1]
    assert nc < 0.9 * ne, (nc, ne)  # (a 4 M-character text: the tables weigh more than at 256 MiB)
    assert fc.suffix_table_info()[0] == fe.suffix_table_info()[0] >= 2
    pat, off, starts = workload.reference_queries(t, 60_000)
    orc.counters_reset()
    oc, ost = o.count_batch(pat, off, threads=8)
    steps = orc.counters()["lf_steps"]
    for f in (fc, fe):
        c, st, lf = f.count_batch(pat, off, want_steps=True)
        assert (c == oc).all() and (st == ost).all() and int(lf.astype(np.int64).sum()) == steps
    k = 20_000
    ol, of, _ = o.locate_batch(pat[: off[k]], off[: k + 1], 8, threads=8)
    for f in (fc, fe):
        locs, found, st = f.locate_batch(pat[: off[k]], off[: k + 1], 8)
        live = np.arange(8)[None, :] < found[:, None]
        assert (found == of).all() and (locs[live] == ol[live]).all() and int(st.max()) == 0
    fr = np.ascontiguousarray(ol[:, 0][of > 0][:8000]).astype(np.int32)
    odst, olen, ost2, oaux = o.extract_until_boundary_batch(0, fr, "\n", 600, threads=8)
    for f in (fc, fe):
        dst, out_len, st, aux = f.extract_boundary_batch(fr, "\n", 0, 600)
        assert (out_len == olen).all() and (st == ost2).all() and (dst == odst).all()
    xs = starts[:8000].astype(np.int32)
    xe = (xs + 32).astype(np.int32)
    d1, l1, s1 = fc.extract_batch(xs, xe, 32)[:3]
    d2, l2, s2 = fe.extract_batch(xs, xe, 32)[:3]
    assert (d1 == d2).all() and (l1 == l2).all() and (s1 == s2).all()
    fc.close()
    fe.close()


def test_a_compact_image_travels_and_damage_is_refused(compact):
    import torch

    text = ia.synth_log(1 << 20)
    fm = ia.FmIndex(text, 16, True, device=None)
    blob = np.frombuffer(fm.blob(), np.uint8).copy()
    good = torch.from_numpy(blob).cuda()
    q = ia.FmIndex.attach_device_blob(good.data_ptr(), good.numel(), 0)
    o = orc.OracleFmIndex(text, 16, True)
    pat, off, _ = ia.synth_patterns(text, 8, 30_000, seed=3)
    c, st = q.count_batch(pat, off)
    oc, ost = o.count_batch(pat, off, threads=8)
    assert (c == oc).all() and (st == ost).all()
    rnd = random.Random(2)
    refused = 0
    for _ in range(12):  # a flipped bit in the body fails the checksum; in a record it would fail the record checks too
        bad = blob.copy()
        bad[rnd.randrange(256, len(bad))] ^= 1 << rnd.randrange(8)
        t = torch.from_numpy(bad).cuda()
        try:
            ia.FmIndex.attach_device_blob(t.data_ptr(), t.numel(), 0)
        except Exception:  # noqa: BLE001
            refused += 1
    assert refused == 12
    q.close()
