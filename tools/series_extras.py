#!/usr/bin/env python3
"""The rows of BASELINE.md §1 that the reference-shaped series (tools/ref_series.py) does not carry, each with an MI355X
figure beside the published one and each checked against the oracle in the same run (bench.py --series-extras):

* RrrVectorThroughputBenchmark.java:43-51 — stand-alone RrrVector.rankOnes on 10 M random bits at sampleSize 16 / 32 / 64 /
  256, and on a 1 %-dense vector at sampleSize 32: fmx_rrr_rank_ones_batch_dev (compressed 15-bit blocks, class + offset,
  value-of-offset table in LDS), positions resident in HBM;
* FmIndexThroughputBenchmark.java:231-249 locateAndExtractBenchmark (20,000 queries of 8..31 chars, maxMatches 1000, 64 chars
  per hit, sampleRate 32) as the fused device pipeline fmx_locate_extract_batch_dev, on the 1,099-symbol 2^text_log2 text;
* FmIndexIngestBenchmark.java:48 and FmIndexSerializedSizeBenchmark.java:57,59 — build time and serialized bytes per
  character of that text at sampleRate 32 / 64.
usage: python tools/series_extras.py [--text-log2 28]   (GPU box)"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

HBM_PEAK_GBS = 8000.0
# BASELINE.md §1 (JMH, one core of a Xeon W-10885)
PUBLISHED_RRR = {(0.5, 16): 7.16e6, (0.5, 32): 6.12e6, (0.5, 64): 4.49e6, (0.5, 256): 1.69e6, (0.01, 32): 7.16e6}


def _mean_ms(torch, stream, fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        fn()
    e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def rrr_rows(ia, torch, orc, dev, log, n_bits=10_000_000, queries=1 << 22):
    stream = torch.cuda.current_stream()
    sp = C.c_void_p(stream.cuda_stream)
    cores = os.cpu_count() or 1
    rng = np.random.default_rng(42)
    pos = rng.integers(0, n_bits + 1, queries).astype(np.int32)
    d_pos = torch.from_numpy(pos).to(dev)
    d_out = torch.zeros(queries, dtype=torch.int32, device=dev)
    rows = []
    for density, s in ((0.5, 16), (0.5, 32), (0.5, 64), (0.5, 256), (0.01, 32)):
        bits = (np.random.default_rng(7).random(n_bits) < density).astype(np.uint8)
        rv = ia.RrrVector(bits, s, device=dev.index or 0)
        ref = orc.Rrr(bits=bits, sample=s)

        def call():
            rc = ia.lib.fmx_rrr_rank_ones_batch_dev(rv._h, d_pos.data_ptr(), queries, d_out.data_ptr(), sp)
            assert rc == 0, ia.lib.fmx_last_error()

        call()
        torch.cuda.synchronize()
        got = d_out.cpu().numpy()
        # the definition (ones before the position) for every query, the reference's algorithm (oracle) for every query too
        prefix = np.concatenate([[0], np.cumsum(bits, dtype=np.int64)])
        if not (got == prefix[pos]).all():
            raise RuntimeError("rankOnes differs from the bit count (density %g, sampleSize %d)" % (density, s))
        orc.counters_reset()
        exp = ref.rank_ones_batch(pos, threads=cores)
        alg = orc.counters()["alg_bytes"]
        if not (got == exp).all():
            raise RuntimeError("rankOnes differs from the oracle (density %g, sampleSize %d)" % (density, s))
        k1 = min(queries, 1 << 20)
        t0 = time.perf_counter()
        ref.rank_ones_batch(pos[:k1], threads=1)
        cpu_s = time.perf_counter() - t0
        ms = _mean_ms(torch, stream, call, 20)
        rows.append({"benchmark": "RrrVector.rankOnes", "bits": n_bits, "density": density, "sample_size": s, "queries": queries,
                     "ms_per_batch": ms, "ops_per_s": queries / ms * 1e3,
                     "alg_bytes_per_op": alg / queries,
                     "roofline": {"bound": "hbm", "achieved": alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "note": "algorithmic bytes of RRR:358-396 (sample pair + scanned class nibbles + offset bits); the "
                                          "vector is 2-3 MB and L2-resident: the kernel is bound by the class scan's instructions"},
                     "cpu_oracle_1core_ops_per_s": k1 / cpu_s,
                     "published_reference_ops_per_s_1core_xeon": PUBLISHED_RRR[(density, s)],
                     "checked_vs_oracle": "all %d ranks (oracle RRR:358-396 and the plain bit count)" % queries})
        log("[extras] rankOnes density %g sampleSize %d: %.4f ms per %d, %.3g ops/s (C port 1 core %.3g, published %.3g)"
            % (density, s, ms, queries, rows[-1]["ops_per_s"], k1 / cpu_s, PUBLISHED_RRR[(density, s)]))
        rv.close()
        del ref
    return rows


def pipeline_and_ingest_rows(ia, torch, orc, dev, log, text_log2=28, queries=20000, max_matches=1000, extract_len=64, check=400):
    from index4j_amd import workload

    stream = torch.cuda.current_stream()
    sp = C.c_void_p(stream.cuda_stream)
    cores = os.cpu_count() or 1
    text = workload.reference_text(text_log2)
    n_text = len(text)
    rows = []
    ser = {}
    fm32 = None
    for s in (32, 64):
        t0 = time.perf_counter()
        fm = ia.FmIndex(text, s, True, device=None, build_device=dev.index or 0)
        t_build = time.perf_counter() - t0
        raw = fm.write(False)
        ser[s] = len(raw)
        rows.append({"benchmark": "ingest + serialized size", "sample_rate": s, "text_chars": n_text, "symbols": int(len(np.unique(text))),
                     "build_s": t_build, "chars_per_s": n_text / t_build, "wavelet_encode_on_device_s": (fm.build_stats or {}).get("wavelet_device_seconds"),
                     "suffix_array_on_device_s": (fm.build_stats or {}).get("device_stage_seconds"),
                     "serialized_bytes": len(raw), "serialized_bytes_per_char": len(raw) / n_text,
                     "published_reference": {"build_s_184MB": 70.5, "serialized_fraction_of_text_bytes": 0.44 if s == 32 else 0.31,
                                             "note": "FmIndexIngestBenchmark.java:48, FmIndexSerializedSizeBenchmark.java:57,59 "
                                                     "(Android.log, 184 MB of bytes)"}})
        log("[extras] sampleRate %d: built in %.2f s (%.3g chars/s), serialized %.3f B/char" % (s, t_build, n_text / t_build, len(raw) / n_text))
        if s == 32:
            fm32 = fm
            ref = orc.OracleFmIndex.read(raw)
        else:
            fm.close()
        del raw
    fm = fm32
    fm.to_device(dev.index or 0)
    pat, off, _starts = workload.reference_queries(text, queries)
    n, mm, xl = queries, max_matches, extract_len
    inlen = fm.getInputLength()
    d_pat = torch.from_numpy(pat.view(np.int16)).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    slots = n * mm
    d_locs = torch.zeros(slots, dtype=torch.int32, device=dev)
    d_found = torch.zeros(n, dtype=torch.int32, device=dev)
    d_len = torch.zeros(slots, dtype=torch.int32, device=dev)
    d_hst = torch.zeros(slots, dtype=torch.int32, device=dev)
    d_lf = torch.zeros(n, dtype=torch.int32, device=dev)
    d_st = torch.zeros(n, dtype=torch.int32, device=dev)
    d_ws = torch.zeros(2 * n, dtype=torch.int32, device=dev)
    d_dst = torch.zeros(slots * xl, dtype=torch.int16, device=dev)

    def run():
        rc = ia.lib.fmx_locate_extract_batch_dev(fm.handle, d_pat.data_ptr(), d_off.data_ptr(), n, mm, xl, d_locs.data_ptr(),
                                                 d_found.data_ptr(), d_dst.data_ptr(), d_len.data_ptr(), d_lf.data_ptr(),
                                                 d_st.data_ptr(), d_hst.data_ptr(), d_ws.data_ptr(), sp)
        assert rc == 0, ia.lib.fmx_last_error()

    run()
    torch.cuda.synchronize()
    found = d_found.cpu().numpy()
    locs = d_locs.cpu().numpy().reshape(n, mm)
    hst = d_hst.cpu().numpy().reshape(n, mm)
    olen = d_len.cpu().numpy().reshape(n, mm)
    assert int(d_st.max().item()) == 0
    # every query's hits against the oracle's locate; the extracted rows of the first `check` queries against its extract
    olocs, ofound, ost = ref.locate_batch(pat, off, mm, threads=cores)
    live = np.arange(mm)[None, :] < found[:, None]
    if not ((found == ofound).all() and (locs[live] == olocs[live]).all()):
        raise RuntimeError("locateAndExtract: hits differ from the oracle")
    rows_dev = d_dst.view(slots, xl)
    t0 = time.perf_counter()
    cpu_hits = 0
    for i in range(min(check, n)):
        k = int(found[i])
        got = rows_dev[i * mm:i * mm + k].cpu().numpy().view(np.uint16)
        starts = locs[i, :k].astype(np.int32)
        stops = np.minimum(inlen, starts + xl).astype(np.int32)
        for j in range(k):
            if stops[j] >= inlen:
                assert hst[i, j] == 3  # "Stop position longer than index string" FM:572-574
                continue
            m, d = ref.extract(int(starts[j]), int(stops[j]), dest_len=xl)
            if not (hst[i, j] == 0 and olen[i, j] == m and (d == got[j]).all()):
                raise RuntimeError("locateAndExtract: extracted row differs from the oracle (query %d, hit %d)" % (i, j))
        cpu_hits += k
    cpu_s = time.perf_counter() - t0
    hits = int(found.sum())
    ms = _mean_ms(torch, stream, run, 5)
    rows.append({"benchmark": "locateAndExtract", "sample_rate": 32, "queries": n, "max_matches": mm, "extract_chars": xl,
                 "ms_per_batch": ms, "ops_per_s": n / ms * 1e3, "hits": hits, "hits_per_s": hits / ms * 1e3,
                 "chars_extracted": int(olen[olen > 0].sum()),
                 "published_reference_ops_per_s_1core_xeon": None,
                 "published_note": "FmIndexThroughputBenchmark.java:231-249 carries no result block for this benchmark",
                 "cpu_oracle_1core": {"queries": min(check, n), "hits": cpu_hits, "seconds": cpu_s,
                                      "note": "extract of every hit of the checked queries through the oracle, incl. the per-hit Python call"},
                 "checked_vs_oracle": "hits of all %d queries (SA order); extracted rows and statuses of the first %d queries' %d hits"
                                      % (n, min(check, n), cpu_hits)})
    log("[extras] locateAndExtract: %.3f ms per %d queries (%d hits), %.3g queries/s" % (ms, n, hits, n / ms * 1e3))
    fm.close()
    return rows


def run_extras(ia, torch, orc, dev, text_log2=28, log=lambda *a: None):
    rows = rrr_rows(ia, torch, orc, dev, log)
    rows += pipeline_and_ingest_rows(ia, torch, orc, dev, log, text_log2=text_log2)
    return {"rows": rows}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--text-log2", type=int, default=28)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "series_extras.json"))
    args = ap.parse_args()
    import torch

    import index4j_amd as ia
    import orc

    out = run_extras(ia, torch, orc, torch.device("cuda", 0), text_log2=args.text_log2, log=lambda *a: print(*a, file=sys.stderr, flush=True))
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    for r in out["rows"]:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
