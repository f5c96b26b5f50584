#!/usr/bin/env python3
"""The reference's own benchmark shapes on one MI355X: countBenchmark, locateBenchmark with
maxMatches 1 / 10 / 100 / 1000 and extractBenchmark (32 chars), each at sampleRate 1 / 32 / 64, queries of 8..31 chars
sampled from the text (FmIndexThroughputBenchmark.java:44-249, FmIndexThroughputState.java:76-83) — on a text with the
published data set's alphabet size (> 1,000 symbols, README.md:291-292; synthetic: index4j_amd/workload.py).

Operands resident in HBM, HIP-event time over the device-pointer entry points, EVERY result of every row checked against
the oracle in the run (counts, found, every located position in SA order, extracted chars, statuses, LF-step totals);
`roofline.frac` = algorithmic bytes of the LF-steps the kernels EXECUTE (oracle counting mode, minus what the suffix
table answers) / time / 8 TB/s.

    python tools/ref_series.py [--text-log2 28] [--queries 262144] [--out gpurun_out/ref_series.json]

bench.py calls run_series() for its `secondary.reference_series` block."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBS = 8000.0

# ops/s on one core of a Xeon W-10885 (JMH), Android.log 184 MB; key = (benchmark, maxMatches, sampleRate)
PUBLISHED = {("locate", 1, 1): 57444, ("locate", 1, 32): 26031, ("locate", 1, 64): 13749, ("locate", 10, 32): 7223,
             ("locate", 100, 32): 1120, ("locate", 1000, 32): 172.8, ("extract", 32, 1): 43004, ("extract", 32, 32): 19545,
             ("extract", 32, 64): 12451}


def row_key(bench, mm, s):
    """name of a series row: what bench.py's contract line, tools/pmc_rows.py and profiles/pmc_latest.json call it"""
    return "series %s%s s=%d" % (bench, "" if mm is None else "(%d)" % mm, s)


# the rows of bench.py's DEFAULT run: the reference's published shape at sampleRate 32 and the weak rows at 1
DEFAULT_PLAN = ((32, (("count", None), ("locate", 1), ("locate", 100), ("extract", 32))),
                (1, (("count", None), ("extract", 32))))


def full_plan(sample_rates=(1, 32, 64), max_matches=(1, 10, 100, 1000)):
    return tuple((s, (("count", None),) + tuple(("locate", mm) for mm in max_matches) + (("extract", 32),)) for s in sample_rates)


def series_queries(Q, s, bench, mm, bounded=True):
    """queries of a row: the locate rows with maxMatches 100 / 1000 (sampleRate > 1) take the first Q / 4 and Q / 16 of the batch"""
    if bench != "locate" or not bounded or mm < 100 or s == 1:
        return Q
    return max(1, Q // (4 if mm < 1000 else 16))


def settle_frac(roof, ms, traffic, ratio_rule=True):
    """One rule for every `frac` on bench.py's lines.  ratio_rule = False (the contract's own `roofline`
    object, whose `achieved` the task defines as algorithmic bytes / launch time): only the "never above 1" half applies, the
    counter figure stands beside it as traffic / traffic_frac.  roof["frac"] comes in as the ALGORITHMIC fraction
    (oracle-counted bytes of the executed LF-steps / time / peak).  `traffic` = bytes the memory system moved per call by the
    committed counters (FETCH_SIZE x calibration + WRITE_SIZE; None: no counters for these kernel sources).  The algorithmic
    figure stays under `frac_algorithmic`; `frac` is the counter figure wherever the algorithmic one exceeds 1 or 1.5 x the
    counter figure (its numerator then counts bytes that lanes shared in L1 / L2 and the memory system never moved)."""
    alg = roof.get("frac")
    roof["frac_algorithmic"] = alg
    roof["achieved_algorithmic"] = roof.get("achieved")
    roof["traffic"] = traffic
    if traffic:
        tf = traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        roof["traffic_frac"] = tf
        if alg is not None and (alg > 1.0 or (ratio_rule and alg > 1.5 * tf)):
            roof["frac"] = tf
            roof["achieved"] = traffic / (ms * 1e-3) / 1e9
            roof["frac_is"] = "counter traffic (the algorithmic figure counts bytes that equal work side by side shares in L1 / L2)"
        else:
            roof["frac_is"] = "algorithmic bytes"
    else:
        roof["traffic_frac"] = None
        if alg is not None and alg > 1.0:  # no counters to fall back to: never report a fraction above 1
            roof["frac"] = None
            roof["achieved"] = None
            roof["frac_is"] = "withheld: the algorithmic figure exceeds 1 and no counter traffic is on file for these kernel sources"
        else:
            roof["frac_is"] = "algorithmic bytes"
    return roof


def _timed(torch, stream, fn, reps):
    """MEAN milliseconds per call over 3 x reps back-to-back calls between one HIP-event pair (the same standard as the
    headline's ms_per_step; minima were reported until round 3)"""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(3 * reps):
        fn()
    e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * reps)


def run_series(ia, torch, orc, dev, text_log2=28, queries=1 << 20, sample_rates=(1, 32, 64), max_matches=(1, 10, 100, 1000),
               symbols=None, build_device=0, log=lambda *a: None, bounded=True, plan=None, check=True, calls=None,
               traffic_lookup=None):
    """bounded: the locate rows with maxMatches 100 / 1000 (sampleRate > 1) take the first queries / 4 and queries / 16 of the batch — the
    oracle's check of every located position is what takes the time (256 host cores: 150 s for 262,144 queries x 1000 at
    sampleRate 64), and bench.py's default run has to finish within minutes.
    plan: ((sampleRate, ((benchmark, maxMatches | chars | None), ...)), ...) — the rows to run (default: every row of
    sample_rates x max_matches).  check = False (counter passes, tools/pmc_rows.py): no oracle, every row's call is launched
    `calls` times and nothing is reported.  traffic_lookup(row key, queries) -> bytes per call by the committed counters."""
    from index4j_amd import workload

    if plan is None:
        plan = full_plan(sample_rates, max_matches)

    symbols = workload.REFERENCE_SYMBOLS if symbols is None else symbols
    t_start = time.time()
    user_log = log

    def log(msg):  # every line with the seconds since the series began
        user_log("%s  [+%.1fs]" % (msg, time.time() - t_start))

    cores = os.cpu_count() or 1
    stream = torch.cuda.current_stream()
    sp = C.c_void_p(stream.cuda_stream)
    t0 = time.time()
    text = workload.reference_text(text_log2, symbols)
    n_text = len(text)
    pat, off, starts = workload.reference_queries(text, queries)
    Q = queries
    lens = np.diff(off)
    d_pat = torch.from_numpy(pat.view(np.int16)).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    d_cnt = torch.zeros(Q, dtype=torch.int32, device=dev)
    d_lf = torch.zeros(Q, dtype=torch.int32, device=dev)
    d_st = torch.zeros(Q, dtype=torch.int32, device=dev)
    d_rng = torch.zeros(2 * Q, dtype=torch.int32, device=dev)
    d_found = torch.zeros(Q, dtype=torch.int32, device=dev)
    xl = workload.MAX_QUERY
    d_start = torch.from_numpy(starts).to(dev)
    d_stop = torch.from_numpy((starts + xl).astype(np.int32)).to(dev)
    d_dst = torch.zeros(Q * xl, dtype=torch.int16, device=dev)
    d_len = torch.zeros(Q, dtype=torch.int32, device=dev)
    rows = []
    info = {"text": "synthetic log with runs of multi-byte characters, 2^%d chars, %d distinct symbols "
                    "(fmx_synth_log_multichar, seed 42)" % (text_log2, len(np.unique(text))),
            "queries": Q, "query_shape": "substrings of the text, 8..31 chars (mean %.1f), start uniform" % lens.mean(),
            "indexes": []}
    log("[series] text + %d queries in %.1fs" % (Q, time.time() - t0))

    def check_rc(rc, what):
        if rc != 0:
            raise RuntimeError("%s failed: %s" % (what, ia.lib.fmx_last_error().decode()))

    for s, wanted in plan:
        t0 = time.time()
        fm = ia.FmIndex(text, s, True, device=None, build_device=build_device)
        t_build = time.time() - t0
        ser = fm.write(False)
        t1 = time.time()
        fm.to_device(dev.index or 0)
        t_dev = time.time() - t1
        ref = orc.OracleFmIndex.read(ser) if check else None
        image_bytes = fm.device_blob()[1]
        table_chars, table_bytes = fm.suffix_table_info()
        window_bytes = fm.window_cells_bytes()
        info["indexes"].append({"sample_rate": s, "build_s": t_build, "flatten_upload_table_s": t_dev,
                                "serialized_bytes_per_char": len(ser) / n_text, "image_bytes_per_char": image_bytes / n_text,
                                "suffix_table_chars": table_chars, "suffix_table_bytes_per_char": table_bytes / n_text,
                                "window_directory_bytes_per_char": window_bytes / n_text,
                                "resident_bytes_per_char": (image_bytes + table_bytes + window_bytes) / n_text})
        del ser
        log("[series] sampleRate %d: built %.1fs, resident %.1fs, image %.3f B/char, table %d chars"
            % (s, t_build, t_dev, image_bytes / n_text, table_chars))
        # what the suffix table answers of every pattern's backward search: all steps of the last `table_chars` characters
        def table_part(n_q):
            """(algorithmic bytes, LF-steps) the table answers for the first n_q queries"""
            if not table_chars:
                return 0, 0
            ends = off[1: n_q + 1].astype(np.int64)
            idx = (ends[:, None] - table_chars + np.arange(table_chars)[None, :]).reshape(-1)
            tail = np.ascontiguousarray(pat[idx])
            orc.counters_reset()
            ref.count_batch(tail, (np.arange(n_q + 1, dtype=np.int64) * table_chars).astype(np.int32), threads=cores)
            c = orc.counters()
            return c["alg_bytes"], c["lf_steps"]

        table_alg, table_steps = table_part(Q) if check else (0, 0)

        def launches_only(fn):  # (counter passes: the row's call, nothing else)
            for _ in range(calls or 3):
                fn()
            torch.cuda.synchronize()

        def row(bench, mm, ms, units, c, extra, n_q=None):
            n_q = Q if n_q is None else n_q
            t_alg, t_steps = (table_alg, table_steps) if n_q == Q else table_part(n_q)
            executed_alg = c["alg_bytes"] - (t_alg if bench != "extract" else 0)
            executed_steps = c["lf_steps"] - (t_steps if bench != "extract" else 0)
            r = {"benchmark": bench, "sample_rate": s, "queries": n_q, "ms_per_batch": ms, "ops_per_s": n_q / ms * 1e3,
                 "lf_steps_reference": c["lf_steps"], "lf_steps_executed": executed_steps,
                 "lf_steps_per_s_executed": executed_steps / ms * 1e3,
                 "alg_bytes_per_lf_step": c["alg_bytes"] / max(1, c["lf_steps"]),
                 "roofline": {"bound": "hbm", "achieved": executed_alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": executed_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "frac_reference_equivalent": c["alg_bytes"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                 "published_reference_ops_per_s_1core_xeon": PUBLISHED.get((bench, mm, s))}
            if mm is not None:
                r["max_matches" if bench == "locate" else "chars"] = mm
            r["key"] = row_key(bench, mm, s)
            settle_frac(r["roofline"], ms, traffic_lookup(r["key"], n_q) if traffic_lookup else None)
            r.update(units)
            r.update(extra)
            rows.append(r)
            rf = r["roofline"]
            log("[series] s=%d %s%s: %.3f ms, %.3g ops/s, frac %s (algorithmic %.2f, counter traffic %s)" % (
                s, bench, "" if mm is None else "(%d)" % mm, ms, r["ops_per_s"], "-" if rf["frac"] is None else "%.2f" % rf["frac"],
                rf["frac_algorithmic"], "-" if rf["traffic_frac"] is None else "%.2f" % rf["traffic_frac"]))

        # ---- countBenchmark ----
        def count(with_lf):
            check_rc(ia.lib.fmx_count_batch_dev(fm.handle, d_pat.data_ptr(), d_off.data_ptr(), Q, d_cnt.data_ptr(),
                                                d_lf.data_ptr() if with_lf else None, d_st.data_ptr() if with_lf else None, sp),
                     "fmx_count_batch_dev")

        if ("count", None) in wanted and not check:
            launches_only(lambda: count(False))
        elif ("count", None) in wanted:
            count(True)
            torch.cuda.synchronize()
            orc.counters_reset()
            oc, ost = ref.count_batch(pat, off, threads=cores)
            c = orc.counters()
            if not ((d_cnt.cpu().numpy() == oc).all() and (d_st.cpu().numpy() == ost).all()
                    and int(d_lf.sum(dtype=torch.int64).item()) == c["lf_steps"]):
                raise RuntimeError("count differs from the oracle (sampleRate %d)" % s)
            row("count", None, _timed(torch, stream, lambda: count(False), 10), {"matches": int(oc.astype(np.int64).sum())}, c,
                {"checked_vs_oracle": "all %d counts, statuses, LF-step total" % Q})

        # ---- locateBenchmark ----
        for mm in [w[1] for w in wanted if w[0] == "locate"]:
            # (sampleRate 1 has no walks to check: its rows keep every query)
            Qm = series_queries(Q, s, "locate", mm, bounded)
            d_locs = torch.zeros(Qm * mm, dtype=torch.int32, device=dev)

            def locate(with_lf):
                check_rc(ia.lib.fmx_locate_batch_dev(fm.handle, d_pat.data_ptr(), d_off.data_ptr(), Qm, mm, d_locs.data_ptr(), mm,
                                                     d_found.data_ptr(), d_lf.data_ptr() if with_lf else None, d_st.data_ptr(),
                                                     d_rng.data_ptr(), sp), "fmx_locate_batch_dev")

            if not check:
                launches_only(lambda: locate(False))
                del d_locs
                continue
            d_lf.zero_()
            d_st.zero_()
            locate(True)
            torch.cuda.synchronize()
            found = d_found.cpu().numpy()[:Qm]
            locs = d_locs.cpu().numpy().reshape(Qm, mm)
            orc.counters_reset()
            olocs, ofound, ost = ref.locate_batch(pat[: off[Qm]], off[: Qm + 1], mm, threads=cores)
            c = orc.counters()
            live = np.arange(mm)[None, :] < found[:, None]
            if not ((found == ofound).all() and (locs[live] == olocs[live]).all() and int(d_st[:Qm].max().item()) == 0
                    and int(d_lf[:Qm].sum(dtype=torch.int64).item()) == c["lf_steps"]):
                raise RuntimeError("locate differs from the oracle (sampleRate %d, maxMatches %d)" % (s, mm))
            hits = int(found.astype(np.int64).sum())
            del olocs, locs, live
            row("locate", mm, _timed(torch, stream, lambda: locate(False), 3 if mm >= 100 else 5), {"hits": hits}, c,
                {"checked_vs_oracle": "all %d queries: found, every position in SA order (%d hits), LF-step total" % (Qm, hits)},
                n_q=Qm)
            row_ms = rows[-1]["ms_per_batch"]
            rows[-1]["hits_per_s"] = hits / row_ms * 1e3
            del d_locs

        # ---- extractBenchmark: extract(start, start + 32) ----
        def extract(with_lf):
            check_rc(ia.lib.fmx_extract_batch_dev(fm.handle, d_start.data_ptr(), d_stop.data_ptr(), Q, d_dst.data_ptr(), xl, 0,
                                                  d_len.data_ptr(), d_lf.data_ptr() if with_lf else None, d_st.data_ptr(), sp),
                     "fmx_extract_batch_dev")

        if not any(w[0] == "extract" for w in wanted) or not check:
            if any(w[0] == "extract" for w in wanted):
                launches_only(lambda: extract(False))
            fm.close()
            del ref
            continue
        extract(True)
        torch.cuda.synchronize()
        orc.counters_reset()
        odst, olen, ost = ref.extract_batch(starts, starts + xl, xl, threads=cores)
        c = orc.counters()
        dst = d_dst.cpu().numpy().view(np.uint16).reshape(Q, xl)
        want = text[starts[:, None].astype(np.int64) + np.arange(xl)[None, :]]
        # (rows may differ from the TEXT: inverseSelect masks a run block's symbol to 8 bits, WFBB:1332 — on an alphabet
        # above 256 codes the reference itself extracts other characters there, and so do the oracle and the kernels)
        q1_rows = int((odst != want).any(1).sum())
        if not ((dst == odst).all() and (d_len.cpu().numpy() == olen).all()
                and (d_st.cpu().numpy() == ost).all() and int(d_lf.sum(dtype=torch.int64).item()) == c["lf_steps"]):
            raise RuntimeError("extract differs from the oracle (sampleRate %d): rows != oracle %d, rows != text %d, oracle != text %d, "
                               "lengths %d, statuses %d (gpu max %d, oracle max %d), LF-steps %d vs %d" % (
                                   s, int((dst != odst).any(1).sum()), int((dst != want).any(1).sum()), int((odst != want).any(1).sum()),
                                   int((d_len.cpu().numpy() != olen).sum()), int((d_st.cpu().numpy() != ost).sum()),
                                   int(d_st.max().item()), int(ost.max()), int(d_lf.sum(dtype=torch.int64).item()), c["lf_steps"]))
        row("extract", xl, _timed(torch, stream, lambda: extract(False), 5), {"chars_per_s": None}, c,
            {"checked_vs_oracle": "all %d rows, lengths, statuses, LF-step total" % Q,
             "rows_where_the_reference_differs_from_the_text": q1_rows,
             "note": None if q1_rows == 0 else "WFBB:1332 masks the symbol of a run block to 8 bits: with more than 256 codes the "
                                               "reference extracts other characters there; reproduced bit for bit (docs/DESIGN_HISTORY.md Q1)"})
        rows[-1]["chars_per_s"] = Q * xl / rows[-1]["ms_per_batch"] * 1e3
        fm.close()
        del ref
    info["rows"] = rows
    return info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--text-log2", type=int, default=28)
    ap.add_argument("--queries", type=int, default=1 << 20)
    ap.add_argument("--symbols", type=int, default=None)
    ap.add_argument("--sample-rates", default="1,32,64")
    ap.add_argument("--max-matches", default="1,10,100,1000")
    ap.add_argument("--full", action="store_true", help="every row over all the queries (maxMatches 1000 at sampleRate 64: minutes of oracle time)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "ref_series.json"))
    args = ap.parse_args()
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch

    import index4j_amd as ia
    import orc

    out = run_series(ia, torch, orc, torch.device("cuda", 0), args.text_log2, args.queries,
                     tuple(int(x) for x in args.sample_rates.split(",")), tuple(int(x) for x in args.max_matches.split(",")),
                     args.symbols, log=lambda *a: print(*a, file=sys.stderr, flush=True), bounded=not args.full)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    for i in out["indexes"]:
        print("s=%-3d image %.3f + table %.3f B/char (table: %d chars), serialized %.3f; build %.2fs, to device %.2fs" % (
            i["sample_rate"], i["image_bytes_per_char"], i["suffix_table_bytes_per_char"], i["suffix_table_chars"],
            i["serialized_bytes_per_char"], i["build_s"], i["flatten_upload_table_s"]))
    for r in out["rows"]:
        print("s=%-3d %-8s %-5s %7d q %9.3f ms  %10.4g ops/s  (published %s)  executed %.3g LF-steps/s  frac %.3f" % (
            r["sample_rate"], r["benchmark"], r.get("max_matches", r.get("chars", "")), r["queries"], r["ms_per_batch"], r["ops_per_s"],
            r["published_reference_ops_per_s_1core_xeon"], r["lf_steps_per_s_executed"], r["roofline"]["frac_algorithmic"]))


if __name__ == "__main__":
    main()
