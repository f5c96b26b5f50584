"""bench.py's contract line: whatever the detail record holds (the 18-row series, host buffers, 8 ranks), the LAST stdout
line stays below 4 KB and carries the driver's fields.  Round 3's single 21.7 KB line outgrew the driver's 8,000-character
tail and went unparsed; the detail records of that round are the worst cases on file."""
import copy
import glob
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


RECORDS = sorted(glob.glob(os.path.join(ROOT, "profiles", "r03_f_*.json")) + glob.glob(os.path.join(ROOT, "profiles", "r04_*bench*.json")))


@pytest.mark.parametrize("path", RECORDS, ids=[os.path.basename(p) for p in RECORDS])
def test_contract_line_of_every_recorded_detail_is_small_and_complete(path):
    b = _bench()
    text = open(path).read().strip().splitlines()
    recs = []
    for ln in text:
        if ln.startswith("{") or ln.startswith("BENCH_DETAIL "):
            try:
                recs.append(json.loads(ln[len("BENCH_DETAIL "):] if ln.startswith("BENCH_DETAIL ") else ln))
            except ValueError:
                pass  # a pretty-printed file, not a line record
    recs = [r for r in recs if isinstance(r, dict) and "metric" in r and "detail" not in r]
    if not recs:
        pytest.skip("not a bench record")
    for rec in recs:
        line = b.compact_line(rec)
        assert len(line) < b.COMPACT_LIMIT
        c = json.loads(line)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                  "vs_baseline", "dtype", "data"):
            assert c[k] == rec[k]
        assert c["config"]["workload"]
        if rec.get("roofline"):
            for k in ("bound", "achieved", "peak", "unit", "frac"):
                assert c["roofline"][k] == rec["roofline"][k]
            assert "traffic" in c["roofline"]
        if rec.get("cpu_baseline"):
            for k in ("value", "unit", "cores", "kind"):
                assert c["cpu_baseline"][k] == rec["cpu_baseline"][k]
            assert c["cpu_baseline"]["sample"]


def test_contract_line_sheds_optional_blocks_before_it_grows():
    b = _bench()
    rec = json.loads(open(os.path.join(ROOT, "profiles", "r03_f_bench_unprofiled.json")).read().strip().splitlines()[-1])
    big = copy.deepcopy(rec)
    big["n_gpus"] = 8
    big["ranks_seen"] = [[r, r, r] for r in range(8)]
    big["secondary"] = big["secondary"] * 4  # 4 x (2 configs + 18 series rows)
    line = b.compact_line(big)
    assert len(line) < b.COMPACT_LIMIT
    c = json.loads(line)
    assert c["value"] == rec["value"] and c["roofline"]["frac"] == rec["roofline"]["frac"]


def _ref_series():
    spec = importlib.util.spec_from_file_location("ref_series_mod", os.path.join(ROOT, "tools", "ref_series.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_no_frac_above_one_leaves_the_rule():
    """a `frac` is a fraction.  The algorithmic figure of locate passed 1.0 in round 4 (hits of equal ranges
    share their lines); the rule reports the counter figure there and keeps the algorithmic one under its own name."""
    rs = _ref_series()
    ms = 0.4
    # algorithmic 1.003, counters 0.55 -> frac = the counter figure
    alg_bytes = 1.003 * rs.HBM_PEAK_GBS * 1e9 * ms * 1e-3
    traffic = 0.55 * rs.HBM_PEAK_GBS * 1e9 * ms * 1e-3
    r = rs.settle_frac({"frac": 1.003, "achieved": alg_bytes / (ms * 1e-3) / 1e9}, ms, traffic)
    assert abs(r["frac"] - 0.55) < 1e-9 and abs(r["frac_algorithmic"] - 1.003) < 1e-9 and abs(r["traffic_frac"] - 0.55) < 1e-9
    assert r["frac"] <= 1.0 and r["traffic"] == traffic
    # algorithmic 0.68, counters 0.27: more than 1.5 x apart -> the counter figure
    r = rs.settle_frac({"frac": 0.68, "achieved": 0.68 * rs.HBM_PEAK_GBS}, ms, 0.27 * rs.HBM_PEAK_GBS * 1e9 * ms * 1e-3)
    assert abs(r["frac"] - 0.27) < 1e-9 and r["frac_algorithmic"] == 0.68
    # algorithmic 0.71, counters 0.62: the algorithmic figure stands
    r = rs.settle_frac({"frac": 0.71, "achieved": 0.71 * rs.HBM_PEAK_GBS}, ms, 0.62 * rs.HBM_PEAK_GBS * 1e9 * ms * 1e-3)
    assert r["frac"] == 0.71 and abs(r["traffic_frac"] - 0.62) < 1e-9
    # no counters on file: a figure above 1 is withheld, one below stays
    r = rs.settle_frac({"frac": 1.2, "achieved": 1.2 * rs.HBM_PEAK_GBS}, ms, None)
    assert r["frac"] is None and r["frac_algorithmic"] == 1.2 and r["traffic_frac"] is None
    r = rs.settle_frac({"frac": 0.5, "achieved": 0.5 * rs.HBM_PEAK_GBS}, ms, None)
    assert r["frac"] == 0.5


def test_secondary_rows_of_the_contract_line_carry_both_fractions():
    b = _bench()
    rs = _ref_series()
    rec = json.loads(open(os.path.join(ROOT, "profiles", "r03_f_bench_unprofiled.json")).read().strip().splitlines()[-1])
    rec = copy.deepcopy(rec)
    rows = []
    for s, wanted in rs.DEFAULT_PLAN:
        for bench, mm in wanted:
            roof = rs.settle_frac({"frac": 1.1, "achieved": 1.1 * rs.HBM_PEAK_GBS}, 1.0, 0.5 * rs.HBM_PEAK_GBS * 1e9 * 1e-3)
            rows.append({"benchmark": bench, "sample_rate": s, "key": rs.row_key(bench, mm, s), "ms_per_batch": 1.0, "roofline": roof})
    rec["secondary"] = [{"config": "configs[2]", "ms": 0.4, "roofline": rs.settle_frac({"frac": 1.003, "achieved": 1.0}, 0.4, 1.7e9)},
                        {"config": "reference_series", "series": {"rows": rows}}]
    line = b.compact_line(rec)
    assert len(line) < b.COMPACT_LIMIT
    c = json.loads(line)
    assert len(c["secondary"]) == 1 + 6
    for row in c["secondary"]:
        assert row["frac"] is None or row["frac"] <= 1.0
        assert "alg" in row and "tfrac" in row
    assert [r["config"] for r in c["secondary"][1:]] == [rs.row_key(bn, mm, s) for s, w in rs.DEFAULT_PLAN for bn, mm in w]
