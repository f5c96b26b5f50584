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
