"""A/B of option boundary_order_min on the fused pipelines (tools/bench_pipeline.py: locateAndExtract and the grep pipeline, 20,000 queries x
<= 1,000 hits): hits taken by text position against the slot order.  GPU box only."""
import contextlib, io, json, os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: F401  (before the library: see tests/conftest.py)
import index4j_amd as ia
for opt in (0, 32768, 0, 32768):
    assert ia.lib.fmx_set_option(b"boundary_order_min", opt) == 0
    sys.argv = ["bench_pipeline.py", "--queries", "20000", "--max-matches", "1000", "--check", "10"]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        runpy.run_path(os.path.join(ROOT, "tools", "bench_pipeline.py"), run_name="__main__")
    for line in buf.getvalue().split("\n"):
        if line.startswith("{"):
            d = json.loads(line)
            print(opt, d["config"][:44], round(d["ms"], 3), d.get("hits"), flush=True)
