#!/usr/bin/env python3
"""stdin = bench.py's stdout -> stdout = the DETAIL record (the `BENCH_DETAIL {...}` line before the compact contract
line).  For scripts that want more than the contract line: `python bench.py ... | python tools/bench_detail.py | ...`"""
import sys

detail = None
for ln in sys.stdin:
    if ln.startswith("BENCH_DETAIL "):
        detail = ln[len("BENCH_DETAIL "):]
if detail is None:
    sys.exit("no BENCH_DETAIL line on stdin")
sys.stdout.write(detail)
