#!/usr/bin/env python3
"""Wall time of text -> resident index: fmx_build_on_device, flatten (fmx_blob), upload (fmx_to_device).  GPU box only."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import index4j_amd as ia

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 28
t = ia.synth_log(1 << lg)
ia.FmIndex("warm up", 4, True, device=0, build_device=0)
for rep in range(2):
    t0 = time.perf_counter()
    f = ia.FmIndex(t, 32, True, device=None, build_device=0)
    t1 = time.perf_counter()
    b = f.blob()
    t2 = time.perf_counter()
    f.to_device(0)
    t3 = time.perf_counter()
    print("2^%d chars: build %.3f s, flatten %.3f s (%d MB image), upload %.3f s, total %.3f s" % (lg, t1 - t0, t2 - t1, len(b) >> 20, t3 - t2, t3 - t0), flush=True)
    del f, b
