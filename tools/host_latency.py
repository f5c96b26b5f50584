"""What ONE query costs through the host entry points (a Java caller's count(char[]) / locate(...) is a batch of one): microseconds
per call of fmx_count_batch / fmx_locate_batch / fmx_extract_batch with n = 1 and n = 1000, host arrays.  GPU box only."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import index4j_amd as ia

t = ia.synth_log(1 << 24)
fm = ia.FmIndex(t, 32, True, device=0, build_device=0)
pat, off, pos = ia.synth_patterns(t, 8, 1000)


def per_call(f, reps=300):
    f()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    return (time.perf_counter() - t0) / reps * 1e6


for n in (1, 1000):
    p, o = pat[:n * 8].copy(), off[:n + 1].copy()
    print("count   n=%4d: %7.1f us per call" % (n, per_call(lambda: fm.count_batch(p, o))))
    print("locate  n=%4d: %7.1f us per call (maxMatches 16)" % (n, per_call(lambda: fm.locate_batch(p, o, 16))))
    a = np.arange(n, dtype=np.int32) * 1000
    print("extract n=%4d: %7.1f us per call (64 characters)" % (n, per_call(lambda: fm.extract_batch(a, a + 64, 64))))
    print("locate -> extract n=%4d: %7.1f us per call (4 hits x 32 characters)" % (n, per_call(lambda: fm.locate_extract_batch(p, o, 4, 32))))
