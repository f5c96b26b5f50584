import time, numpy as np, sys
sys.path.insert(0,'/root/repo')
import index4j_amd as ia
t = ia.synth_log(1<<20)
fm = ia.FmIndex(t, 32, True, device=0)
pat, off, pos = ia.synth_patterns(t, 8, 1000)
for n in (1, 1000):
    p, o = pat[:n*8], off[:n+1]
    fm.count_batch(p, o)
    t0=time.perf_counter()
    for _ in range(200): fm.count_batch(p, o)
    dt=(time.perf_counter()-t0)/200
    print("host-buffer count_batch n=%d: %.1f us per call" % (n, dt*1e6))
