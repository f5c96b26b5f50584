"""fmx_count_batch with host arrays, call by call (pageable, then registered): where a call's time goes (FMX_PIPE_TIMING=1
prints the pipeline's phases to stderr).  GPU box only."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import index4j_amd as ia
from index4j_amd import workload
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 26
text = ia.synth_log(1 << lg)
fm = ia.FmIndex(text, 32, True, device=0, build_device=0)
n = 1 << 20
if os.environ.get("FMX_CHUNK"):
    assert ia.lib.fmx_set_option(b"host_pipeline_chunk", int(os.environ["FMX_CHUNK"])) == 0
if os.environ.get("FMX_MAPPED"):
    assert ia.lib.fmx_set_option(b"host_mapped", int(os.environ["FMX_MAPPED"])) == 0
if os.environ.get("FMX_DIRECT"):
    assert ia.lib.fmx_set_option(b"host_direct_stores", int(os.environ["FMX_DIRECT"])) == 0
pat, off, pos = workload.count_batch_patterns(text, n, 8)
counts = np.zeros(n, np.int32); status = np.zeros(n, np.int32)
def call():
    t0 = time.perf_counter()
    rc = ia.lib.fmx_count_batch(fm.handle, pat.ctypes.data, off.ctypes.data, n, counts.ctypes.data, None, status.ctypes.data)
    assert rc == 0
    return (time.perf_counter() - t0) * 1e3
print("pageable  ", " ".join("%.3f" % call() for _ in range(12)), file=sys.stderr)
for a in (pat, off, counts, status): ia.lib.fmx_host_register(a.ctypes.data, a.nbytes)
print("registered", " ".join("%.3f" % call() for _ in range(12)), file=sys.stderr)
ref, _ = fm.count_batch(pat, off)
assert (ref == counts).all() and (status == 0).all()
print("results equal", file=sys.stderr)
