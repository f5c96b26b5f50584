import sys, time, numpy as np
sys.path.insert(0,'/root/repo')
import index4j_amd as ia
from index4j_amd import workload
text=workload.log_text(26)
fm=ia.FmIndex(text,32,True,device=0,build_device=0)
pat,off,pos=workload.count_batch_patterns(text,1<<20,8)
n=1<<20
counts=np.zeros(n,np.int32); status=np.zeros(n,np.int32)
def call():
    t0=time.perf_counter()
    rc=ia.lib.fmx_count_batch(fm.handle,pat.ctypes.data,off.ctypes.data,n,counts.ctypes.data,None,status.ctypes.data)
    assert rc==0
    return (time.perf_counter()-t0)*1e3
for i in range(4): print("pageable %.3f ms"%call(), file=sys.stderr)
for a in (pat,off,counts,status): ia.lib.fmx_host_register(a.ctypes.data,a.nbytes)
for i in range(4): print("registered %.3f ms"%call(), file=sys.stderr)
