import sys, ctypes as C, numpy as np
sys.path.insert(0,'/root/repo')
import torch, bench, index4j_amd as ia
text, fm, path = bench.build_or_load_index(ia, 28, 32, "/tmp/fmx_cache")
fm.to_device(0)
dev=torch.device("cuda",0); sp=C.c_void_p(torch.cuda.current_stream().cuda_stream)
def timed(fn,reps=40):
    for i in range(6): fn(i)
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); e0.record()
    for i in range(reps): fn(i)
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/reps
import os
if os.environ.get('FMX_COARSE'): assert ia.lib.fmx_set_option(b'coarse_bits', int(os.environ['FMX_COARSE'])) == 0
for n in [int(x) for x in os.environ.get('FMX_SIZES','393216,524288,655360,786432,917504,1048576,2097152,4194304').split(',')]:
    bs=[]
    for b in range(3):
        pat,off,_=ia.synth_patterns(text,8,n,seed=43+b)
        bs.append((torch.from_numpy(pat.view(np.int16)).to(dev), torch.from_numpy(off).to(dev), torch.zeros(n,dtype=torch.int32,device=dev)))
    def step(i):
        p,o,c=bs[i%3]; assert ia.lib.fmx_count_batch_dev(fm.handle,p.data_ptr(),o.data_ptr(),n,c.data_ptr(),None,None,sp)==0
    r=[]
    for sa_min in (1<<30, 0):
        ia.lib.fmx_set_option(b"plan_sa_min", sa_min); r.append(timed(step))
    print("n %8d: caller's order %.4f ms, SA-row plan %.4f ms (%+.1f %%)"%(n,r[0],r[1],(r[1]/r[0]-1)*100), flush=True)
