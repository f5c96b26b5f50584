import sys, ctypes as C, numpy as np
sys.path.insert(0,'/root/repo')
import torch, bench, index4j_amd as ia
text, fm, path = bench.build_or_load_index(ia, 28, 32, "/tmp/fmx_cache")
fm.to_device(0)
dev=torch.device("cuda",0); n=1<<20; sp=C.c_void_p(torch.cuda.current_stream().cuda_stream)
pat,off,_=ia.synth_patterns(text,8,n,seed=43)
p=torch.from_numpy(pat.view(np.int16)).to(dev); o=torch.from_numpy(off).to(dev); c=torch.zeros(n,dtype=torch.int32,device=dev)
plan=C.c_void_p()
def timed(fn,reps=40):
    for i in range(6): fn()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); e0.record()
    for i in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/reps
for sa in (0,1,2):
    ia.lib.fmx_set_option(b"plan_sa_key", sa)
    tp=timed(lambda: ia.lib.fmx_count_plan_dev(fm.handle,p.data_ptr(),o.data_ptr(),n,C.byref(plan),sp))
    ia.lib.fmx_count_plan_dev(fm.handle,p.data_ptr(),o.data_ptr(),n,C.byref(plan),sp)
    tc=timed(lambda: ia.lib.fmx_count_ordered_dev(fm.handle,p.data_ptr(),o.data_ptr(),plan,n,c.data_ptr(),None,None,sp))
    print("sa_key %d: plan %.4f ms  k_count %.4f ms"%(sa,tp,tc))
