import sys, ctypes as C, numpy as np
sys.path.insert(0,'/root/repo')
import torch, bench, index4j_amd as ia
text, fm, path = bench.build_or_load_index(ia, 28, 32, "/tmp/fmx_cache")
dev=torch.device("cuda",0); n=1<<20; sp=C.c_void_p(torch.cuda.current_stream().cuda_stream)
bs=[]
for b in range(4):
    pat,off,_=ia.synth_patterns(text,8,n,seed=43+b)
    bs.append((torch.from_numpy(pat.view(np.int16)).to(dev), torch.from_numpy(off).to(dev), torch.zeros(n,dtype=torch.int32,device=dev)))
def step(i):
    p,o,c=bs[i%4]; assert ia.lib.fmx_count_batch_dev(fm.handle,p.data_ptr(),o.data_ptr(),n,c.data_ptr(),None,None,sp)==0
def timed():
    for i in range(6): step(i)
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); e0.record()
    for i in range(40): step(i)
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/40
for depth in (0,2,3,4,5,6):
    ia.lib.fmx_set_option(b"suffix_table_mb", 0 if depth==0 else 4096); ia.lib.fmx_set_option(b"suffix_table_image_fraction",0); ia.lib.fmx_set_option(b"suffix_table_chars", max(2,depth))
    fm.to_device(0)
    ia.lib.fmx_set_option(b"sort_min", 16384); a=timed()
    ia.lib.fmx_set_option(b"sort_min", 1<<30); b=timed()
    print("table depth %d: planned %.4f ms, caller's order %.4f ms"%(fm.suffix_table_info()[0],a,b), flush=True)
