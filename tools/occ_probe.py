import sys, os, time, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench, index4j_amd as ia
text, fm, path = bench.build_or_load_index(ia, 28, 32, "/tmp/fmx_cache")
for mode in (0, 1):
    ia.lib.fmx_set_option(b"occ_cells", mode)
    t0=time.time(); fm.to_device(0); torch.cuda.synchronize(); t1=time.time()
    print("occ_cells", mode, "to_device %.3fs" % (t1-t0), "occ bytes", fm.occ_cells_bytes(), "win bytes", fm.window_cells_bytes())
    dev = torch.device("cuda", 0)
    n = 1 << 20
    pat, off, _ = ia.synth_patterns(text, 8, n, seed=43)
    d_pat, d_off = torch.from_numpy(pat.view(np.int16)).to(dev), torch.from_numpy(off).to(dev)
    d_cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def launch():
        assert ia.lib.fmx_count_batch_dev(fm.handle, d_pat.data_ptr(), d_off.data_ptr(), n, d_cnt.data_ptr(), None, None, sp) == 0
    for _ in range(3): launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): launch()
    e1.record(); torch.cuda.synchronize()
    print("  step ms", e0.elapsed_time(e1)/20, "checksum", int(d_cnt.sum(dtype=torch.int64).item()))
