# ad hoc check of a big index (default 2^29 chars = 513 superblocks: beyond the LDS header cache): build on the GPU,
# count / locate / extractUntilBoundary properties and an oracle sample.  usage: python tools/check_big_index.py [log2]
import sys, time, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import index4j_amd as ia, orc
n = (2**31 - 2) if (len(sys.argv) > 1 and sys.argv[1] == "max") else 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 29)
t0=time.time(); t = ia.synth_log(n); print("text", time.time()-t0, flush=True)
t0=time.time(); fm = ia.FmIndex(t, 32, True, device=0, build_device=0); print("build+to_device", time.time()-t0, fm.build_stats, flush=True)
print("window directory MB", fm.window_cells_bytes() / 1e6, "suffix table", fm.suffix_table_info(), flush=True)
# (the orders of round 4 forced on: the plan by SA row for every batch of sort_min patterns, the locate walk by range start)
assert ia.lib.fmx_set_option(b"plan_sa_min", 0) == 0 and ia.lib.fmx_set_option(b"walk_order_min", 1) == 0
pat, off, pos = ia.synth_patterns(t, 8, 200000)
t0=time.time(); cnt, st, lf = fm.count_batch(pat, off, want_steps=True); print("count", time.time()-t0, flush=True)
assert (st==0).all() and (cnt>=1).all() and (lf==14).all()
locs, found, st2 = fm.locate_batch(pat[:8*20000], off[:20001], 8)
P = pat.reshape(-1,8)
for k in range(8):
    sel = found > k
    idx = locs[sel, k].astype(np.int64)
    assert (t[idx[:,None]+np.arange(8)[None,:]] == P[:20000][sel]).all()
dst, ol, st3, aux = fm.extract_boundary_batch(locs[:2000,0], "\n", 0, 1024)
nl = np.flatnonzero(t == 10)
for i in range(2000):
    p = int(locs[i,0]); j = np.searchsorted(nl, p)
    if j < len(nl) and t[p] != 10:
        lo = nl[j-1]+1 if j>0 else 0
        assert st3[i]==0 and ol[i]==nl[j]-lo and (dst[i,:ol[i]]==t[lo:nl[j]]).all(), i
ser = fm.write(False); print("serialized MB", len(ser)/1e6, flush=True)
o = orc.OracleFmIndex.read(ser)
oc, _ = o.count_batch(pat[:8*3000], off[:3001], threads=8)
assert (oc == cnt[:3000]).all()
print("big index ok: n_sb =", (n+1+(1<<20)-1)>>20)
