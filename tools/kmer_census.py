#!/usr/bin/env python3
"""How many distinct k-mers does the synthetic log hold?  (what a sparse suffix table of depth k would have to store)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import index4j_amd as ia

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 28
t = ia.synth_log(1 << lg)
d = torch.from_numpy(t.astype(np.int64)).cuda()
for k in (3, 4, 5, 6, 7, 8):
    key = torch.zeros(len(d) - k + 1, dtype=torch.int64, device="cuda")
    for j in range(k):
        key = key * 128 + d[j:len(d) - k + 1 + j]
    n = int(torch.unique(key).numel())
    print("k = %d: %d distinct %d-mers (%.1f MB at 8 bytes each)" % (k, n, k, n * 8 / 1e6), flush=True)
    del key
