#!/usr/bin/env python3
"""Walk-kernel time vs wavefronts/CU (forced through the visited-set capacity) -- diagnostic."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gbnns_dim_red_amd import synth
ds = synth.make_dataset(device="cuda:0", cache_dir=os.environ.get("GBNNS_CACHE", "/tmp/gbnns_cache"))
ix = ds.index()
q = ds.queries
FIXED = 16 * 8 + 66 * 8  # hot kernel: tie list + merge buffer (the query is staged inside it)
for waves in (28, 26, 24, 22, 20, 18, 16):
    share = (160 * 1024 // waves) // 512 * 512
    cap = ((share - FIXED) // 4) & ~3
    for _ in range(3):
        ix.search(q, 64, want=(), hash_capacity=cap)
    torch.cuda.synchronize()
    ix.profile_read(reset=True); ix.profile_enable(True)
    for _ in range(10):
        ix.search(q, 64, want=(), hash_capacity=cap)
    torch.cuda.synchronize()
    p = ix.profile_read(reset=True); ix.profile_enable(False)
    print(f"waves/CU {waves:3d} cap {cap:5d} walk_ms {p['walk_ms']/p['calls']:.4f}")
