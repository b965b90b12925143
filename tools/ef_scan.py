#!/usr/bin/env python3
"""Walk-kernel time vs ef on the bench workload (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gbnns_dim_red_amd import synth
ds = synth.make_dataset(device="cuda:0", cache_dir=os.environ.get("GBNNS_CACHE", "/tmp/gbnns_cache"))
ix = ds.index()
q = ds.queries
for ef in (1, 4, 8, 16, 32, 48, 64):
    for _ in range(4):
        r = ix.search(q, ef, want=("hops", "dist_calc"))
    torch.cuda.synchronize()
    ix.profile_read(reset=True); ix.profile_enable(True)
    for _ in range(10):
        r = ix.search(q, ef, want=("hops", "dist_calc"))
    torch.cuda.synchronize()
    p = ix.profile_read(reset=True); ix.profile_enable(False)
    print(f"ef {ef:3d} hops {r['hops'].float().mean().item():6.1f} dc {r['dist_calc'].float().mean().item():7.1f} walk_ms {p['walk_ms']/p['calls']:.4f} general {p['general_queries']}")
for nq in (1000, 2500, 5000, 10000):
    qq = q[:nq].contiguous()
    for _ in range(4):
        ix.search(qq, 64, want=())
    torch.cuda.synchronize()
    ix.profile_read(reset=True); ix.profile_enable(True)
    for _ in range(10):
        ix.search(qq, 64, want=())
    torch.cuda.synchronize()
    p = ix.profile_read(reset=True); ix.profile_enable(False)
    print(f"nq {nq:6d} ef 64 walk_ms {p['walk_ms']/p['calls']:.4f}")
