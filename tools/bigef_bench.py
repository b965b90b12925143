#!/usr/bin/env python3
"""GPU box: GloVe-like shape (200 -> 32, n = 1.2e6, L2) at the reference's large efs (parameters_of_databases.txt:41)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gbnns_dim_red_amd as g
from gbnns_dim_red_amd import synth
g.load_library()
ds = synth.make_dataset(device="cuda:0", cache_dir=os.environ.get("GBNNS_CACHE", "/tmp/gbnns_cache"), n=1_200_000, nq=10_000,
                        d=200, d_low=32, d_hidden=256, seed=1234)
ix = ds.index()
for ef in (300, 400, 600, 800, 1000):
    for _ in range(3):
        r = ix.search(ds.queries, ef, want=("hops", "dist_calc"))
    torch.cuda.synchronize(); ix.profile_read(reset=True); ix.profile_enable(True)
    t1 = time.perf_counter()
    for _ in range(3):
        r = ix.search(ds.queries, ef, want=("hops", "dist_calc"))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t1) / 3; p = ix.profile_read(reset=True); ix.profile_enable(False)
    print(json.dumps(dict(ef=ef, qps=round(ds.nq / dt), ms=round(dt * 1e3, 2), walk=round(p["walk_ms"] / p["calls"], 2),
                          rerank=round(p["rerank_ms"] / p["calls"], 2), dist_calc=round(r["dist_calc"].float().mean().item()),
                          general=p["general_queries"])), flush=True)
