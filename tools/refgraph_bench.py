#!/usr/bin/env python3
"""GPU box: the bench workload with the graph pruned as the reference does (prepare_graph.cpp:70, M = 30 -> degree
up to 60, adjacency rows of 64 slots): which walk kernel serves it and how fast."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gbnns_dim_red_amd as g
from gbnns_dim_red_amd import synth
g.load_library()
ds = synth.make_dataset(device="cuda:0", cache_dir=os.environ.get("GBNNS_CACHE", "/tmp/gbnns_cache"), n=1_000_000, nq=10_000,
                        d=128, d_low=32, d_hidden=256, seed=1234, M=30, knn_k=96, native_knn=True, verbose=True)
import numpy as np
deg = np.diff(ds.graph_off.astype(np.int64))
print("M=30 graph: avg degree %.1f max %d" % (deg.mean(), deg.max()), flush=True)
ix = ds.index()
for ef in (40, 64):
    for _ in range(6):
        r = ix.search(ds.queries, ef, want=("hops", "dist_calc"))
    torch.cuda.synchronize()
    ix.profile_read(reset=True); ix.profile_enable(True)
    t1 = time.perf_counter()
    for _ in range(10):
        r = ix.search(ds.queries, ef, want=("hops", "dist_calc"))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t1) / 10
    p = ix.profile_read(reset=True); ix.profile_enable(False)
    rec = (r["ids"].long() == ds.gt).float().mean().item()
    print(json.dumps(dict(ef=ef, recall=round(rec, 4), qps=round(ds.nq / dt), walk_ms=round(p["walk_ms"] / p["calls"], 4),
                          hops=round(r["hops"].float().mean().item(), 1), dist_calc=round(r["dist_calc"].float().mean().item(), 1))), flush=True)
