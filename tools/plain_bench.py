#!/usr/bin/env python3
"""GPU box: the plain-graph baseline of final_test.cpp:84 (performRealTests: walk in the ORIGINAL space, k = 1) on
the bench workload -- kernel time per 10 k batch for a few of the reference's efs_hnsw values."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gbnns_dim_red_amd as g
from gbnns_dim_red_amd import synth
g.load_library()
ds = synth.make_dataset(device="cuda:0", cache_dir=os.environ.get("GBNNS_CACHE", "/tmp/gbnns_cache"), n=1_000_000,
                        nq=10_000, d=128, d_low=32, d_hidden=256, seed=1234)
ix = ds.index()
for ef in (15, 40, 64, 100, 140):
    for _ in range(3):
        r = ix.search(ds.queries, ef, mode=g.MODE_PLAIN, k=1, want=("hops", "dist_calc"))
    torch.cuda.synchronize()
    ix.profile_read(reset=True); ix.profile_enable(True)
    t1 = time.perf_counter()
    for _ in range(5):
        r = ix.search(ds.queries, ef, mode=g.MODE_PLAIN, k=1, want=("hops", "dist_calc"))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t1) / 5
    p = ix.profile_read(reset=True); ix.profile_enable(False)
    rec = (r["ids"].long() == ds.gt).float().mean().item()
    dc = r["dist_calc"].float().mean().item()
    print("plain d=128 ef=%d: recall %.4f  %.3f ms/batch  %.2f M q/s  walk %.3f ms  dist_calc %.0f  -> %.0f GB/s of row bytes"
          % (ef, rec, dt * 1e3, ds.nq / dt / 1e6, p["walk_ms"] / p["calls"], dc, dc * 512 * ds.nq / (p["walk_ms"] / p["calls"] * 1e-3) / 1e9), flush=True)
ix.close()
