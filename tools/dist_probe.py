#!/usr/bin/env python3
"""GPU box: how wide is the dist_calc distribution of a batch (the visited set is sized for the longest walk seen) on synthetic workloads of
other recipes than the bench's -- quantiles, the sizing the library chose, first-pass kernel time.
usage: dist_probe.py [--intrinsic 24 --clusters 100 --scale 1.0 --sigma 0.1] [--efs 64,160]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gbnns_dim_red_amd as g
from gbnns_dim_red_amd import synth
ap = argparse.ArgumentParser()
ap.add_argument("--intrinsic", type=int, default=16); ap.add_argument("--clusters", type=int, default=1000)
ap.add_argument("--scale", type=float, default=0.5); ap.add_argument("--sigma", type=float, default=0.03)
ap.add_argument("--M", type=int, default=16); ap.add_argument("--efs", default="64,160")
a = ap.parse_args()
g.load_library()
ds = synth.make_dataset(device="cuda:0", n=1_000_000, nq=10_000, d=128, d_low=32, d_hidden=256, seed=1234, intrinsic=a.intrinsic,
                        n_clusters=a.clusters, cluster_scale=a.scale, sigma=a.sigma, M=a.M, cache_dir="/tmp/gbnns_cache")
ix = ds.index()
q = ds.queries
print("recipe: intrinsic %d, %d clusters, scale %.2f, sigma %.3f, GD(M = %d)" % (a.intrinsic, a.clusters, a.scale, a.sigma, a.M))
for ef in [int(x) for x in a.efs.split(",")]:
    for _ in range(4):
        r = ix.search(q, ef, want=("hops", "dist_calc"))
    torch.cuda.synchronize()
    ix.profile_read(reset=True); ix.profile_enable(True)
    for _ in range(5):
        r = ix.search(q, ef, want=("hops", "dist_calc"))
    torch.cuda.synchronize()
    p = ix.profile_read(reset=True); ix.profile_enable(False)
    dc = r["dist_calc"].double()
    qs = torch.quantile(dc, torch.tensor([0.5, 0.9, 0.99, 0.999, 1.0], dtype=torch.float64, device=dc.device)).tolist()
    rec = (r["ids"].long() == ds.gt).float().mean().item()
    print("ef %4d: recall %.4f  dist_calc 50/90/99/99.9/max %d / %d / %d / %d / %d (max / median %.2f)  walk %.4f ms  %s  handed to the general kernel %d"
          % (ef, rec, *[int(v) for v in qs], qs[4] / qs[0], p["walk_ms"] / p["calls"], p["walk_kernel"].split(" (")[0][:30], p["general_queries"]), flush=True)
