#!/usr/bin/env python3
"""Probe (not a test): does the ORDER of the queries inside a batch matter?  Workgroup b of the walk kernel runs query b
and lands on XCD b % 8 (each XCD has its own 4 MB L2).  Compares the kernel time of the bench batch in its random
order, sorted by a locality key of the projected query (sign bits of its first 12 coordinates), and sorted + dealt so
that every XCD works through one contiguous range of the sorted batch.  python tools/order_probe.py [--n N] [--ef EF]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, gbnns_dim_red_amd as g
from gbnns_dim_red_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--ef", type=int, default=64)
ap.add_argument("--native-knn", action="store_true")
args = ap.parse_args()
g.load_library()
os.makedirs("/tmp/gbnns_cache", exist_ok=True)
ds = synth.make_dataset(n=args.n, nq=10_000, d=128, d_low=32, d_hidden=256, seed=1234, cache_dir="/tmp/gbnns_cache",
                        native_knn=args.n > 2_000_000, device="cuda:0")
ix = ds.index()
q = ds.queries
ql = ix.project(q)
key = ((ql[:, :12] > 0).long() * (2 ** torch.arange(12, device=q.device))).sum(1)
order = torch.argsort(key, stable=True)
nq = q.shape[0]
b = torch.arange(nq, device=q.device)
dealt = order[((b % 8) * (nq // 8) + b // 8).clamp(max=nq - 1)]
variants = {"random": q, "sorted": q[order].contiguous(), "sorted+dealt": q[dealt].contiguous()}
for _ in range(6):
    for v in variants.values():
        ix.search(v, args.ef, want=())
torch.cuda.synchronize()
for name, v in variants.items():
    ix.profile_read(reset=True)
    ix.profile_enable(True)
    for _ in range(30):
        ix.search(v, args.ef, want=())
    torch.cuda.synchronize()
    p = ix.profile_read(reset=True)
    ix.profile_enable(False)
    print("%-14s walk kernel %.4f ms per launch" % (name, p["walk_ms"] / p["calls"]), flush=True)
