#!/usr/bin/env python3
"""Tuning tool: first-pass walk time of a bench workload by visited-set capacity (explicit hash_capacity) -- what a smaller
table (more wavefronts per CU) buys before any hand-over cost.  python tools/cap_probe.py --ef 140 --caps 0,2800,2400,2100,1880"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import gbnns_dim_red_amd as g  # noqa: E402
from gbnns_dim_red_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="sift")
    ap.add_argument("--ef", type=int, default=140)
    ap.add_argument("--caps", default="0,2800,2400,2100,1880")
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    cfg = bench.CONFIGS[args.config]
    g.load_library()
    kw = dict(n=cfg["n"], nq=cfg["nq"], d=cfg["d"], d_low=cfg["d_low"], d_hidden=cfg["d_hidden"], seed=1234,
              cache_dir="/tmp/gbnns_cache")
    if cfg.get("unit_norm"):
        kw["unit_norm"] = True
    if cfg.get("native_knn") and cfg["n"] > 2_000_000:
        kw["native_knn"] = True
    if cfg.get("strong"):
        kw["gt_queries"] = 20_000
    os.makedirs("/tmp/gbnns_cache", exist_ok=True)
    ds = synth.make_dataset(device="cuda:0", **kw)
    ix = ds.index()
    ref = None
    for cap in [int(x) for x in args.caps.split(",")]:
        for _ in range(4):
            r = ix.search(ds.queries, args.ef, hash_capacity=cap)
        torch.cuda.synchronize()
        ix.profile_enable(True)
        for _ in range(args.reps):
            r = ix.search(ds.queries, args.ef, hash_capacity=cap)
        torch.cuda.synchronize()
        p = ix.profile_read()
        ix.profile_enable(False)
        ids = r["ids"].cpu().numpy()
        dc = r["dist_calc"].cpu().numpy()
        if ref is None:
            ref = ids
            print("dist_calc quantiles 50/90/99/max:", np.percentile(dc, [50, 90, 99]).round(0), dc.max())
        print("cap %5d: walk %.4f ms  retry+general %.4f ms  handed over %d  total %.4f  %s  ids %s" % (
            cap, p["walk_ms"] / p["calls"], p["walk_general_ms"] / p["calls"], p["general_queries"], p["total_ms"] / p["calls"],
            p["walk_kernel"][:40], "same" if (ids == ref).all() else "DIFFER"), flush=True)


if __name__ == "__main__":
    main()
