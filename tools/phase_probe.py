#!/usr/bin/env python3
"""Probe (GPU box): small batches in flight with projections and walks in SEPARATE time slices -- `group` projections one after the
other (gbnns_project on the caller's stream), then the group's walks side by side (MODE_LOWQ, GBNNS_FLAG_DEFER_JOIN), then the
join -- against the ordinary two-stage calls in flight.  python tools/phase_probe.py [--config gist] [--group 3]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import gbnns_dim_red_amd as g  # noqa: E402
from gbnns_dim_red_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="gist")
    ap.add_argument("--ef", type=int, default=None)
    ap.add_argument("--nq", type=int, default=None)
    ap.add_argument("--groups", default="2,3,4")
    ap.add_argument("--reps", type=int, default=96)
    args = ap.parse_args()
    cfg = bench.CONFIGS[args.config]
    ef = args.ef or cfg["ef"]
    g.load_library()
    kw = dict(n=cfg["n"], nq=cfg["nq"], d=cfg["d"], d_low=cfg["d_low"], d_hidden=cfg["d_hidden"], seed=1234,
              cache_dir="/tmp/gbnns_cache")
    if cfg.get("unit_norm"):
        kw["unit_norm"] = True
    os.makedirs("/tmp/gbnns_cache", exist_ok=True)
    ds = synth.make_dataset(device="cuda:0", **kw)
    ix = ds.index()
    nq = args.nq or ds.nq
    qs = [torch.roll(ds.queries, shifts=-i * (ds.nq // 4), dims=0)[:nq].contiguous() for i in range(4)]
    ref = [ix.search(q, ef, want=())["ids"].cpu().numpy() for q in qs]
    for _ in range(8):
        ix.search(qs[0], ef, want=())
    torch.cuda.synchronize()

    def net_in_flight(depth):
        outs = [{} for _ in range(depth)]
        def step(i):
            return ix.search(qs[i & 3], ef, want=(), out=outs[i % depth], flags=g.FLAG_DEFER_JOIN, defer_depth=depth)
        for i in range(12):
            step(i)
        ix.join(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.reps):
            r = step(i)
        ix.join(); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.reps
        ok = (r["ids"].cpu().numpy() == ref[(args.reps - 1) & 3]).all()
        print("two-stage calls, %d in flight:            %.4f ms per batch = %.3f M queries/s%s" % (depth, dt * 1e3, nq / dt / 1e6, "" if ok else " WRONG"), flush=True)

    def phased(group):
        outs = [{} for _ in range(group)]
        def round_(i0):
            lows = [ix.project(qs[(i0 + k) & 3]) for k in range(group)]
            for k in range(group):
                r = ix.search(qs[(i0 + k) & 3], ef, mode=g.MODE_LOWQ, queries_low=lows[k], want=(), out=outs[k],
                              flags=g.FLAG_DEFER_JOIN, defer_depth=max(group, 2))
            ix.join()
            return r
        for i in range(4):
            round_(i * group)
        torch.cuda.synchronize()
        n = args.reps // group
        t0 = time.perf_counter()
        for i in range(n):
            r = round_(i * group)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (n * group)
        ok = (r["ids"].cpu().numpy() == ref[((n - 1) * group + group - 1) & 3]).all()
        print("%d projections, then their %d walks side by side: %.4f ms per batch = %.3f M queries/s%s" % (group, group, dt * 1e3, nq / dt / 1e6, "" if ok else " WRONG"), flush=True)

    net_in_flight(3)
    for grp in [int(x) for x in args.groups.split(",")]:
        phased(grp)
    net_in_flight(3)


if __name__ == "__main__":
    main()
