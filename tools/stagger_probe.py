#!/usr/bin/env python3
"""Probe (GPU box): do batches in flight stay staggered when they START staggered?  Three lanes; the first three calls after an idle
point are issued `gap` microseconds apart (host sleep), then back to back.  python tools/stagger_probe.py [--gaps 0,100,200]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import gbnns_dim_red_amd as g  # noqa: E402
from gbnns_dim_red_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaps", default="0,100,200,300")
    ap.add_argument("--ef", type=int, default=64)
    ap.add_argument("--depth", type=int, default=3)
    ap.add_argument("--reps", type=int, default=150)
    args = ap.parse_args()
    g.load_library()
    ds = synth.make_dataset(n=1_000_000, nq=10_000, d=128, d_low=32, d_hidden=256, seed=1234, device="cuda:0",
                            cache_dir=os.environ.get("GBNNS_CACHE", "/tmp/gbnns_cache"))
    ix = ds.index()
    qs = [ds.queries] + [synth.more_queries(ds, ds.nq, batch=j) for j in range(1, 4)]
    for _ in range(8):
        ix.search(qs[0], args.ef, want=())
    depth = args.depth
    outs = [{} for _ in range(depth)]

    def step(i):
        ix.search(qs[i & 3], args.ef, want=(), out=outs[i % depth], flags=g.FLAG_DEFER_JOIN, defer_depth=depth)

    for gap in [int(x) for x in args.gaps.split(",")]:
        for rep in range(2):
            for i in range(12):
                step(i)
            ix.join(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.reps):
                step(i)
                if i < depth - 1 and gap:
                    t1 = time.perf_counter()
                    while time.perf_counter() - t1 < gap * 1e-6:
                        pass
            ix.join(); torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.reps
            print("initial gap %4d us: %.4f ms per batch = %.2f M queries/s" % (gap, dt * 1e3, ds.nq / dt / 1e6), flush=True)


if __name__ == "__main__":
    main()
