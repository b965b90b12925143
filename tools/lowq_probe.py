#!/usr/bin/env python3
"""Probe (GPU box): how much of the in-flight step is the projection's place in the pipeline?  The same batches in flight
(GBNNS_FLAG_DEFER_JOIN) with the projection done beforehand (MODE_LOWQ: precomputed low-dim queries, the walk kernels
are ready the moment they are enqueued) against the ordinary two-stage call, and against one batch at a time.
  python tools/lowq_probe.py [ef] [depth]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gbnns_dim_red_amd as g  # noqa: E402
from gbnns_dim_red_amd import synth  # noqa: E402

ef = int(sys.argv[1]) if len(sys.argv) > 1 else 64
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 3
g.load_library()
ds = synth.make_dataset(n=1_000_000, nq=10_000, d=128, d_low=32, d_hidden=256, seed=1234, device="cuda:0",
                        cache_dir=os.environ.get("GBNNS_CACHE", "/tmp/gbnns_cache"))
ix = ds.index()
batches = [ds.queries] + [synth.more_queries(ds, ds.nq, batch=j) for j in range(1, 4)]
lows = [ix.search(b, ef, want=("q_low",))["q_low"].clone() for b in batches]
torch.cuda.synchronize()


def run(mode, flight, steps=120):
    outs = [{} for _ in range(max(depth, 2))]
    def step(i):
        kw = dict(want=(), out=outs[i % len(outs)])
        if flight:
            kw.update(flags=g.FLAG_DEFER_JOIN, defer_depth=depth)
        if mode == "lowq":
            ix.search(batches[i % 4], ef, mode=g.MODE_LOWQ, queries_low=lows[i % 4], **kw)
        else:
            ix.search(batches[i % 4], ef, **kw)
    for i in range(12):
        step(i)
    ix.join(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    ix.join(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print("ef %d %-5s %-9s %.4f ms per 10k batch = %.2f M queries/s" % (ef, mode, "in flight" if flight else "serial", dt * 1e3, 1e-2 / dt), flush=True)


for mode in ("net", "lowq"):
    for flight in (False, True):
        run(mode, flight)
        run(mode, flight)
ix.close()
