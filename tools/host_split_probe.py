#!/usr/bin/env python3
"""Probe: ONE synchronous host batch (page-locked buffers) as 1 / 2 / 3 / 4 blocks on the handle's lanes (GBNNS_MEM_HOST +
GBNNS_FLAG_DEFER_JOIN per block, then gbnns_index_wait(0)): does a block's copy-in under the block before it shorten the call?
python tools/host_split_probe.py [CONFIG] [EF]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gbnns_dim_red_amd as g
from gbnns_dim_red_amd import synth
import bench
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "sift"]
ef = int(sys.argv[2]) if len(sys.argv) > 2 else cfg["ef"]
kw = dict(n=cfg["n"], nq=cfg["nq"], d=cfg["d"], d_low=cfg["d_low"], d_hidden=cfg["d_hidden"], seed=1234)
ds = synth.make_dataset(device="cuda:0", cache_dir=os.environ.get("GBNNS_CACHE", "/tmp/gbnns_cache"), **kw)
ix = ds.index()
nq = len(ds.queries)
hq = ds.queries.cpu().pin_memory()
ref = ix.search(ds.queries, ef, want=())["ids"].cpu()
for nblk in (1, 2, 3, 4, 1):
    bounds = [(i * nq) // nblk for i in range(nblk + 1)]
    parts = [hq[bounds[i]:bounds[i + 1]] for i in range(nblk)]   # (views of one pinned buffer: still page-locked)
    outs = [{} for _ in range(nblk)]

    def call():
        if nblk == 1:
            ix.search(hq, ef, want=(), out=outs[0])
            return
        for i in range(nblk):
            ix.search(parts[i], ef, want=(), out=outs[i], flags=g.FLAG_DEFER_JOIN, defer_depth=min(nblk, 4))
        ix.wait(0)
    for _ in range(10):
        call()
    t = time.perf_counter()
    reps = 50
    for _ in range(reps):
        call()
    dt = (time.perf_counter() - t) / reps
    got = torch.cat([o["ids"] for o in outs])
    print("blocks %d: %.4f ms per %d-query batch = %.2f M queries/s, ids identical %s" % (nblk, dt * 1e3, nq, nq / dt / 1e6, bool((got == ref).all())), flush=True)
