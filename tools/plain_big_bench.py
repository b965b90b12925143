#!/usr/bin/env python3
"""GPU box: PLAIN walks (final_test.cpp:84, performRealTests: the original space, k = 1) at beams of more than 128 on the sift- and
deep1m-shaped workloads -- the two-list kernels over 512- / 384-byte rows.  python tools/plain_big_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import gbnns_dim_red_amd as g
from gbnns_dim_red_amd import synth
lib = g.load_library()
for name, efs in (("sift", (100, 130, 140, 200)), ("deep1m", (120, 160, 200))):
    cfg = bench.CONFIGS[name]
    ds = synth.make_dataset(device="cuda:0", cache_dir=os.environ.get("GBNNS_CACHE", "/tmp/gbnns_cache"), n=cfg["n"], nq=cfg["nq"],
                            d=cfg["d"], d_low=cfg["d_low"], d_hidden=cfg["d_hidden"], seed=1234)
    ix = ds.index()
    for ef in efs:
        for late in (0, 1):
            lib.gbnns_debug_knob(b"late_rows", late)
            for _ in range(3):
                r = ix.search(ds.queries, ef, mode=g.MODE_PLAIN, k=1, want=("hops", "dist_calc"))
            torch.cuda.synchronize()
            ix.profile_read(reset=True); ix.profile_enable(True)
            for _ in range(5):
                r = ix.search(ds.queries, ef, mode=g.MODE_PLAIN, k=1, want=("hops", "dist_calc"))
            torch.cuda.synchronize()
            p = ix.profile_read(reset=True); ix.profile_enable(False)
            dc = r["dist_calc"].float().mean().item()
            wm = p["walk_ms"] / p["calls"]
            print("%s plain d=%d ef=%d late_rows=%d: walk %.3f ms  %-44s dist_calc %.0f -> %.0f GB/s of row bytes (%.3f of 8 TB/s)"
                  % (name, ds.d, ef, late, wm, p["walk_kernel"][:44], dc, dc * ds.d * 4 * ds.nq / (wm * 1e-3) / 1e9, dc * ds.d * 4 * ds.nq / (wm * 1e-3) / 8e12), flush=True)
    ix.close()
lib.gbnns_debug_knob(b"late_rows", -1)
