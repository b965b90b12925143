#!/usr/bin/env python3
"""Diagnostic (needs a build with EXTRA_DEFS=-DGBNNS_LIFE_STAMPS in place of the library): a wavefront's life in walk_hot_kernel in three
parts -- before the first hop (table, query, entry), the hops, after the last hop (outputs, fused re-rank) -- in microseconds
(s_memrealtime), the launch's span and the longest life.  CONFIG=<bench.py configuration> (default sift), argv = beams."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gbnns_dim_red_amd as g
from gbnns_dim_red_amd import synth
import bench
cfg = bench.CONFIGS[os.environ.get("CONFIG", "sift")]
kw = dict(n=cfg["n"], nq=cfg["nq"], d=cfg["d"], d_low=cfg["d_low"], d_hidden=cfg["d_hidden"], seed=1234)
ds = synth.make_dataset(device="cuda:0", cache_dir=os.environ.get("GBNNS_CACHE", "/tmp/gbnns_cache"), **kw)
ix = ds.index()
lib = g.load_library()
lib.gbnns_debug_read_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
q = ds.queries
for ef in (int(a) for a in (sys.argv[1:] or [str(cfg["ef"])])):
    for _ in range(5):
        r = ix.search(q, ef, want=("hops",))
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 32)()
    lib.gbnns_debug_read_stamps(ix._h, buf)
    r = ix.search(q, ef, want=("hops",))
    torch.cuda.synchronize()
    lib.gbnns_debug_read_stamps(ix._h, buf)
    n = max(buf[3], 1)
    hops = r["hops"].double().mean().item()
    t0 = 2**62 - buf[5]
    print(f"ef={ef}: {buf[3]} wavefronts, {hops:.1f} hops each; life before the first hop {buf[0]/n/100:.2f} us, hops {buf[1]/n/100:.2f} us ({buf[1]/n/100/hops:.3f} per hop), "
          f"outputs + re-rank {buf[2]/n/100:.2f} us; longest life {buf[4]/100:.1f} us; the last wavefront started {(buf[6]-t0)/100:.1f} us and the last one ended "
          f"{(buf[7]-t0)/100:.1f} us after the first one started")
