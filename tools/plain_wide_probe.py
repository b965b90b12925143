import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, bench
import gbnns_dim_red_amd as g
from gbnns_dim_red_amd import synth
g.load_library()
for name, efs in (("gist", (100, 200, 400)), ("glove1m", (300, 600))):
    cfg = bench.CONFIGS[name]
    kw = dict(n=cfg["n"], nq=cfg["nq"], d=cfg["d"], d_low=cfg["d_low"], d_hidden=cfg["d_hidden"], seed=1234, cache_dir="/tmp/gbnns_cache")
    if cfg.get("unit_norm"): kw["unit_norm"] = True
    ds = synth.make_dataset(device="cuda:0", **kw)
    ix = ds.index()
    for ef in efs:
        for _ in range(2): r = ix.search(ds.queries, ef, mode=g.MODE_PLAIN, k=1, want=("hops", "dist_calc"))
        torch.cuda.synchronize(); ix.profile_read(reset=True); ix.profile_enable(True)
        for _ in range(3): r = ix.search(ds.queries, ef, mode=g.MODE_PLAIN, k=1, want=("hops", "dist_calc"))
        torch.cuda.synchronize(); p = ix.profile_read(reset=True); ix.profile_enable(False)
        dc = r["dist_calc"].float().mean().item(); wm = p["walk_ms"] / p["calls"]
        print("%s plain d=%d nq=%d ef=%d: walk %.3f ms %-40s dist_calc %.0f -> %.0f GB/s (%.3f)" % (name, ds.d, ds.nq, ef, wm, p["walk_kernel"][:40], dc, dc*ds.d*4*ds.nq/(wm*1e-3)/1e9, dc*ds.d*4*ds.nq/(wm*1e-3)/8e12), flush=True)
    ix.close()
