import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import gbnns_dim_red_amd as g
from gbnns_dim_red_amd import synth
import bench
cfg = bench.CONFIGS["gist"]
kw = dict(n=cfg["n"], nq=cfg["nq"], d=cfg["d"], d_low=cfg["d_low"], d_hidden=cfg["d_hidden"], seed=1234)
ds = synth.make_dataset(device="cuda:0", cache_dir="/tmp/gbnns_cache", **kw)
ix = ds.index()
q = ds.queries
ref = None
# A/B of the small-batch walks on the gist shape (1 000 queries): one / two / three wavefronts per query, twice (box drift)
VARIANTS = {
    "waves": (("one wavefront", dict(coop=0)), ("two wavefronts", dict(coop=1)), ("three wavefronts", dict(coop=2)),
              ("two wavefronts, packed", dict(coop=1, coop_pack=1)), ("three wavefronts, packed", dict(coop=2, coop_pack=1)),
              ("one wavefront", dict(coop=0)), ("two wavefronts", dict(coop=1)), ("three wavefronts", dict(coop=2))),
    # the visited set's share of LDS: at most `max_waves` wavefronts (workgroups) per CU are planned for, the table takes the rest
    "table": (("two wavefronts", dict(coop=1)), ("two wavefronts, 8 per CU", dict(coop=1, max_waves=8)), ("two wavefronts, 4 per CU", dict(coop=1, max_waves=4)),
              ("one wavefront", dict(coop=0)), ("one wavefront, 8 per CU", dict(coop=0, max_waves=8)), ("one wavefront, 4 per CU", dict(coop=0, max_waves=4)),
              ("two wavefronts", dict(coop=1)), ("two wavefronts, 4 per CU", dict(coop=1, max_waves=4))),
}
for name, knobs in VARIANTS[os.environ.get("AB", "waves")]:
    for k, v in {**dict(coop=-1, coop_pack=0, max_waves=0), **knobs}.items():
        ix.knob(k, v)
    for ef in (200, 400, 1000):
        for _ in range(5):
            r = ix.search(q, ef, want=())
        torch.cuda.synchronize()
        ix.profile_read(reset=True); ix.profile_enable(True)
        for _ in range(20):
            r = ix.search(q, ef, want=())
        torch.cuda.synchronize()
        p = ix.profile_read(reset=True); ix.profile_enable(False)
        if ref is None or ef not in ref:
            ref = ref or {}
            ref[ef] = r["ids"].clone()
        print("%-26s ef %4d: %s  %.4f ms  ids identical %s" % (name, ef, p["walk_kernel"].split(" (")[0][:40], p["walk_ms"] / p["calls"], bool((r["ids"] == ref[ef]).all())), flush=True)
