"""GPU box: the profile record of a lone gist-shaped call per small-batch walk form (knob coop 0 / 1 / 2) -- which kernels ran, how long,
how many queries were handed over."""
import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import gbnns_dim_red_amd as g
from gbnns_dim_red_amd import synth
import bench
cfg = bench.CONFIGS[os.environ.get("CONFIG", "gist")]
kw = dict(n=cfg["n"], nq=cfg["nq"], d=cfg["d"], d_low=cfg["d_low"], d_hidden=cfg["d_hidden"], seed=1234)
ds = synth.make_dataset(device="cuda:0", cache_dir="/tmp/gbnns_cache", **kw)
ix = ds.index()
q = ds.queries
ef = int(sys.argv[1]) if len(sys.argv) > 1 else cfg["ef"]
for coop in (0, 1, 2, 2, 1):
    ix.knob("coop", coop)
    for _ in range(5):
        ix.search(q, ef, want=())
    torch.cuda.synchronize()
    ix.profile_read(reset=True); ix.profile_enable(True)
    for _ in range(10):
        r = ix.search(q, ef, want=("hops",))
    torch.cuda.synchronize()
    p = ix.profile_read(reset=True); ix.profile_enable(False)
    print("coop", coop, json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in p.items()}), flush=True)
