#!/usr/bin/env python3
"""GPU box: the throughput option's projection (GBNNS_FLAG_MFMA_PROJECTION) beside the exact one on the bench workload -- project_ms of
each from the library's hipEvent pairs, q_low error, answers that differ.  CONFIG=<bench.py configuration> (default sift).
Run under rocprofv3 by tools/mfma_option_profile.sh for the kernel durations and the matrix-pipe counters."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gbnns_dim_red_amd as g
from gbnns_dim_red_amd import synth
import bench
cfg = bench.CONFIGS[os.environ.get("CONFIG", "sift")]
kw = dict(n=cfg["n"], nq=cfg["nq"], d=cfg["d"], d_low=cfg["d_low"], d_hidden=cfg["d_hidden"], seed=1234)
if cfg.get("unit_norm"):
    kw["unit_norm"] = True
ds = synth.make_dataset(device="cuda:0", cache_dir=os.environ.get("GBNNS_CACHE", "/tmp/gbnns_cache"), **kw)
ix = ds.index()
q, ef = ds.queries[:int(os.environ.get("NQ", len(ds.queries)))].contiguous(), cfg["ef"]
out = {"config": os.environ.get("CONFIG", "sift"), "nq": len(q), "ef": ef}
ex = ix.search(q, ef, want=("q_low",), flags=g.FLAG_SERIAL)
for name, fl in (("exact", 0), ("option", g.FLAG_MFMA_PROJECTION)):
    for _ in range(5):
        r = ix.search(q, ef, want=("q_low",), flags=fl | g.FLAG_SERIAL)
    torch.cuda.synchronize()
    ix.profile_read(reset=True); ix.profile_enable(True)
    for _ in range(int(os.environ.get("REPS", "20"))):
        r = ix.search(q, ef, want=("q_low",), flags=fl | g.FLAG_SERIAL)
    torch.cuda.synchronize()
    p = ix.profile_read(reset=True); ix.profile_enable(False)
    out[name] = {"project_kernel": p["project_kernel"], "project_ms": round(p["project_ms"] / p["calls"], 4),
                 "max_abs_q_low_err": float((r["q_low"] - ex["q_low"]).abs().max().item()),
                 "id_mismatches_vs_exact": int((r["ids"] != ex["ids"]).sum().item())}
print(json.dumps(out))
