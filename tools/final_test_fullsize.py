#!/usr/bin/env python3
"""GPU box: the drop-in `final_test` binary (gbnns_dim_red_amd/search/final_test, the reference's final_test.cpp flow) end to end at
full size -- the bench's SIFT1M-shaped synthetic workload written to disk in the reference's file formats (fvecs / ivecs / edge lists /
net-as-matrix / parameter table with the reference's own efs lists), then the binary run as a user would run it.  Prints its result
lines (the reference's format) with work_time turned into queries/s, and where the wall time of the whole run goes.
usage: final_test_fullsize.py [--config sift|deep1m|gist|glove1m] [--keep]"""
import argparse
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
import gbnns_dim_red_amd as g  # noqa: E402
from gbnns_dim_red_amd import synth  # noqa: E402

BIN = os.path.join(ROOT, "gbnns_dim_red_amd", "search", "final_test")
SWEEPS = {  # search/parameters_of_databases.txt:7-8, 29-30
    "sift": ("1,3,8,15,20,25,40,60,80,100,120,140,160,180", "1,4,7,11,15,20,30,40,60,80,100,120,130,140"),
    "deep1m": ("40,80,120,160,200", "40,80,120,160,200"),
    "gist": ("200,400,600,800,1000", "100,150,200,300,400"),      # :18-19 (no d_hidden row there: the width comes from second_part, :20)
    "glove1m": ("300,400,600,800,1000", "300,400,600,800,1000"),  # :40-41
}
NAMES = {"sift": "sift", "deep1m": "deep", "gist": "gist", "glove1m": "glove"}


def write_xvecs(path, a):
    a = np.ascontiguousarray(a)
    n, d = a.shape
    rec = np.empty((n, d + 1), np.uint32)
    rec[:, 0] = d
    rec[:, 1:] = a.view(np.uint32)
    rec.tofile(path)


def write_edges(path, off, nbr):
    off = off.astype(np.int64)
    n = len(off) - 1
    deg = np.diff(off)
    out = np.empty(n + len(nbr), np.uint32)
    head = off[:-1] + np.arange(n)          # position of row i's length word
    out[head] = deg.astype(np.uint32)
    body = np.ones(n + len(nbr), bool)
    body[head] = False
    out[body] = nbr.astype(np.uint32)
    out.tofile(path)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="sift", choices=sorted(SWEEPS))
    ap.add_argument("--dir", default="/tmp/gbnns_ft")
    ap.add_argument("--keep", action="store_true")
    args = ap.parse_args()
    cfg = bench.CONFIGS[args.config]
    name = NAMES[args.config]
    g.load_library()
    t0 = time.time()
    kw = dict(n=cfg["n"], nq=cfg["nq"], d=cfg["d"], d_low=cfg["d_low"], d_hidden=cfg["d_hidden"], seed=1234,
              cache_dir=os.environ.get("GBNNS_CACHE", "/tmp/gbnns_cache"))
    if cfg.get("unit_norm"):
        kw["unit_norm"] = True
    ds = synth.make_dataset(device="cuda:0", **kw)
    data, models, out = (os.path.join(args.dir, x) for x in ("data", "models", "out"))
    for p in (data, models, out):
        os.makedirs(p, exist_ok=True)
    write_xvecs(os.path.join(data, f"{name}_base.fvecs"), ds.base.cpu().numpy())
    write_xvecs(os.path.join(data, f"{name}_query.fvecs"), ds.queries.cpu().numpy())
    write_xvecs(os.path.join(data, f"{name}_groundtruth.ivecs"), ds.gt2.cpu().numpy().astype(np.uint32).view(np.float32))
    write_xvecs(os.path.join(data, f"{name}_base_angular_optimal.fvecs"), ds.db_low.cpu().numpy())
    write_edges(os.path.join(models, "hnsw_synthgraph.ivecs"), ds.graph_off, ds.graph_nbr)
    write_edges(os.path.join(models, "hnsw_synthgraph_angular_optimal.ivecs"), ds.graph_off, ds.graph_nbr)
    for i, layer in enumerate(ds.net, 1):
        write_xvecs(os.path.join(models, f"{name}_net_as_matrix_angular_optimal_{i}.fvecs"), layer.cpu().numpy())
    efs, efs_hnsw = SWEEPS[args.config]
    params = os.path.join(args.dir, "params.txt")
    with open(params, "w") as f:
        f.write("\n".join([f"{name} n {ds.n}", f"{name} n_q {ds.nq}", f"{name} n_tr 2", f"{name} d {ds.d}", f"{name} d_low {ds.d_low}",
                           (f"{name} second_part _{ds.d_low}_l_2_1m_5_40_w_{ds.d_hidden}_e_40" if name == "gist" else f"{name} d_hidden {ds.d_hidden}"),
                           f"{name} efs {efs}", f"{name} efs_hnsw {efs_hnsw}",
                           f"{name} hnsw_name synthgraph"]) + "\n")
    print("# %s-shaped synthetic (n = %d, %d queries, %d -> %d) written in the reference's file formats: %.1f s"
          % (args.config, ds.n, ds.nq, ds.d, ds.d_low, time.time() - t0), flush=True)
    del ds
    import torch
    torch.cuda.empty_cache()
    res = os.path.join(out, f"final_results_{name}.txt")
    if os.path.exists(res):
        os.remove(res)
    t0 = time.time()
    p = subprocess.run([BIN, name, data, models, out, params], capture_output=True, text=True, timeout=1500)
    wall = time.time() - t0
    if p.returncode != 0:
        print(p.stdout[-3000:], p.stderr[-3000:])
        return 1
    lines = open(res).read().splitlines()
    print("# final_test %s: %d result lines, wall time of the whole run %.1f s (files read, two indexes created, 2 sweeps x 5 repeats)"
          % (name, len(lines), wall))
    tot = 0.0
    for ln in lines:
        wt = float(ln.split(" work_time ")[1])
        tot += wt
        print("%-96s -> %7.3f M queries/s" % (ln, 1e-6 / wt))
    print("# timed regions of all lines: %.3f s of the %.1f s (5 repeats x %d queries each)" % (tot * 5 * cfg["nq"], wall, cfg["nq"]))
    other = [ln for ln in p.stdout.splitlines() if ln not in lines]
    print("# other output of the binary:")
    for ln in other[:40]:
        print("#   " + ln)
    if not args.keep:
        shutil.rmtree(args.dir, ignore_errors=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
