#!/usr/bin/env python3
"""Tuning tool (not a test, not the bench): step time of a bench workload for the call layouts of gbnns_search_ex --
device buffers plain / deferred join (alternating lanes), pageable host buffers; four distinct batches rotating.
python tools/split_bench.py [--config sift] [--ef 64]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import gbnns_dim_red_amd as g  # noqa: E402
from gbnns_dim_red_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="sift")
    ap.add_argument("--ef", type=int, default=None)
    ap.add_argument("--reps", type=int, default=60)
    ap.add_argument("--modes", default="device,defer,host")
    ap.add_argument("--depths", default="2,3,4")
    args = ap.parse_args()
    cfg = bench.CONFIGS[args.config]
    ef = args.ef or cfg["ef"]
    g.load_library()
    kw = dict(n=cfg["n"], nq=cfg["nq"], d=cfg["d"], d_low=cfg["d_low"], d_hidden=cfg["d_hidden"], seed=1234,
              cache_dir="/tmp/gbnns_cache")
    if cfg.get("unit_norm"):
        kw["unit_norm"] = True
    os.makedirs("/tmp/gbnns_cache", exist_ok=True)
    ds = synth.make_dataset(device="cuda:0", **kw)
    ix = ds.index()
    qs = [torch.roll(ds.queries, shifts=-i * (ds.nq // 4), dims=0).contiguous() for i in range(4)]
    qh = [q.cpu().numpy() for q in qs]
    ref = [ix.search(q, ef, want=())["ids"].cpu().numpy().astype(np.int64) for q in qs]
    for _ in range(8):
        ix.search(qs[0], ef, want=())
    torch.cuda.synchronize()

    def run(batches, flags=0, depth=0):
        outs = [{} for _ in range(max(depth, 2))]
        nb = len(outs)
        for i in range(12):
            ix.search(batches[i & 3], ef, want=(), out=outs[i % nb], flags=flags, defer_depth=depth)
        ix.join()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.reps):
            r = ix.search(batches[i & 3], ef, want=(), out=outs[i % nb], flags=flags, defer_depth=depth)
        ix.join()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.reps
        ids = r["ids"] if isinstance(r["ids"], np.ndarray) else r["ids"].cpu().numpy()
        ok = bool((ids.astype(np.int64) == ref[(args.reps - 1) & 3]).all())
        return "%.4f ms (%.2f M/s)%s" % (dt * 1e3, ds.nq / dt / 1e6, "" if ok else " WRONG")

    if "device" in args.modes:
        print("device plain        ", run(qs), flush=True)
    if "defer" in args.modes:
        for dpt in args.depths.split(","):
            print("device defer-join, %s in flight" % dpt, run(qs, g.FLAG_DEFER_JOIN, int(dpt)), flush=True)
    if "host" in args.modes:
        print("host pageable       ", run(qh), flush=True)


if __name__ == "__main__":
    main()
