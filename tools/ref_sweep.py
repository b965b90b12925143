#!/usr/bin/env python3
"""GPU box: the reference's OWN sweeps at full size -- every beam of `efs` (two-stage search, final_test.cpp:87) and of
`efs_hnsw` (the plain walk in the original space, final_test.cpp:84) that search/parameters_of_databases.txt lists for a
dataset, on the bench's synthetic workload of that shape.  Per beam: the first-pass kernel that took it, ms per batch one
call at a time and three in flight, and the answers / hops / dist_calc of the first `--sample` queries against the compiled
reference (oracle/_ref, the checker) on the host's threads.
usage: ref_sweep.py --config sift|gist|deep1m|glove1m|glove|glove-dot [--sample 256] [--reps 5] [--only net|plain]"""
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
import oracle  # noqa: E402
from gbnns_dim_red_amd import synth  # noqa: E402

# search/parameters_of_databases.txt:7-8, 18-19, 29-30, 40-41 (deep and glove there are the 96 -> 48 and 300 -> 144 shapes)
SWEEPS = {
    "sift": ([1, 3, 8, 15, 20, 25, 40, 60, 80, 100, 120, 140, 160, 180], [1, 4, 7, 11, 15, 20, 30, 40, 60, 80, 100, 120, 130, 140]),
    "gist": ([200, 400, 600, 800, 1000], [100, 150, 200, 300, 400]),
    "deep1m": ([40, 80, 120, 160, 200], [40, 80, 120, 160, 200]),
    "glove1m": ([300, 400, 600, 800, 1000], [300, 400, 600, 800, 1000]),
    # BASELINE.json config 4 (GloVe-1.2M 200 -> 32; L2 on unit vectors and the negative-dot metric): the bench's beam + the reference's glove beams
    "glove": ([64, 300, 400, 600, 800, 1000], [64, 300, 400, 600]),
    "glove-dot": ([64, 300, 400, 600, 800, 1000], [64, 300, 400, 600]),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="sift", choices=sorted(SWEEPS))
    ap.add_argument("--sample", type=int, default=256)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", default=None, choices=["net", "plain"])
    ap.add_argument("--graph-M", type=int, default=None, help="GD pruning parameter of the workload's graph (default 16: rows of <= 32 slots; "
                    "20 / 30: rows of two passes, like the reference's M18 / M20 hnsw graphs)")
    ap.add_argument("--efs", default=None, help="comma-separated beams instead of the reference's lists (A/B runs)")
    args = ap.parse_args()
    cfg = bench.CONFIGS[args.config]
    g.load_library()
    kw = dict(n=cfg["n"], nq=cfg["nq"], d=cfg["d"], d_low=cfg["d_low"], d_hidden=cfg["d_hidden"], seed=1234,
              cache_dir=os.environ.get("GBNNS_CACHE", "/tmp/gbnns_cache"))
    if cfg.get("unit_norm"):
        kw["unit_norm"] = True
    if args.graph_M:
        kw["M"] = args.graph_M
    os.makedirs(kw["cache_dir"], exist_ok=True)
    ds = synth.make_dataset(device="cuda:0", **kw)
    metric_id = g.METRIC_NEG_DOT if cfg.get("negdot") else g.METRIC_L2
    ix = ds.index(metric=metric_id)
    q = ds.queries
    nq = ds.nq
    S = min(args.sample, nq)
    threads = bench.cpu_threads_available()
    ref = oracle.Ref()
    base_h = ds.base.cpu().numpy()
    dbl_h = ds.db_low.cpu().numpy()
    net_h = tuple(t.cpu().numpy() for t in ds.net)
    qh = q[:S].cpu().numpy()
    ref.prepare(base_h)
    print("# %s-shaped synthetic, n = %d, %d-query batches, %d -> %d (d_hidden %d); reference = oracle/_ref on %d host threads, "
          "first %d queries of the batch; graph: GD(M = %d), longest adjacency row %d"
          % (args.config, ds.n, nq, ds.d, ds.d_low, ds.d_hidden, threads, S, args.graph_M or 16, int(np.diff(ds.graph_off.astype(np.int64)).max())), flush=True)
    print("# mode   ef   kernel                                            ms/batch  in flight  M q/s   hops  dist_calc   "
          "ids hops dist_calc vs reference (of %d)" % S, flush=True)

    def run(mode_name, ef):
        plain = mode_name == "plain"
        kws = dict(mode=g.MODE_PLAIN, k=1) if plain else {}
        for _ in range(3):
            r = ix.search(q, ef, want=("hops", "dist_calc"), **kws)
        torch.cuda.synchronize()
        ix.profile_read(reset=True)
        ix.profile_enable(True)
        ix.search(q, ef, want=(), **kws)
        torch.cuda.synchronize()
        p = ix.profile_read(reset=True)
        ix.profile_enable(False)
        t0 = time.perf_counter()
        for _ in range(args.reps):
            ix.search(q, ef, want=(), **kws)
        torch.cuda.synchronize()
        serial = (time.perf_counter() - t0) / args.reps
        outs = [{} for _ in range(3)]
        n_fl = max(6, 3 * args.reps)
        for i in range(6):
            ix.search(q, ef, want=(), out=outs[i % 3], flags=g.FLAG_DEFER_JOIN, defer_depth=3, **kws)
        ix.join()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n_fl):
            ix.search(q, ef, want=(), out=outs[i % 3], flags=g.FLAG_DEFER_JOIN, defer_depth=3, **kws)
        ix.join()
        torch.cuda.synchronize()
        flight = (time.perf_counter() - t0) / n_fl
        if plain:
            e = ref.search_batch(oracle.MODE_PLAIN, qh, base_h, ds.graph_off, ds.graph_nbr, ef, k=1, threads=threads, metric=metric_id)
        else:
            e = ref.search_batch(oracle.MODE_NET, qh, base_h, ds.graph_off, ds.graph_nbr, ef, db_low=dbl_h, net=net_h, threads=threads, metric=metric_id)
        ids = r["ids"][:S].cpu().numpy().astype(np.int64)
        hops = r["hops"][:S].cpu().numpy().astype(np.int64)
        dc = r["dist_calc"][:S].cpu().numpy().astype(np.int64)
        same_ids = int((ids == e["ids"].astype(np.int64)).sum())
        same_hops = int((hops == e["hops"]).sum())
        # (performNetTest counts the re-ranked candidates too, search_function.h:362: + recheck_size = ef per query)
        same_dc = int((dc + (0 if plain else ef) == e["dist_calc"]).sum())
        kern = p.get("walk_kernel", "?")
        kern = kern.split(" (")[0]
        print("%-6s %5d  %-48s %8.3f  %8.3f  %6.2f  %5.1f  %8.1f    %d %d %d%s"
              % (mode_name, ef, kern[:48], serial * 1e3, flight * 1e3, nq / flight / 1e6, r["hops"].float().mean().item(),
                 r["dist_calc"].float().mean().item(), same_ids, same_hops, same_dc,
                 "" if same_ids == S and same_hops == S and same_dc == S else "   <-- DIFFERS"), flush=True)
        return same_ids == S and same_hops == S and same_dc == S

    efs, efs_plain = SWEEPS[args.config]
    if args.efs:
        efs = efs_plain = [int(x) for x in args.efs.split(",")]
    ok = True
    if args.only != "plain":
        for ef in efs:
            ok = run("net", ef) and ok
    if args.only != "net":
        for ef in efs_plain:
            ok = run("plain", ef) and ok
    print("# every sampled answer, hop count and dist_calc identical to the reference: %s" % ok, flush=True)
    ix.close()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
