#!/usr/bin/env python3
"""Kernel times on the other BASELINE.json configuration shapes (synthetic data, one GPU).
Not the contract bench (that is bench.py / SIFT1M): a survey of where the time goes per shape."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gbnns_dim_red_amd import synth

CONFIGS = {
    "sift": dict(n=1_000_000, nq=10_000, d=128, d_low=32, d_hidden=256, efs=[64, 128]),
    "gist": dict(n=1_000_000, nq=1_000, d=960, d_low=64, d_hidden=1024, efs=[200, 400]),
    "glove": dict(n=1_200_000, nq=10_000, d=200, d_low=32, d_hidden=256, efs=[64, 300]),
    "deep": dict(n=10_000_000, nq=125_000, d=96, d_low=32, d_hidden=128, efs=[40, 120]),
}
ap = argparse.ArgumentParser()
ap.add_argument("configs", nargs="*", default=["gist", "glove"])
ap.add_argument("--scale", type=float, default=1.0, help="scale n (and nq for deep) down for a quick look")
ap.add_argument("--negdot", action="store_true", help="walk / re-rank with the negative-dot metric (Angular::Dist)")
ap.add_argument("--aux", type=int, default=0, help="attach a random long-link auxiliary graph of this degree and "
                "run the reference's use_second_graph walk (llf, hops_bound 50) beside the plain one")
ap.add_argument("--inflight", type=int, default=0, help="also measure with this many batches in flight (one index handle and HIP stream each)")
ap.add_argument("--native-knn", action="store_true", help="build the dataset's kNN lists / ground truth with gbnns_exact_knn (needed for n = 10^7)")
a = ap.parse_args()
for name in a.configs:
    c = dict(CONFIGS[name])
    efs = c.pop("efs")
    c["n"] = int(c["n"] * a.scale)
    t0 = time.time()
    ds = synth.make_dataset(device="cuda:0", cache_dir=os.environ.get("GBNNS_CACHE", "/tmp/gbnns_cache"),
                            native_knn=a.native_knn, verbose=a.native_knn, **c)
    if a.negdot:
        from gbnns_dim_red_amd import binding
        ix = ds.index(metric=binding.METRIC_NEG_DOT)
    else:
        ix = ds.index()
    print(f"== {name}: n={ds.n} nq={ds.nq} {ds.d}->{ds.d_low} (h {ds.d_hidden}) built in {time.time()-t0:.1f}s", flush=True)
    variants = [dict()]
    if a.aux:
        import numpy as np
        rng = np.random.Generator(np.random.PCG64(5))
        anbr = rng.integers(0, ds.n, size=(ds.n, a.aux), dtype=np.int64).astype(np.uint32)
        ix.set_aux_graph(np.arange(ds.n + 1, dtype=np.uint64) * a.aux, anbr.reshape(-1))
        variants.append(dict(aux=True, llf=True, hops_bound=50))
    for ef, kw in [(e, v) for e in efs for v in variants]:
        for _ in range(3):
            r = ix.search(ds.queries, ef, want=("hops", "dist_calc"), **kw)
        torch.cuda.synchronize()
        ix.profile_read(reset=True); ix.profile_enable(True)
        t1 = time.perf_counter()
        for _ in range(5):
            r = ix.search(ds.queries, ef, want=("hops", "dist_calc"), **kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / 5
        p = ix.profile_read(reset=True); ix.profile_enable(False)
        rec = (r["ids"].long() == ds.gt).float().mean().item()
        k = {x: round(p[x + "_ms"] / p["calls"], 4) for x in ("project", "walk", "walk_general", "rerank")}
        rr_bytes = ds.nq * ef * 4.0 * ds.d
        print(json.dumps(dict(config=name, ef=ef, aux=bool(kw), recall=round(rec, 4), qps=round(ds.nq / dt), ms=round(dt * 1e3, 3),
                              kernels_ms=k, rerank_GBps=(round(rr_bytes / (k["rerank"] * 1e-3) / 1e9, 1) if k["rerank"] > 0.02 else None),
                              hops=round(r["hops"].float().mean().item(), 1),
                              dist_calc=round(r["dist_calc"].float().mean().item(), 1),
                              general=p["general_queries"])), flush=True)
    if a.inflight > 1:
        hs = [ix] + [ds.index() for _ in range(a.inflight - 1)]
        ss = [torch.cuda.Stream() for _ in hs]
        outs = [{} for _ in hs]
        for ef in efs:
            for i in range(4 * len(hs)):
                hs[i % len(hs)].search(ds.queries, ef, want=(), stream=ss[i % len(hs)], out=outs[i % len(hs)])
            torch.cuda.synchronize()
            reps = 8 * len(hs)
            t1 = time.perf_counter()
            for i in range(reps):
                hs[i % len(hs)].search(ds.queries, ef, want=(), stream=ss[i % len(hs)], out=outs[i % len(hs)])
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / reps
            print(json.dumps(dict(config=name, ef=ef, batches_in_flight=len(hs), qps=round(ds.nq / dt), ms_per_batch=round(dt * 1e3, 3))), flush=True)
        for h in hs[1:]:
            h.close()
    ix.close()
    del ds
    torch.cuda.empty_cache()
