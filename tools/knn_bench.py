#!/usr/bin/env python3
"""GPU box: time gbnns_exact_knn on the low-dim base set of the bench workload (kNN graph of the set over
itself) and compare the lists with torch's formula-based top-k (tooling in synth.py).  usage: knn_bench.py [n] [k]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gbnns_dim_red_amd as g
from gbnns_dim_red_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 48
g.load_library()
ds = synth.make_dataset(device="cuda:0", cache_dir=os.environ.get("GBNNS_CACHE", "/tmp/gbnns_cache"), n=n, nq=10_000,
                        d=128, d_low=32, d_hidden=256, seed=1234)
x = ds.db_low.contiguous()
torch.cuda.synchronize()
slice_q = int(sys.argv[3]) if len(sys.argv) > 3 else n
lib = g.load_library()
pairs = n * (n - 1)
results = {}
# KNN_FILTER_ONLY=1: time the filter path alone (k = 1000: the exact scan of every row takes minutes)
paths = ((1, "matrix-core filter + exact distances of the kept rows"), (0, "exact scan of every row (rounds 1-3)"))
if os.environ.get("KNN_FILTER_ONLY"):
    paths = paths[:1]
for knob, what in paths:
    lib.gbnns_debug_knob(b"knn_filter", knob)
    g.exact_knn(x[:200000].contiguous(), x[:4096].contiguous(), k, self_offset=0)  # (code objects loaded, allocator warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = []
    for s0 in range(0, n, slice_q):
        out.append(g.exact_knn(x, x[s0:s0 + slice_q], k, self_offset=s0))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    results[knob] = torch.cat(out)
    print("gbnns_exact_knn, %s: n=%d d=32 k=%d  %.3f s  = %.2f T pairs/s (%.1f TFLOP/s-equivalent of ordered f32 sub/mul/add; the filter's "
          "matrix-core work: %.1f TFLOP/s bf16)" % (what, n, k, dt, pairs / dt / 1e12, pairs * 32 * 3 / dt / 1e12, pairs * 32 * 2 * 3 / dt / 1e12), flush=True)
lib.gbnns_debug_knob(b"knn_filter", 1)
if os.environ.get("KNN_FILTER_ONLY"):
    sys.exit(0)
print("filter path byte-identical to the exact scan:", bool((results[0] == results[1]).all().item()))
ids = results[1]
t0 = time.perf_counter()
ref = synth.knn_exact(x, k)
torch.cuda.synchronize()
print("torch formula top-k (synth.knn_exact): %.2f s" % (time.perf_counter() - t0))
same = (ids == ref.to(ids.dtype)).all(dim=1).float().mean().item()
inter = torch.tensor([len(set(a.tolist()) & set(b.tolist())) for a, b in zip(ids[:2000].cpu(), ref[:2000].cpu())]).float().mean().item()
print("rows identical to torch's: %.4f; mean overlap on 2000 rows: %.2f of %d" % (same, inter, k))
