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
t0 = time.perf_counter()
out = []
for s0 in range(0, n, slice_q):
    out.append(g.exact_knn(x, x[s0:s0 + slice_q], k, self_offset=s0))
    print("  slice", s0, "done at %.2fs" % (time.perf_counter() - t0), flush=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ids = torch.cat(out)
pairs = n * (n - 1)
print("gbnns_exact_knn: n=%d d=32 k=%d  %.2f s  = %.2f T distance evaluations/s (%.1f TFLOP/s of ordered f32 sub/mul/add)"
      % (n, k, dt, pairs / dt / 1e12, pairs * 32 * 3 / dt / 1e12))
t0 = time.perf_counter()
ref = synth.knn_exact(x, k)
torch.cuda.synchronize()
print("torch formula top-k (synth.knn_exact): %.2f s" % (time.perf_counter() - t0))
same = (ids == ref.to(ids.dtype)).all(dim=1).float().mean().item()
inter = torch.tensor([len(set(a.tolist()) & set(b.tolist())) for a, b in zip(ids[:2000].cpu(), ref[:2000].cpu())]).float().mean().item()
print("rows identical to torch's: %.4f; mean overlap on 2000 rows: %.2f of %d" % (same, inter, k))
