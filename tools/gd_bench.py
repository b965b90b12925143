#!/usr/bin/env python3
"""GPU box: time the GD graph builder (support_func.h:521-575) on the bench workload's low-dim base set -- exact
kNN lists by gbnns_exact_knn, then per-node pruning on the device vs on the host, same graph.
usage: gd_bench.py [n] [K] [M] [host: 0/1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gbnns_dim_red_amd as g
from gbnns_dim_red_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 48
M = int(sys.argv[3]) if len(sys.argv) > 3 else 16
with_host = int(sys.argv[4]) if len(sys.argv) > 4 else 1
g.load_library()
ds = synth.make_dataset(device="cuda:0", cache_dir=os.environ.get("GBNNS_CACHE", "/tmp/gbnns_cache"), n=n, nq=10_000,
                        d=128, d_low=32, d_hidden=256, seed=1234)
x = ds.db_low.contiguous()
t0 = time.perf_counter()
parts = [g.exact_knn(x, x[s0:s0 + (1 << 18)], K, self_offset=s0) for s0 in range(0, n, 1 << 18)]
knn = torch.cat(parts).cpu().numpy().astype(np.uint32)
print("exact %d-NN lists: %.2f s" % (K, time.perf_counter() - t0), flush=True)
xh = x.cpu().numpy()
koff = np.arange(n + 1, dtype=np.uint64) * np.uint64(K)
t0 = time.perf_counter()
off, nbr, on_host = g.build_graph_gd_device(koff, knn.reshape(-1), xh, M)
td = time.perf_counter() - t0
print("GD(M=%d) with the pruning on the device: %.2f s (%d of %d nodes finished on the host), avg degree %.2f"
      % (M, td, on_host, n, len(nbr) / n), flush=True)
if with_host:
    t0 = time.perf_counter()
    off2, nbr2 = g.build_graph_gd(koff, knn.reshape(-1), xh, M)
    print("GD on the host (%d threads): %.2f s; identical: %s"
          % (os.cpu_count(), time.perf_counter() - t0, bool(np.array_equal(off, off2) and np.array_equal(nbr, nbr2))), flush=True)
