#!/usr/bin/env python3
"""Stress (GPU box): does page-locking small heap arrays through gbnns_host_pin / gbnns_host_unpin, then freeing them,
ever upset a later gbnns_index_create (round 4: two of six full GPU-suite runs aborted inside the index creation that
follows test_host_batches_in_flight)?   python tools/pin_stress.py [rounds] [aligned]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen  # noqa: E402
import gbnns_dim_red_amd as g  # noqa: E402
from gbnns_dim_red_amd import binding as B  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
aligned = len(sys.argv) > 2 and sys.argv[2] == "aligned"
lib = g.load_library()
rng = np.random.Generator(np.random.PCG64(1))
c = datagen.Case("pin", 7300, 6000, 1500, 48, 32, 64)
off, nbr = datagen.random_graph(rng, c.n, 4, 24)
db_low = np.ascontiguousarray(c.base[:, :32])


def buf(shape, dtype):
    if not aligned:
        return np.zeros(shape, dtype)
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    raw = np.zeros(n + 8192, np.uint8)
    o = (-raw.ctypes.data) % 4096
    return raw[o:o + n].view(dtype).reshape(shape)


for it in range(rounds):
    ix = g.Index(c.base, off, nbr, db_low=db_low, net=c.net)
    qn = buf((c.nq, c.d), np.float32)
    qn[:] = c.queries[:c.nq]
    ids_n, hops_n, dc_n = buf(c.nq, np.uint32), buf(c.nq, np.int32), buf(c.nq, np.int32)
    for arr in (qn, ids_n, hops_n, dc_n):
        assert lib.gbnns_host_pin(arr.ctypes.data, arr.nbytes) == 0
    a = B._SearchArgs(struct_size=C.sizeof(B._SearchArgs), mode=g.MODE_NET, ef=24, k=24, mem_kind=B.MEM_HOST, n_q=c.nq,
                      queries=qn.ctypes.data, out_ids=ids_n.ctypes.data, out_hops=hops_n.ctypes.data,
                      out_dist_calc=dc_n.ctypes.data, stream=None, flags=g.FLAG_DEFER_JOIN, defer_depth=3)
    B._check(lib.gbnns_search_ex(ix._h, C.byref(a)))
    ix.wait(0)
    ix.join()
    for arr in (qn, ids_n, hops_n, dc_n):
        assert lib.gbnns_host_unpin(arr.ctypes.data) == 0
    B._check(lib.gbnns_search_ex(ix._h, C.byref(a)))
    ix.close()
    del qn, ids_n, hops_n, dc_n
    junk = [np.zeros(rng.integers(100, 200000), np.uint8) for _ in range(20)]  # churn the heap
    if it % 10 == 0:
        print("round", it, flush=True)
print("done", rounds, "aligned" if aligned else "heap arrays", flush=True)
