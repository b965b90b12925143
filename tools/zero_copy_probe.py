#!/usr/bin/env python3
"""Probe (not a test): the SURVEY-8(d) rate (queries start in host memory, ids end there) three ways --
(a) pageable host buffers through GBNNS_MEM_HOST (what the C++ drop-in does), (b) page-locked host buffers through
GBNNS_MEM_HOST, (c) page-locked host buffers handed over as DEVICE pointers: the projection reads the queries across
PCIe itself and the walk kernel stores the ids straight into host memory (zero copy, wire time under the kernels).
python tools/zero_copy_probe.py [--ef EF]"""
import argparse, ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import gbnns_dim_red_amd as g
from gbnns_dim_red_amd import synth, binding as B

ap = argparse.ArgumentParser()
ap.add_argument("--ef", type=int, default=64)
ap.add_argument("--config", default="sift")
args = ap.parse_args()
g.load_library()
os.makedirs("/tmp/gbnns_cache", exist_ok=True)
shape = dict(sift=dict(d=128, d_low=32, d_hidden=256, nq=10_000), gist=dict(d=960, d_low=64, d_hidden=1024, nq=1_000))[args.config]
ds = synth.make_dataset(n=1_000_000, seed=1234, cache_dir="/tmp/gbnns_cache", device="cuda:0", **shape)
ix = ds.index()
q = ds.queries
nq = q.shape[0]
ref = ix.search(q, args.ef, want=())["ids"].cpu()
qh = q.cpu().numpy()
qp = q.cpu().pin_memory()
out_p = torch.empty(nq, dtype=torch.int32).pin_memory()
s = torch.cuda.current_stream()


def call(mem_kind, qptr, optr):
    a = B._SearchArgs(struct_size=C.sizeof(B._SearchArgs), mode=0, ef=args.ef, k=1, mem_kind=mem_kind, n_q=nq,
                      queries=qptr, out_ids=optr, stream=s.cuda_stream, flags=0)
    B._check(ix._lib.gbnns_search_ex(ix._h, C.byref(a)))
    if mem_kind == B.MEM_DEVICE:
        s.synchronize()


def rate(fn, reps=20):
    for _ in range(3):
        fn()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    return reps * nq / (time.perf_counter() - t)


out_h = np.empty(nq, np.uint32)
print("pageable, MEM_HOST        %.2f M q/s" % (rate(lambda: call(B.MEM_HOST, qh.ctypes.data, out_h.ctypes.data)) / 1e6), flush=True)
print("  ids identical:", bool((torch.from_numpy(out_h.astype(np.int32)) == ref).all()))
out_p.zero_()
print("page-locked, MEM_HOST     %.2f M q/s" % (rate(lambda: call(B.MEM_HOST, qp.data_ptr(), out_p.data_ptr())) / 1e6), flush=True)
print("  ids identical:", bool((out_p == ref).all()))
out_p.zero_()
print("page-locked, zero copy    %.2f M q/s" % (rate(lambda: call(B.MEM_DEVICE, qp.data_ptr(), out_p.data_ptr())) / 1e6), flush=True)
print("  ids identical:", bool((out_p == ref).all()))
qd = q.clone()
out_d = torch.empty(nq, dtype=torch.int32, device=q.device)
print("device buffers, sync each %.2f M q/s" % (rate(lambda: call(B.MEM_DEVICE, qd.data_ptr(), out_d.data_ptr())) / 1e6), flush=True)
out_p.zero_()
print("device in, zero-copy out  %.2f M q/s" % (rate(lambda: call(B.MEM_DEVICE, qd.data_ptr(), out_p.data_ptr())) / 1e6), flush=True)
print("  ids identical:", bool((out_p == ref).all()))

# ---- a stream of host batches: page-locked queries, hipMemcpyAsync on a copy stream that runs one batch ahead, the
# call deferred (GBNNS_FLAG_DEFER_JOIN), ids stored by the kernel straight into page-locked memory
NSETS = 8
sets = [q] + [synth.more_queries(ds, nq, b) for b in range(1, NSETS)]
qps = [x.cpu().pin_memory() for x in sets]
qds = [torch.empty_like(x) for x in sets]
outs = [torch.empty(nq, dtype=torch.int32).pin_memory() for _ in sets]
refs = [ix.search(x, args.ef, want=())["ids"].cpu() for x in sets]
cs = torch.cuda.Stream()


def stream_of_batches(reps, depth, ahead):
    """ahead = 0: the copy sits on the caller's stream in front of its call; 1: a copy stream runs one batch ahead"""
    copied = [torch.cuda.Event() for _ in range(NSETS)]
    issued = None

    def copy(i):
        k = i % NSETS
        if ahead:
            if issued is not None:
                cs.wait_event(issued)  # set k was last used by batch i - NSETS, joined on s before `issued`
            with torch.cuda.stream(cs):
                qds[k].copy_(qps[k], non_blocking=True)
                copied[k].record(cs)
        else:
            qds[k].copy_(qps[k], non_blocking=True)

    if ahead:
        copy(0)
    for i in range(reps):
        k = i % NSETS
        if ahead:
            if i + 1 < reps:
                copy(i + 1)
            s.wait_event(copied[k])
        else:
            copy(i)
        a = B._SearchArgs(struct_size=C.sizeof(B._SearchArgs), mode=0, ef=args.ef, k=1, mem_kind=B.MEM_DEVICE, n_q=nq,
                          queries=qds[k].data_ptr(), out_ids=outs[k].data_ptr(),
                          stream=s.cuda_stream, flags=B.FLAG_DEFER_JOIN, defer_depth=depth)
        B._check(ix._lib.gbnns_search_ex(ix._h, C.byref(a)))
        issued = torch.cuda.Event()
        issued.record(s)
    ix.join()
    s.synchronize()
    cs.synchronize()


for depth, ahead in ((3, 0), (4, 0), (3, 1), (4, 1)):
    stream_of_batches(8, depth, ahead)
    for o in outs:
        o.zero_()
    t = time.perf_counter()
    stream_of_batches(64, depth, ahead)
    dt = time.perf_counter() - t
    print("host batches in flight: depth %d, copy %s  %.2f M q/s" % (
        depth, "one batch ahead on its own stream" if ahead else "on the caller's stream", 64 * nq / dt / 1e6), flush=True)
    print("  ids identical:", all(bool((o == r).all()) for o, r in zip(outs, refs)))
