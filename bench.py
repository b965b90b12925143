#!/usr/bin/env python3
"""bench.py -- queries/sec of the two-stage graph search on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (MLP projection -> low-dim beam walk -> original-space
re-rank) over one 10k-query batch, inputs and index resident in HBM, through the C ABI
(gbnns_search_ex with device buffers).  At N > 1 every rank holds a replica of the index and
searches its own 10k batch (weak scaling); the answer ids are all-gathered over RCCL each step.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--ef EF]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  See DESIGN.md "Measurement" for how every field is derived.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 measured)
REF_EFS = [1, 3, 8, 15, 20, 25, 40, 60, 80, 100, 120, 140, 160, 180]  # parameters_of_databases.txt:7


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--ef", type=int, default=64, help="beam width of the timed steps (config: 64)")
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--nq", type=int, default=10_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the diagnostic sections (separate stages, MFMA option, host buffers): profiling runs")
    ap.add_argument("--hash-capacity", type=int, default=0, help="0 = library default (tuning knob)")
    ap.add_argument("--sweep", action="store_true", help="also time every reference ef (stderr)")
    ap.add_argument("--cache-dir", default=os.environ.get("GBNNS_CACHE", "/tmp/gbnns_cache"))
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU path)"
    # rehearsal on a one-GPU box (not a measurement): GBNNS_BENCH_REHEARSAL=1 puts every rank on cuda:0 and
    # runs the collective over gloo, to exercise the N > 1 control flow without an 8-GPU node
    rehearsal = os.environ.get("GBNNS_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    import gbnns_dim_red_amd as g
    from gbnns_dim_red_amd import synth
    g.load_library()

    # ---- synthetic SIFT1M-shaped workload (rank 0 builds, the others load the cached copy) ----
    os.makedirs(args.cache_dir, exist_ok=True)
    kw = dict(n=args.n, nq=args.nq, d=128, d_low=32, d_hidden=256, seed=1234,
              cache_dir=args.cache_dir)
    if rank == 0:
        ds = synth.make_dataset(device=str(dev), verbose=False, **kw)
    if world > 1:
        dist.barrier()
    if rank != 0:
        ds = synth.make_dataset(device=str(dev), **kw)
    ix = ds.index(device_index=local)
    # every rank searches the same query pool in a different rotation (its own 10k batch)
    shift = (rank * args.nq) // world
    q = torch.roll(ds.queries, shifts=-shift, dims=0).contiguous()
    gt = torch.roll(ds.gt, shifts=-shift, dims=0)
    gt2 = torch.roll(ds.gt2[:, 1], shifts=-shift, dims=0)
    dup = ((ds.base[ds.gt2[:, 0]] - ds.base[ds.gt2[:, 1]]) ** 2).sum(1) == 0
    dup = torch.roll(dup, shifts=-shift, dims=0)

    def recall_of(ids):
        # search_function.h:391-400: hit on GT[0], or on GT[1] when the two are exact duplicates
        ids = ids.long()
        return ((ids == gt) | (dup & (ids == gt2))).float().mean().item()

    # ---- recall sweep (untimed) ------------------------------------------------------------
    sweep = {}
    for ef in sorted(set([16, 32, 64, 128, args.ef])):
        r = ix.search(q, ef, want=())
        torch.cuda.synchronize()
        sweep[ef] = recall_of(r["ids"])
    ef = args.ef
    if sweep[ef] < 0.95:
        cands = [e for e in REF_EFS + [256, 512] if e > ef]
        for e in cands:
            r = ix.search(q, e, want=())
            torch.cuda.synchronize()
            sweep[e] = recall_of(r["ids"])
            if sweep[e] >= 0.95:
                ef = e
                break
    recall = sweep[ef]

    # ---- timed region -----------------------------------------------------------------------
    want = ("hops", "dist_calc", "edges")
    # Two sets of output buffers used alternately: the all-gather of step i (RCCL's own stream) runs
    # beside the kernels of step i+1; step i+2 reuses step i's buffers and therefore waits for that
    # gather first (a stream-level wait, the host never blocks).
    outs = [{}, {}]
    gathered = [None, None]
    pending = [None, None]
    nstep = 0

    def step():
        nonlocal nstep
        b = nstep & 1
        nstep += 1
        if pending[b] is not None:
            pending[b].wait()
            pending[b] = None
        r = ix.search(q, ef, want=want, out=outs[b], hash_capacity=args.hash_capacity)
        if world > 1:
            # the path's only exchange step: all-gather of the int32 answer ids over RCCL/xGMI
            if gathered[b] is None:
                gathered[b] = torch.empty(world * args.nq, dtype=r["ids"].dtype, device=dev)
            pending[b] = dist.all_gather_into_tensor(gathered[b], r["ids"], async_op=True)
        return r

    def drain():
        for b in (0, 1):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

    # The library sizes its visited sets from the walks it has seen and drops the retry launch once a few batches
    # of a configuration were quiet: let that settle before the W warm-up steps, whatever W is.
    for _ in range(8):
        ix.search(q, ef, want=(), hash_capacity=args.hash_capacity)
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    ix.profile_read(reset=True)
    ix.profile_enable(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    drain()  # every step's gather is inside the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    prof = ix.profile_read(reset=True)
    ix.profile_enable(False)

    ms_per_step = elapsed * 1e3 / args.steps
    qps = world * args.nq * args.steps / elapsed

    # ---- algorithmic bytes of the dominant kernel (the beam walk), SURVEY.md section 8d ------
    dc = res["dist_calc"].double()
    hops = res["hops"].double()
    if rank == 0:
        qs = torch.tensor([0.5, 0.9, 0.99, 0.999, 1.0], dtype=torch.float64, device=dev)
        log("dist_calc quantiles 50/90/99/99.9/max:", [int(v) for v in torch.quantile(dc, qs).tolist()],
            "hops max", int(hops.max().item()))
    edges = res["edges"].double()
    d_low, d = ds.d_low, ds.d
    walk_bytes = (dc * 4 * d_low + edges * 4 + hops * 8 + 4 * d_low + 4 * ef).sum().item()
    rerank_bytes = args.nq * (ef * 4.0 * d + 4 * d + 4 + 4 * ef)
    walk_ms = prof["walk_ms"] / max(prof["calls"], 1)
    rerank_ms = prof["rerank_ms"] / max(prof["calls"], 1)
    project_ms = prof["project_ms"] / max(prof["calls"], 1)
    max_degree = int(np.diff(np.asarray(ds.graph_off).astype(np.int64)).max())
    # ef <= 64, 128-B rows, adjacency rows of <= 32 slots: the hand-laid-out instance, which also re-ranks
    # each query at the end of its walk (no re-rank launch) -- its algorithmic bytes are SURVEY 8d's full B(q)
    hot_shape = ds.d_low == 32 and max_degree <= 32
    hot = ef <= 64 and hot_shape
    fused = ef <= 512 and ds.d % 8 == 0   # every register-list first pass re-ranks at the end of the walk
    kernel_bytes = walk_bytes + (rerank_bytes if fused else 0.0)
    achieved = kernel_bytes / (walk_ms * 1e-3) / 1e9 if walk_ms > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("walk_fast_hbm_bytes_per_launch")
        except Exception:
            traffic = None

    result = {
        "metric": "queries/sec @ recall@1>=0.95, SIFT1M 128->32",
        "value": round(qps, 1),
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "SIFT1M-shaped synthetic 128->32 (d_hidden 256), n=%d, %d-query batch per GPU, "
                        "ef=%d, two-stage (project+walk+rerank), index resident in HBM" % (args.n, args.nq, ef),
            "ef": ef,
            "recall_at_1": round(recall, 4),
            "recall_sweep": {str(k): round(v, 4) for k, v in sorted(sweep.items())},
            "mean_hops": round(hops.mean().item(), 1),
            "mean_dist_calc": round(dc.mean().item(), 1),
            "graph": "kNN(%d)->GD(M=%d,reverse), avg degree %.1f" % (
                ds.recipe["knn_k"], ds.recipe["M"], len(ds.graph_nbr) / ds.n),
            "parallelism": "query-sharded replicas x%d" % world,
            "recipe": ds.recipe,
        },
        "roofline": {
            "bound": "hbm",
            # ef <= 64, 128-B rows, adjacency rows of <= 32 slots: the hand-laid-out instance
            "kernel": (("walk_hot_kernel" if hot else
                        ("walk_hot%d_kernel" % ((ef + 63) // 64) if ef <= 256 else "walk_hotN_kernel<%d>" % ((ef + 63) // 64))
                        if hot_shape and ef <= 512 else "walk_reg_kernel" if ef <= 512 else "walk_fast_kernel")
                       + (" (walk + fused re-rank)" if fused else "")),
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic,
            "algorithmic_bytes_per_launch": round(kernel_bytes),
            "algorithmic_bytes_walk_part": round(walk_bytes),
            "algorithmic_bytes_rerank_part": round(rerank_bytes if fused else 0),
            "kernel_ms": round(walk_ms, 4),
        },
        "kernels_ms": {"project": round(project_ms, 4), "walk": round(walk_ms, 4),
                       "walk_general": round(prof["walk_general_ms"] / max(prof["calls"], 1), 4),
                       "rerank": None if fused else round(rerank_ms, 4),  # fused: inside the walk kernel
                       "rerank_GBps": (round(rerank_bytes / (rerank_ms * 1e-3) / 1e9, 1)
                                       if rerank_ms > 0 and not fused else None),
                       "general_queries": prof["general_queries"]},
    }

    # ---- the same step with the re-rank in its own launch (diagnostic flag): per-stage kernel times -------
    if world == 1 and fused and not args.no_extras:
        for _ in range(3):
            ix.search(q, ef, want=(), flags=g.FLAG_NO_FUSED_RERANK)
        torch.cuda.synchronize()
        ix.profile_read(reset=True)
        ix.profile_enable(True)
        t1 = time.perf_counter()
        for _ in range(10):
            ru = ix.search(q, ef, want=(), flags=g.FLAG_NO_FUSED_RERANK)
        torch.cuda.synchronize()
        dtu = (time.perf_counter() - t1) / 10
        pu = ix.profile_read(reset=True)
        ix.profile_enable(False)
        wu, ru_ms = pu["walk_ms"] / max(pu["calls"], 1), pu["rerank_ms"] / max(pu["calls"], 1)
        result["separate_stages"] = {
            "ms_per_step": round(dtu * 1e3, 4), "walk_ms": round(wu, 4), "rerank_ms": round(ru_ms, 4),
            "walk_GBps": round(walk_bytes / (wu * 1e-3) / 1e9, 1) if wu > 0 else None,
            "rerank_GBps": round(rerank_bytes / (ru_ms * 1e-3) / 1e9, 1) if ru_ms > 0 else None,
            "answers_identical": bool((ru["ids"] == res["ids"]).all().item()),
        }

    # ---- opt-in matrix-core projection (not bit-exact): how fast, and how many answers change --------
    if world == 1 and not args.no_extras:
        for _ in range(3):
            rm = ix.search(q, ef, want=(), flags=g.FLAG_MFMA_PROJECT)
        torch.cuda.synchronize()
        ix.profile_read(reset=True)
        ix.profile_enable(True)
        for _ in range(10):
            rm = ix.search(q, ef, want=(), flags=g.FLAG_MFMA_PROJECT)
        torch.cuda.synchronize()
        pm = ix.profile_read(reset=True)
        ix.profile_enable(False)
        result["mfma_project_option"] = {
            "project_ms": round(pm["project_ms"] / max(pm["calls"], 1), 4),
            "answers_changed": int((rm["ids"] != res["ids"]).sum().item()),
            "recall_at_1": round(recall_of(rm["ids"]), 4),
            "note": "f32 MFMA fma-chain rounding; off by default, default path is bit-exact",
        }

    # ---- two batches in flight: two handles over the same resident index, one HIP stream each, used
    # alternately, so the tail of batch i (a 10 k batch is < 2 "rounds" of resident wavefronts) runs
    # beside the projection and the first round of batch i+1.  A serving-throughput figure; never `value`
    # (which keeps one batch at a time on one stream).
    if world == 1 and not args.no_extras:
        ix2 = ds.index(device_index=local)
        handles = (ix, ix2)
        streams = (torch.cuda.Stream(dev), torch.cuda.Stream(dev))
        outs2 = ({}, {})
        torch.cuda.synchronize()
        for i in range(12):  # both handles reach their settled sizing
            handles[i & 1].search(q, ef, want=(), stream=streams[i & 1], out=outs2[i & 1])
        torch.cuda.synchronize()
        nrep = 40
        t1 = time.perf_counter()
        for i in range(nrep):
            handles[i & 1].search(q, ef, want=(), stream=streams[i & 1], out=outs2[i & 1])
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t1
        result["two_batches_in_flight"] = {
            "queries_per_s": round(nrep * args.nq / dt2, 1),
            "ms_per_batch": round(dt2 * 1e3 / nrep, 4),
            "answers_identical": bool(((outs2[0]["ids"] == res["ids"]) & (outs2[1]["ids"] == res["ids"])).all().item()),
            "note": "two index handles (shared resident tensors) on two HIP streams, batches alternate",
        }
        ix2.close()

    # ---- PCIe-inclusive rate (host buffers in, ids out: what the C++ drop-in times); never `value`
    if world == 1 and not args.no_extras:
        qh = q.cpu().numpy()
        ix.search(qh, ef, want=())
        t1 = time.perf_counter()
        for _ in range(5):
            ix.search(qh, ef, want=())
        result["host_buffers_qps"] = round(5 * args.nq / (time.perf_counter() - t1), 1)

    # ---- CPU baseline (rank 0, N = 1 only): the compiled reference if present, else the port ----
    if world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(ds, q, ef, res["ids"])

    if args.sweep and rank == 0:
        for e in REF_EFS:
            ix.search(q, e, want=())
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(5):
                r = ix.search(q, e, want=())
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / 5
            log(f"sweep ef={e}: recall={recall_of(r['ids']):.4f} qps={args.nq / dt:.0f}")

    if rank == 0:
        print(json.dumps(result), flush=True)
    ix.close()
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(ds, q, ef, gpu_ids):
    """Times the reference's per-query body on host cores over the SAME batch and checks that the
    GPU answers are identical.  oracle/ is used here only as the reported baseline / checker."""
    import oracle
    base = ds.base.cpu().numpy()
    dbl = ds.db_low.cpu().numpy()
    net = tuple(t.cpu().numpy() for t in ds.net)
    qh = q.cpu().numpy()
    off, nbr = ds.graph_off, ds.graph_nbr
    if oracle.have_ref():
        impl, kind = oracle.Ref(), "reference"
        impl.prepare(base)
        impl.search_batch(oracle.MODE_NET, qh[:8], base, off, nbr, ef, db_low=dbl, net=net)  # graph conv
    else:
        impl, kind = oracle.Oracle(), "port"
    cores = impl.max_threads()
    # 1 thread (what final_test.cpp ships, :71) on a bounded sample, then all cores on the batch
    ns = min(len(qh), 2000)
    t0 = time.perf_counter()
    impl.search_batch(oracle.MODE_NET, qh[:ns], base, off, nbr, ef, db_low=dbl, net=net, threads=1)
    t1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    r = impl.search_batch(oracle.MODE_NET, qh, base, off, nbr, ef, db_low=dbl, net=net, threads=cores)
    tn = time.perf_counter() - t0
    same = int((r["ids"].astype(np.int64) == gpu_ids.cpu().numpy().astype(np.int64)).sum())
    return {
        "value": round(len(qh) / tn, 1),
        "unit": "queries/s",
        "cores": cores,
        "kind": kind,
        "sample": "the full %d-query batch at ef=%d, OpenMP over queries (search_function.h:152) on %d "
                  "threads; 1-thread figure on the first %d queries" % (len(qh), ef, cores, ns),
        "value_1thread": round(ns / t1, 1),
        "gpu_ids_identical": same == len(qh),
        "gpu_id_mismatches": len(qh) - same,
    }


if __name__ == "__main__":
    main()
