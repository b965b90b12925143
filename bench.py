#!/usr/bin/env python3
"""bench.py -- queries/sec of the two-stage graph search on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (MLP projection -> low-dim beam walk -> original-space
re-rank) over one query batch, inputs and index resident in HBM, through the C ABI
(gbnns_search_ex with device buffers).  The index is replicated per GPU; ranks search disjoint
query blocks and all-gather the answer ids over RCCL each step.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config NAME] [--ef EF]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

--config picks one of BASELINE.json's shapes (default `sift` = configs[1], the one `metric` is
quoted on; the others are the survey's configurations 3-5 as bench lines of their own):

    sift       SIFT1M-shaped 128->32 (h 256), 10 000-query batch per GPU, ef 64        weak scaling
    gist       GIST1M-shaped 960->64 (h 1024), 1 000-query batch per GPU, ef 200       weak scaling
    glove      GloVe-1.2M-shaped 200->32 (h 256), L2 on normalised data, ef 64         weak scaling
    glove-dot  the same index walked / re-ranked with the negative-dot metric          weak scaling
    deep       DEEP10M-shaped 96->32 (h 128), ONE 1 000 000-query batch block-sharded
               over the ranks (125 000 per GPU at N = 8), ef 40                        strong scaling

Prints ONE JSON line on rank 0.  See DESIGN.md "Measurement" for how every field is derived.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process starts N workers itself
(python -m torch.distributed.run, rendezvous on 127.0.0.1) BEFORE anything touches the GPU, lets rank 0's
line through and exits with their status.  Under an external launcher (WORLD_SIZE set) it is a worker.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 measured)
REF_EFS = [1, 3, 8, 15, 20, 25, 40, 60, 80, 100, 120, 140, 160, 180]  # parameters_of_databases.txt:7

CONFIGS = {
    # name: dataset shape, timed ef, further efs reported in `ef_sweep`, metric, batch policy
    "sift": dict(n=1_000_000, nq=10_000, d=128, d_low=32, d_hidden=256, ef=64, efs=[36, 40, 128, 140, 160, 180],
                 label="SIFT1M 128->32", shape="SIFT1M-shaped"),
    "gist": dict(n=1_000_000, nq=1_000, d=960, d_low=64, d_hidden=1024, ef=200, efs=[400],
                 label="GIST1M 960->64", shape="GIST1M-shaped"),
    "glove": dict(n=1_200_000, nq=10_000, d=200, d_low=32, d_hidden=256, ef=64, efs=[300], unit_norm=True,
                  label="GloVe-1.2M 200->32 (L2 on normalised data)", shape="GloVe-1.2M-shaped"),
    "glove-dot": dict(n=1_200_000, nq=10_000, d=200, d_low=32, d_hidden=256, ef=64, efs=[300], negdot=True, unit_norm=True,
                      label="GloVe-1.2M 200->32 (negative-dot metric)", shape="GloVe-1.2M-shaped"),
    # the reference's own DEEP row (parameters_of_databases.txt:23-33): 10^6 base vectors, 96 -> 48 (192-byte walked rows)
    "deep1m": dict(n=1_000_000, nq=10_000, d=96, d_low=48, d_hidden=128, ef=40, efs=[80, 120, 160, 200],
                   label="DEEP1M 96->48 (the reference's parameter file)", shape="DEEP1M-shaped"),
    # the reference's own GloVe row (parameters_of_databases.txt:35-45): 10^6 vectors, 300 -> 144 (576-byte walked rows)
    # (d_hidden 512 instead of the file's 256: the synthetic net realises its projection exactly through ReLU and needs
    # d_hidden >= 2 d_low; the walk and the re-rank are what this row is about)
    "glove1m": dict(n=1_000_000, nq=10_000, d=300, d_low=144, d_hidden=512, ef=300, efs=[400, 600, 800, 1000], unit_norm=True,
                    label="GloVe1M 300->144 (the reference's parameter file)", shape="GloVe1M-shaped"),
    # a harder data recipe for the tuning constants (tools/dist_probe.py; DESIGN_APPENDIX_R5 6): less clustered, more intrinsic
    # dimensions -- the longest walk of a batch is 1.47 x the median instead of 1.28 x, recall@1 0.86 / 0.97 at ef 64 / 160; timed at ITS
    # OWN recall gate (the sweep below walks the beam up from 64 and bisects; with the appendix's noisier variant, sigma 0.1, no beam up to
    # 1 000 reaches 0.95: the 32-dimensional projection loses the neighbour)
    "sift-hard": dict(n=1_000_000, nq=10_000, d=128, d_low=32, d_hidden=256, ef=64, efs=[160],
                      recipe=dict(intrinsic=24, n_clusters=100, cluster_scale=1.0, sigma=0.03),
                      label="SIFT1M 128->32 (harder synthetic recipe)", shape="SIFT1M-shaped, harder recipe"),
    "deep": dict(n=10_000_000, nq=1_000_000, d=96, d_low=32, d_hidden=128, ef=40, efs=[60, 120], strong=True,
                 native_knn=True, label="DEEP10M 96->32, 1M-query batch", shape="DEEP10M-shaped"),
}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def launcher_command(gpus, argv, environ):
    """The worker launch for `--gpus N` when this process is not already a worker (WORLD_SIZE unset), else None.
    The workers get this process's arguments through GBNNS_BENCH_ARGV (torch.distributed.run's own parser would
    take a script option such as --n for an abbreviation of one of its own)."""
    if gpus <= 1 or "WORLD_SIZE" in environ:
        return None
    with socket.socket() as sk:  # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)]


def launch_probe():
    """`--launch-probe`: what a worker does to prove the launch path without a GPU -- gloo rendezvous, one all-reduce,
    rank 0 prints the number of ranks it saw."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("GBNNS_PROBE_FAIL_RANK") == os.environ.get("RANK", "0"):
        sys.exit(3)  # test hook: a worker that dies must make the launcher's exit status non-zero
    seen = 1
    if world > 1:
        dist.init_process_group("gloo")
        t = torch.ones(1)
        dist.all_reduce(t)
        seen = int(t.item())
        assert dist.get_world_size() == world
    if int(os.environ.get("RANK", "0")) == 0:
        print(json.dumps({"launch_probe": True, "ranks_seen": seen, "world_size": world}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    t_start = time.time()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--launch-probe", action="store_true", help="test hook: exercise the N-worker launch only (no GPU)")
    ap.add_argument("--graph-M", type=int, default=None,
                    help="GD pruning parameter of the synthetic graph (default 16: max degree <= 32, 32-slot adjacency rows; "
                         "18 gives the 48-slot rows of the reference's own gist / deep graphs)")
    ap.add_argument("--batches", type=int, default=4, help="distinct query batches rotated through the timed loop")
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 200; 20 for gist / glove*, 3 for deep)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--settle-ms", type=float, default=50.0,
                    help="untimed stretch of the step loop in front of the warm-up steps (device clocks; 0 = none)")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="sift")
    ap.add_argument("--ef", type=int, default=None, help="beam width of the timed steps (default: the configuration's)")
    ap.add_argument("--n", type=int, default=None, help="override the base-set size (quick looks)")
    ap.add_argument("--nq", type=int, default=None, help="override the batch size")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the diagnostic sections (separate stages, host buffers, ef sweep): profiling runs")
    ap.add_argument("--hash-capacity", type=int, default=0, help="0 = library default (tuning knob)")
    ap.add_argument("--bitmap-pass", action="store_true", help="force the HBM-bitmap first pass (tuning knob: GBNNS_FLAG_BITMAP_PASS)")
    ap.add_argument("--sweep", action="store_true", help="also time every reference ef (stderr)")
    ap.add_argument("--depth", type=int, default=None,
                    help="batches in flight inside the library (GBNNS_FLAG_DEFER_JOIN, defer_depth 2..4); default 3, the measured "
                         "optimum on every configuration (tools/split_bench.py)")
    ap.add_argument("--serial", action="store_true",
                    help="timed steps one batch at a time on one stream (no GBNNS_FLAG_DEFER_JOIN pipelining)")
    ap.add_argument("--cache-dir", default=os.environ.get("GBNNS_CACHE", "/tmp/gbnns_cache"))
    ap.add_argument("--cpu-sample", type=int, default=0,
                    help="CPU baseline on the first N queries only, widest thread count only (what the default run asks of the "
                         "other configurations' child runs: a quick identical-ids check with its rate)")
    ap.add_argument("--capi-multi-child", default=None,
                    help="internal: run the C-ABI multi-replica section (gbnns_multi_*) over --gpus devices in this fresh process; "
                         "the value is the .npy file with the ranks' answer ids of batch 0 to compare with")
    ap.add_argument("--full-line", action="store_true",
                    help="print the complete record on stdout instead of the compact line (tools, the other_configs child runs)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling for the per-rank-batch configurations (sift, gist, glove*): ONE batch of --strong-nq queries "
                         "(default 80 000: eight times the configuration's batch) block-sharded over the ranks, total work fixed as N grows; "
                         "the default stays weak scaling (a 10 000-query batch per rank)")
    ap.add_argument("--strong-nq", type=int, default=None, help="--strong: queries of the one sharded batch (default 8 x the configuration's)")
    ap.add_argument("--no-capi-multi", action="store_true", help="N > 1: skip the C-ABI multi-replica section")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="default sift run at N = 1: do not append the gist / glove / glove-dot (/ deep) lines under other_configs")
    ap.add_argument("--budget-s", type=float, default=float(os.environ.get("GBNNS_BENCH_BUDGET_S", "400")),
                    help="wall-clock budget of the whole default run; the other configurations are started only while it lasts")
    argv = sys.argv[1:]
    if not argv and os.environ.get("GBNNS_BENCH_ARGV") and "WORLD_SIZE" in os.environ:
        argv = json.loads(os.environ["GBNNS_BENCH_ARGV"])  # a worker started by the branch below
    args = ap.parse_args(argv)
    cmd = None if args.capi_multi_child else launcher_command(args.gpus, argv, os.environ)
    if cmd is not None:
        # nothing above has touched the GPU (importing torch does not): the workers are fresh processes
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env["GBNNS_BENCH_ARGV"] = json.dumps(argv)
        sys.exit(subprocess.run(cmd, env=env).returncode)
    if args.launch_probe:
        launch_probe()
        return
    cfg = dict(CONFIGS[args.config])
    cfg["name"] = args.config
    if args.steps is None:
        args.steps = {"sift": 200, "deep": 3}.get(args.config, 20)
    if args.warmup is None:
        args.warmup = {"sift": 10, "deep": 1}.get(args.config, 3)
    if args.n:
        cfg["n"] = args.n
    if args.nq:
        cfg["nq"] = args.nq
    if args.strong and not cfg.get("strong"):
        cfg["strong"] = True
        cfg["strong_by_flag"] = True
        cfg["nq"] = args.strong_nq or 8 * cfg["nq"]

    if args.capi_multi_child:
        capi_multi_child(args, cfg)
        return
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
    from gbnns_dim_red_amd import sharding, synth
    g.load_library()

    # ---- synthetic workload (rank 0 builds, the others load the cached copy) ------------------
    os.makedirs(args.cache_dir, exist_ok=True)
    kw = dict(n=cfg["n"], nq=cfg["nq"], d=cfg["d"], d_low=cfg["d_low"], d_hidden=cfg["d_hidden"], seed=1234,
              cache_dir=args.cache_dir, native_knn=bool(cfg.get("native_knn")) and cfg["n"] > 2_000_000)
    if cfg.get("unit_norm"):
        kw["unit_norm"] = True  # GloVe vectors are normalised before anything else (train_naive_triplet.py:233-237)
    if args.graph_M:
        kw["M"] = args.graph_M
    if cfg.get("recipe"):
        kw.update(cfg["recipe"])
    if cfg.get("strong") and cfg["nq"] > 200_000:
        kw["gt_queries"] = 20_000  # exact ground truth for the first 20 000 queries of the 1M batch (recall sample)
    if rank == 0:
        ds = synth.make_dataset(device=str(dev), verbose=args.config == "deep", **kw)
    if world > 1:
        dist.barrier()
    if rank != 0:
        ds = synth.make_dataset(device=str(dev), **kw)
    metric_id = g.METRIC_NEG_DOT if cfg.get("negdot") else g.METRIC_L2
    ix = ds.index(device_index=local, metric=metric_id)
    strong = bool(cfg.get("strong"))
    cold_ef = args.ef or cfg["ef"]
    if strong:
        # ONE batch, contiguous blocks per rank; entry points would shard with it
        lo, hi = sharding.shard_bounds(ds.nq, world, rank)
        q = ds.queries[lo:hi].contiguous()
        n_gt = min(len(ds.gt), ds.nq)
        g_lo, g_hi = min(lo, n_gt), min(hi, n_gt)   # the part of this rank's block that has ground truth
        gt, gt2 = ds.gt[g_lo:g_hi], ds.gt2[g_lo:g_hi, 1]
        dup = ((ds.base[ds.gt2[g_lo:g_hi, 0]] - ds.base[ds.gt2[g_lo:g_hi, 1]]) ** 2).sum(1) == 0
        n_scored = g_hi - g_lo
    else:
        # every rank searches the same query pool in a different rotation (its own batch)
        shift = (rank * ds.nq) // world
        q = torch.roll(ds.queries, shifts=-shift, dims=0).contiguous()
        gt = torch.roll(ds.gt, shifts=-shift, dims=0)
        gt2 = torch.roll(ds.gt2[:, 1], shifts=-shift, dims=0)
        dup = ((ds.base[ds.gt2[:, 0]] - ds.base[ds.gt2[:, 1]]) ** 2).sum(1) == 0
        dup = torch.roll(dup, shifts=-shift, dims=0)
        n_scored = len(gt)
    nq_rank = int(q.shape[0])
    nq_total = ds.nq if strong else world * ds.nq

    def recall_of(ids):
        # search_function.h:391-400: hit on GT[0], or on GT[1] when the two are exact duplicates
        if n_scored == 0:
            return float("nan")
        ids = ids.long()[:n_scored]
        return ((ids == gt) | (dup & (ids == gt2))).float().mean().item()

    # ---- cold start (untimed, N = 1): the first calls on the fresh handle, one batch at a time ------------------
    # The library sizes its visited sets from the batches it has seen and leaves the retry launch out once a few were
    # calm; the first call also pays the runtime's lazy loading of the code objects and the lanes'
    # workspace allocations.  What a caller sees before any of that has settled:
    cold = None
    if world == 1 and nq_rank <= 20_000:
        times = []
        for j in range(12):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            ix.search(q, cold_ef, want=())
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t1) * 1e3)
        steady = sorted(times[-4:])[1]
        cold = {"first_call_ms": round(times[0], 3), "second_call_ms": round(times[1], 3),
                "calls_until_steady": next(j for j, t in enumerate(times) if t <= 1.1 * steady) + 1,
                "steady_call_ms": round(steady, 4), "call_ms": [round(t, 3) for t in times],
                "note": "fresh handle, ef = %d, one synchronous device-buffer call at a time; `steady` = within 10 %% of the "
                        "median of calls 9-12; the first call includes the runtime's code-object load and the workspace "
                        "allocations, calls 2 .. n the visited-set sizing from the previous batches' statistics" % cold_ef}

    # ---- recall sweep (untimed) ------------------------------------------------------------
    ef = args.ef or cfg["ef"]
    sweep = {}
    for e in sorted(set(([16, 32, 64, 128] if args.config == "sift" else []) + [ef] + cfg["efs"])):
        r = ix.search(q, e, want=())
        torch.cuda.synchronize()
        sweep[e] = recall_of(r["ids"])
    gate_failed = False
    if sweep[ef] < 0.95:
        for e in [e for e in REF_EFS + [200, 256, 300, 400, 512, 600, 800, 1000] if e > ef]:
            if e not in sweep:
                r = ix.search(q, e, want=())
                torch.cuda.synchronize()
                sweep[e] = recall_of(r["ids"])
            if sweep[e] >= 0.95:
                # the reference's ef grid is coarse (40 -> 60 here): bisect down to a grid of 4 for the smallest beam
                # that still meets the metric's recall gate; the reference-grid point stays in `recall_sweep` / `ef_sweep`
                lo, hi = max(x for x in sweep if x < e and sweep[x] < 0.95), e
                while hi - lo > 4:
                    mid = (lo + hi) // 2
                    if mid not in sweep:
                        r = ix.search(q, mid, want=())
                        torch.cuda.synchronize()
                        sweep[mid] = recall_of(r["ids"])
                    if sweep[mid] >= 0.95:
                        hi = mid
                    else:
                        lo = mid
                ef = hi
                break
        else:
            gate_failed = True  # no ef up to 1000 reaches the metric's recall threshold on this data
    recall = sweep[ef]

    # ---- timed region -----------------------------------------------------------------------
    want = ("hops", "dist_calc", "edges")
    # Distinct query batches rotated through the loop (the recall-scored batch first): replaying one batch would touch
    # the same rows every step, the friendliest case for L2 / Infinity Cache.
    nb = max(1, min(args.batches, args.steps))
    batches = [q] + [synth.more_queries(ds, nq_rank, batch=rank * 64 + j) for j in range(1, nb)]
    # Steps are pipelined `depth` deep inside the library (GBNNS_FLAG_DEFER_JOIN, include/gbnns.h): batch i runs on one
    # of the handle's internal streams and the caller's stream is made to wait for it by call i+depth-1, after that
    # call's batch has been released -- the projection of batch i+1 runs in the half-empty tail of batch i's walk
    # kernel, and batches too small to fill the machine run side by side.
    # `depth` sets of output buffers in rotation; the all-gather of step i (RCCL's own stream) is issued once the
    # stream has joined batch i and runs beside the kernels of the following steps; the step that reuses step i's
    # buffers waits for that gather first (a stream-level wait, the host never blocks).
    tune_flags = g.FLAG_BITMAP_PASS if args.bitmap_pass else 0
    pipelined = args.hash_capacity == 0 and not args.bitmap_pass and not args.serial
    depth = max(2, min(4, args.depth or 3)) if pipelined else 1
    step_flags = tune_flags | (g.FLAG_DEFER_JOIN if pipelined else 0)
    nbuf = max(depth, 2)
    outs = [{} for _ in range(nbuf)]
    gathered = [None] * nbuf
    pending = [None] * nbuf
    nstep = 0
    pad = sharding.shard_pad(nq_total, world) if strong else nq_rank  # equal-sized pieces for the all-gather

    def gather(b):
        # the path's only exchange step: all-gather of the int32 answer ids over RCCL/xGMI
        if gathered[b] is None:
            gathered[b] = torch.empty(world * pad, dtype=outs[b]["ids"].dtype, device=dev)
        piece = outs[b]["ids"]
        if pad != nq_rank:
            if "ids_pad" not in outs[b]:
                outs[b]["ids_pad"] = torch.full((pad,), -1, dtype=piece.dtype, device=dev)
            outs[b]["ids_pad"][:nq_rank] = piece
            piece = outs[b]["ids_pad"]
        pending[b] = dist.all_gather_into_tensor(gathered[b], piece, async_op=True)

    def step():
        nonlocal nstep
        i = nstep
        b = i % nbuf
        nstep += 1
        if pending[b] is not None:
            pending[b].wait()
            pending[b] = None
        r = ix.search(batches[i % nb], ef, want=want, out=outs[b], hash_capacity=args.hash_capacity, flags=step_flags,
                      defer_depth=depth if pipelined else 0)
        if world > 1:
            if not pipelined:
                gather(b)
            elif i >= depth - 1:
                gather((i - (depth - 1)) % nbuf)  # that call made the stream wait for batch i-depth+1: its answers are complete in stream order
        return r

    def drain():
        ix.join()  # the stream waits for the batches still unjoined
        if world > 1 and pipelined:
            for j in range(max(0, nstep - (depth - 1)), nstep):
                gather(j % nbuf)
        for b in range(nbuf):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

    # The library sizes its visited sets from the walks it has seen and drops the retry launch once a few batches
    # of a configuration were quiet: let that settle before the W warm-up steps, whatever W is.
    for j in range(8 if nq_rank <= 20_000 else 3):
        ix.search(batches[j % nb], ef, want=(), hash_capacity=args.hash_capacity, flags=step_flags, defer_depth=depth)
        ix.join()
        torch.cuda.synchronize()
    # ... and the device itself: the first ~10 ms of batches in flight after the set-up above run 10 % slower than the
    # steady state (measured: 20 timed steps after 5 warm-up steps 28.4 M queries/s, after 100 warm-up steps 31.1 M, 200
    # timed steps after 5: 31.4 M -- clocks / power state, not the pipeline's fill and drain).  An untimed stretch of the
    # same loop, about 50 ms long (at most 500 steps; the same count on every rank), comes before the W warm-up steps.
    ts = time.perf_counter()
    for _ in range(4):
        step()
    drain()
    torch.cuda.synchronize()
    nstep = 0
    est = max((time.perf_counter() - ts) / 4, 1e-5)
    n_settle = int(min(500, max(0, round(args.settle_ms * 1e-3 / est))))
    if args.settle_ms > 0 and nq_rank <= 20_000:
        n_settle = max(n_settle, 120)  # (the four steps of the estimate are themselves slow ones)
    if world > 1:
        tn = torch.tensor([n_settle], dtype=torch.int64, device=dev)
        dist.all_reduce(tn, op=dist.ReduceOp.MAX)
        n_settle = int(tn.item())
    for _ in range(n_settle):
        step()
    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    nstep = 0
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()  # every step's join and gather is inside the timed region
    torch.cuda.synchronize()
    own_elapsed = time.perf_counter() - t0   # this rank's K steps, before it waits for the others
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    per_rank = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
        # per rank (diagnostic, after the timed region): its own ms per step, how long it then waited at the closing barrier, and the
        # exchange step alone -- one all-gather of the id vectors, timed by itself
        tg0 = time.perf_counter()
        for b in range(4):
            gather(b % nbuf)
            pending[b % nbuf].wait()
            pending[b % nbuf] = None
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - tg0) * 1e3 / 4
        mine = torch.tensor([own_elapsed * 1e3 / args.steps, (elapsed - own_elapsed) * 1e3, gather_ms, float(nq_rank)], dtype=torch.float64, device=dev)
        allr = torch.empty(world * 4, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(allr, mine)
        allr = allr.view(world, 4).cpu().tolist()
        per_rank = {"ms_per_step": [round(v[0], 4) for v in allr], "barrier_wait_ms": [round(v[1], 3) for v in allr],
                    "gather_alone_ms": [round(v[2], 4) for v in allr], "queries": [int(v[3]) for v in allr]}

    ms_per_step = elapsed * 1e3 / args.steps
    qps = nq_total * args.steps / elapsed
    # every rank's own piece of the last gathered id vectors equals what it searched itself (the buffers of the last
    # `nbuf` steps are all still in place): a wrong join / gather order in the pipelined loop would show here
    gather_ok = None
    if world > 1:
        good = True
        for b in range(nbuf):
            if gathered[b] is not None and "ids" in outs[b]:
                good = good and bool((gathered[b][rank * pad: rank * pad + nq_rank] == outs[b]["ids"][:nq_rank]).all().item())
        tg = torch.tensor([1.0 if good else 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(tg, op=dist.ReduceOp.MIN)
        gather_ok = bool(tg.item() == 1.0)

    # ---- the same K steps serialised (GBNNS_FLAG_SERIAL semantics: profiling implies it): one batch at a time on one
    # stream, kernels back to back, with the library's hipEvent pairs around every stage on the launch stream.  This is
    # where roofline.kernel_ms comes from -- overlapped kernels cannot be timed individually.
    ix.profile_read(reset=True)
    ix.profile_enable(True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for i in range(args.steps):
        ix.search(batches[i % nb], ef, want=want, out=outs[i % nbuf], hash_capacity=args.hash_capacity, flags=tune_flags)
    torch.cuda.synchronize()
    elapsed_serial = time.perf_counter() - t1
    prof = ix.profile_read(reset=True)
    ix.profile_enable(False)
    # walk statistics of every rotated batch (algorithmic bytes are averaged over them)
    stats = {k: [] for k in want}
    for j in range(nb):
        r = ix.search(batches[j], ef, want=want, flags=tune_flags | g.FLAG_SERIAL)
        for k in want:
            stats[k].append(r[k].clone())
    res = {k: torch.cat(v) for k, v in stats.items()}
    res["ids"] = ix.search(q, ef, want=(), flags=tune_flags | g.FLAG_SERIAL)["ids"].clone()
    torch.cuda.synchronize()

    # ---- algorithmic bytes of the dominant kernel (the beam walk), SURVEY.md section 8d ------
    max_degree = int(np.diff(np.asarray(ds.graph_off).astype(np.int64)).max())
    rl = roofline_of(ds, res, prof, ef, nq_rank, max_degree, cfg, rank, launches=nb)
    hops = res["hops"].double()
    dc = res["dist_calc"].double()

    result = {
        "metric": "queries/sec @ recall@1>=0.95, %s" % cfg["label"].split(" (")[0].replace(", 1M-query batch", ""),
        "value": round(qps, 1),
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "settle_steps_before_warmup": n_settle + 4,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "%s synthetic %d->%d (d_hidden %d), n=%d, %s, ef=%d, two-stage (project+walk+rerank), "
                        "index resident in HBM%s" % (
                            cfg["shape"], ds.d, ds.d_low, ds.d_hidden, ds.n,
                            ("one %d-query batch block-sharded over %d GPU(s)" % (nq_total, world)) if strong
                            else "%d-query batch per GPU" % nq_rank,
                            ef, ", negative-dot metric in walk and re-rank" if cfg.get("negdot") else ""),
            "name": args.config,
            "ef": ef,
            "recall_at_1": round(recall, 4),
            "recall_scored_queries": int(n_scored),
            "recall_sweep": {str(k): round(v, 4) for k, v in sorted(sweep.items())},
            "mean_hops": round(hops.mean().item(), 1),
            "mean_dist_calc": round(dc.mean().item(), 1),
            "graph": "kNN(%d)->GD(M=%d,reverse), avg degree %.1f, max %d" % (
                ds.recipe["knn_k"], ds.recipe["M"], len(ds.graph_nbr) / ds.n, max_degree),
            "parallelism": "query-sharded replicas x%d" % world,
            "recipe": ds.recipe,
        },
        "config_nq": nq_rank,
        "ranks_seen": dist.get_world_size() if world > 1 else 1,
        "gather_self_check": gather_ok,
        "per_rank": per_rank,
        "strong_batch": nq_total if strong else None,
        "roofline": rl["roofline"],
        "kernels_ms": rl["kernels_ms"],
        "batches_rotated": nb,
        "pipelined": bool(pipelined),
        "batches_in_flight": depth,
        # the same K steps one batch at a time on one stream, kernels back to back (what round 1 / 2 reported as `value`)
        "serial": {"ms_per_step": round(elapsed_serial * 1e3 / args.steps, 4),
                   "queries_per_s": round(nq_rank * args.steps / elapsed_serial, 1),
                   "note": "per rank, GBNNS_FLAG_SERIAL semantics, hipEvent profiling on: roofline.kernel_ms is measured here"},
        # what `value` leaves out, per SURVEY 8d's definition (one gbnns_search_batch incl. H2D / D2H): see
        # survey_8d_value below (filled at N = 1 unless --no-extras)
        "value_definition": "queries of all ranks / wall time of K steps, inputs and outputs resident in HBM (device "
                            "buffers), %s; the PCIe-inclusive rate of the host-buffer call is survey_8d_value" % (
                                "steps pipelined %d deep inside the library (GBNNS_FLAG_DEFER_JOIN: later batches are released "
                                "before the stream waits for batch i), %d distinct batches rotating" % (depth, nb) if pipelined
                                else "one batch at a time"),
    }
    if cold:
        result["cold"] = cold
    if gate_failed:
        result["recall_gate_failed"] = True  # `value` is NOT a figure at recall >= 0.95

    # ---- N > 1: the same batch through the C ABI's own multi-device path (gbnns_multi_create + gbnns_multi_search_device:
    # one host process, one host thread and stream per device, ONE RCCL all-gather of the ids per batch) -- what a C++
    # host links against.  A fresh child process of rank 0 drives all N devices while every rank is parked at a barrier
    # (their GPUs idle); never `value`; an RCCL or launch failure is a field of the line, not a failed bench.
    if world > 1 and not args.no_capi_multi:
        ref = ix.search(q, ef, want=(), flags=tune_flags | g.FLAG_SERIAL)["ids"].clone()
        torch.cuda.synchronize()
        if not strong or nq_total % world == 0:   # (equal blocks: one all-gather of the ranks' answers to batch 0)
            allv = torch.empty(world * nq_rank, dtype=ref.dtype, device=dev)
            dist.all_gather_into_tensor(allv, ref)
            pieces = [allv]
        else:   # uneven blocks: padded pieces, the padding cut out again by the shard bounds
            mine = torch.full((pad,), -1, dtype=ref.dtype, device=dev)
            mine[:nq_rank] = ref
            allv = torch.empty(world * pad, dtype=ref.dtype, device=dev)
            dist.all_gather_into_tensor(allv, mine)
            pieces = []
            for r_ in range(world):
                lo_, hi_ = sharding.shard_bounds(nq_total, world, r_)
                pieces.append(allv[r_ * pad: r_ * pad + (hi_ - lo_)])
        dist.barrier()
        if rank == 0:
            result["capi_multi"] = capi_multi_parent(args, argv, world, pieces, rehearsal)
        dist.barrier()

    extras = world == 1 and not args.no_extras
    small = nq_rank <= 20_000

    # ---- other efs of this configuration: kernel time and roofline fraction each -------------------
    if extras:
        result["ef_sweep"] = ef_sweep(ix, ds, q, cfg, ef, sweep, recall_of, nq_rank, max_degree, rank, batches=batches,
                                      depth=depth if pipelined else 1)
        ok = [e for e in result["ef_sweep"] if e["recall_at_1"] >= 0.95]
        if ok:
            best = max(ok, key=lambda e: e["queries_per_s_in_flight"] or e["queries_per_s"])
            # the metric's own operating point: the smallest beam of the sweep past the recall gate (same derivation as the
            # headline's figures: in flight = three batches inside the library, kernel_ms / frac from serialised steps)
            result["best_ef_at_recall_gate"] = {k: best[k] for k in ("ef", "recall_at_1", "queries_per_s", "queries_per_s_in_flight",
                                                                     "ms_per_step", "kernel", "kernel_ms",
                                                                     "algorithmic_bytes_per_launch", "achieved_GBps", "frac")}

    # ---- the same step with the re-rank in its own launch (diagnostic flag): per-stage kernel times -------
    if extras and small and rl["fused"]:
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
            "walk_GBps": round(rl["walk_bytes"] / (wu * 1e-3) / 1e9, 1) if wu > 0 else None,
            "rerank_GBps": round(rl["rerank_bytes"] / (ru_ms * 1e-3) / 1e9, 1) if ru_ms > 0 else None,
            "answers_identical": bool((ru["ids"] == res["ids"]).all().item()),
        }

    # ---- the throughput option (GBNNS_FLAG_MFMA_PROJECTION: the projection as v_mfma_f32_32x32x2_f32 GEMMs, NOT bit-exact):
    # its projection time, the rate with batches in flight, how far q_low moves and how many of the batch's answers differ
    # from the exact path's (which the cpu_baseline section compares with the reference); never `value`
    if extras and small and cfg["name"] != "plain":
        result["throughput_option"] = throughput_option(g, ix, q, ef, res, batches, depth if pipelined else 1, nq_rank)
    if extras:
        result["graph_prep"] = graph_prep_figures(g, ds) if small and cfg["name"] == "sift" else graph_prep_figures()

    # ---- the multi-device path's exchange leg on the one GPU there is (gbnns_multi_*, csrc/multi.cpp): one replica, librccl
    # loaded, a one-rank communicator, ncclAllGather of the single block per batch -- answers compared, the RCCL version
    # recorded; never `value`
    if extras and small:
        result["rccl_single_rank"] = rccl_single_rank(g, ds, q, ef, metric_id, res["ids"], nq_rank)

    # ---- PCIe-inclusive rate (host buffers in, ids out: what the C++ drop-in times, and SURVEY 8d's
    # "one gbnns_search_batch incl. H2D of queries and D2H of ids"); never `value`
    if extras and small:
        qh = q.cpu().numpy()
        ix.search(qh, ef, want=())
        t1 = time.perf_counter()
        for _ in range(5):
            ix.search(qh, ef, want=())
        result["host_buffers_qps"] = round(5 * nq_rank / (time.perf_counter() - t1), 1)
        # SURVEY 8d defines the metric's rate over one batch call INCLUDING H2D of the queries and D2H of the ids
        result["survey_8d_value"] = result["host_buffers_qps"]
        # ... and as a stream of host batches (gbnns.h "Host batches in flight"): page-locked buffers, HOST mem_kind +
        # GBNNS_FLAG_DEFER_JOIN, three batches in flight, six sets of result buffers and gbnns_index_wait(5) after every
        # call (the set about to be reused is the one that call waited for: the host never blocks on a batch that is
        # still among the three running) -- queries start in host memory, ids end there, the wire time of a batch
        # passes under its neighbours' kernels
        hdepth = int(os.environ.get("GBNNS_BENCH_HDEPTH", "3"))  # measured: 2 / 3 / 4 in flight -> 23.3 / 28.4 / 17.6 M queries/s
        hsets = 2 * hdepth
        hq = [b.cpu().pin_memory() for b in batches]
        houts = [{} for _ in range(hsets)]

        def host_stream(count):
            for i in range(count):
                ix.search(hq[i % len(hq)], ef, want=(), out=houts[i % hsets], flags=g.FLAG_DEFER_JOIN, defer_depth=hdepth)
                ix.wait(hsets - 1)
            ix.wait(0)

        # (the runtime spends 5-7 ms inside the first hipMemcpyAsync out of each page-locked buffer, and now and then
        # inside a later one -- GBNNS_SLOW_US shows them as lone slow calls; a window of 96 batches read 24 or 28 M
        # queries/s depending on whether it caught one: the window is long enough for one to cost ~4 %)
        host_stream(24)
        reps = 480
        t1 = time.perf_counter()
        host_stream(reps)
        result["host_batches_in_flight_qps"] = round(reps * nq_rank / (time.perf_counter() - t1), 1)
        result["survey_8d_in_flight"] = result["host_batches_in_flight_qps"]
        ix.join()
        torch.cuda.synchronize()
        refs = [ix.search(b, ef, want=())["ids"].cpu() for b in batches]   # the last `hsets` batches are still in their buffers
        result["host_batches_in_flight_ids_identical"] = all(
            bool((houts[j % hsets]["ids"] == refs[j % len(hq)]).all().item()) for j in range(reps - hsets, reps))
        # ... the same synchronous call on page-locked buffers with the per-query counters asked for too: what the C++
        # drop-in's perform*Test functions do since round 3 (they pin their vectors before the timed region, gbnns_host_pin)
        qpin = batches[0].cpu().pin_memory()
        opin = {}
        ix.search(qpin, ef, want=("hops", "dist_calc", "edges"), out=opin)
        t1 = time.perf_counter()
        for _ in range(5):
            ix.search(qpin, ef, want=("hops", "dist_calc", "edges"), out=opin)
        result["host_buffers_pinned_qps"] = round(5 * nq_rank / (time.perf_counter() - t1), 1)

    # ---- CPU baseline (rank 0, N = 1 only): the compiled reference if present, else the port ----
    if world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(ds, q, ef, res["ids"], metric_id, sample=args.cpu_sample)

    if isinstance(result.get("throughput_option"), dict) and "_ids" in result["throughput_option"]:
        cbl = result.get("cpu_baseline") or {}
        if "_ref_ids" in cbl:   # measured against the reference's own answers of this run (never inferred)
            nref = len(cbl["_ref_ids"])
            result["throughput_option"]["id_mismatches_vs_reference"] = int((result["throughput_option"]["_ids"][:nref] != cbl["_ref_ids"]).sum())
            result["throughput_option"]["reference_sample"] = nref

    # ---- the other BASELINE.json configurations, each as a short run of its own in a child process --------------
    if world == 1 and args.config == "sift" and not args.no_extras and not args.no_other_configs and not args.n and not args.nq:
        ix.close()
        ix = None
        del ds
        torch.cuda.empty_cache()
        result["other_configs"] = other_configs(args, t_start)

    if args.sweep and rank == 0:
        for e in REF_EFS:
            ix.search(q, e, want=())
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(5):
                r = ix.search(q, e, want=())
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / 5
            log(f"sweep ef={e}: recall={recall_of(r['ids']):.4f} qps={nq_rank / dt:.0f}")

    if rank == 0:
        emit(result, args)
    if ix is not None:
        ix.close()
    if world > 1:
        dist.destroy_process_group()


def throughput_option(g, ix, q, ef, res_exact, batches, depth, nq):
    out = {"flag": "GBNNS_FLAG_MFMA_PROJECTION", "kernel": None, "never_the_default": True}
    try:
        fl = g.FLAG_MFMA_PROJECTION
        ex = ix.search(q, ef, want=("q_low",), flags=g.FLAG_SERIAL)
        op = ix.search(q, ef, want=("q_low",), flags=fl | g.FLAG_SERIAL)
        torch.cuda.synchronize()
        out["_ids"] = op["ids"].cpu().numpy().astype(np.int64)   # (for the comparison with the reference's own ids; dropped before printing)
        out["max_abs_q_low_err"] = float((op["q_low"] - ex["q_low"]).abs().max().item())
        out["id_mismatches_vs_exact_path"] = int((op["ids"] != res_exact["ids"]).sum().item())
        # (against the reference itself: filled in by main() from THIS run's cpu_baseline ids, where that section ran -- never an alias)
        out["batch"] = nq
        for _ in range(3):
            ix.search(q, ef, want=(), flags=fl | g.FLAG_SERIAL)
        torch.cuda.synchronize()
        ix.profile_read(reset=True)
        ix.profile_enable(True)
        for _ in range(10):
            ix.search(q, ef, want=(), flags=fl | g.FLAG_SERIAL)
        torch.cuda.synchronize()
        p = ix.profile_read(reset=True)
        ix.profile_enable(False)
        out["project_ms"] = round(p["project_ms"] / max(p["calls"], 1), 4)
        # (round 6: mlp_mfma_net_kernel -- the three layers of a 32-query tile in one launch, v_mfma_f32_32x32x2_f32, weights repacked into
        # the B-operand order, activations in LDS; nets that do not fit keep mlp_layer_mfma_kernel x 3 + normalize_kernel)
        out["kernel"] = p["project_kernel"]
        if depth > 1 and batches:
            bufs = [{} for _ in range(depth)]
            for i in range(60):
                ix.search(batches[i % len(batches)], ef, want=(), out=bufs[i % depth], flags=fl | g.FLAG_DEFER_JOIN, defer_depth=depth)
            ix.join()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(120):
                ix.search(batches[i % len(batches)], ef, want=(), out=bufs[i % depth], flags=fl | g.FLAG_DEFER_JOIN, defer_depth=depth)
            ix.join()
            torch.cuda.synchronize()
            out["value_in_flight"] = round(120 * nq / (time.perf_counter() - t1), 1)
        # matrix-pipe utilisation of the layer kernel: SQ_VALU_MFMA_BUSY_CYCLES / (1 024 SIMDs x kernel cycles), from the
        # committed counter pass of this command line (rocprofv3 cannot run inside this process)
        src = "profiles/r06_mfma_option_summary.txt" if out["kernel"] == "mlp_mfma_net_kernel" else "profiles/r05_mfma_option_summary.txt"
        mu = profile_figure(src, r"%s matrix-pipe utilisation = [^\n]*= ([0-9.]+)" % out["kernel"])
        out["mfma_util"] = mu
        out["mfma_util_source"] = src if mu is not None else None
    except Exception as e:
        out["failed"] = str(e)[-300:]
    return out


def cpu_threads_available(detail=False):
    """Host threads this process may really use: affinity mask, cgroup v2 CPU quota, and the GPU pool's share of 16 per GPU."""
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = os.cpu_count() or 1
    quota = None  # cgroup v2 CPU quota of this container, in cores
    try:
        a, b = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if a != "max":
            quota = float(a) / float(b)
    except (OSError, ValueError):
        pass
    avail = max(1, min(affinity, int(quota) if quota and quota >= 1 else affinity, 16))
    return (affinity, quota, avail) if detail else avail


def profile_figure(path, pattern):
    import re
    try:
        m = re.search(pattern, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), path)).read())
        return float(m.group(1)) if m else None
    except OSError:
        return None


def graph_prep_figures(g=None, ds=None):
    """Graph preparation (exact kNN with the matrix-core filter + GD pruning, prepare_graph.cpp:64-74) is not part of
    a timed step.  With the bench workload at hand it is measured live, once, on the workload's own low-dim base set: 48-NN lists by
    gbnns_exact_knn (checked byte for byte against the exact scan on a 4 096-row sample), GD pruning M = 16 on the device
    (gbnns_build_graph_gd_device) and the same on the host's threads -- the two graphs compared.  The matrix-pipe utilisation needs
    counters: it stays the committed builder-run figure (tools/knn_profile.sh)."""
    out = {
        "knn_mfma_util": profile_figure("profiles/r04_knn_summary.txt", r"matrix-pipe utilisation = [^=]*= ([0-9.]+)"),
        "knn_mfma_util_source": "profiles/r04_knn_summary.txt (rocprofv3 --pmc, builder-run)",
        "workload": "48-NN lists of the bench workload's low-dim base set (gbnns_exact_knn), then GD pruning M = 16 "
                    "(gbnns_build_graph_gd_device / gbnns_build_graph_gd)",
    }
    if g is None or ds is None or ds.n > 2_000_000:
        out.update({
            "knn_s": profile_figure("profiles/r04_knn_summary.txt", r"matrix-core filter \+ exact distances of the kept rows: n=1000000 d=32 k=48\s+([0-9.]+) s"),
            "knn_exact_scan_s": profile_figure("profiles/r04_knn_summary.txt", r"exact scan of every row \(rounds 1-3\): n=1000000 d=32 k=48\s+([0-9.]+) s"),
            "gd_s": profile_figure("profiles/r05_graph_prep.txt", r"with the pruning on the device: ([0-9.]+) s"),
            "source": "profiles/r04_knn_summary.txt, profiles/r05_graph_prep.txt (builder-run on the GPU box; not re-measured in this run)",
        })
        return out
    try:
        K, M = 48, 16
        x = ds.db_low.contiguous()
        n = int(x.shape[0])
        lib = g.load_library()
        step = 1 << 18
        g.exact_knn(x, x[:4096], K, self_offset=0)  # (first call: the library's buffers)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        knn = torch.cat([g.exact_knn(x, x[s0:s0 + step], K, self_offset=s0) for s0 in range(0, n, step)])
        torch.cuda.synchronize()
        knn_s = time.perf_counter() - t0
        # the filter's lists against the exact scan of every row (knob knn_filter 0) on a sample
        s0 = (n // 2) & ~63
        assert lib.gbnns_debug_knob(b"knn_filter", 0) == 0
        try:
            t0 = time.perf_counter()
            scan = g.exact_knn(x, x[s0:s0 + 4096], K, self_offset=s0)
            torch.cuda.synchronize()
            scan_s = time.perf_counter() - t0
        finally:
            lib.gbnns_debug_knob(b"knn_filter", 1)
        same = bool(torch.equal(scan, knn[s0:s0 + 4096]))
        t0 = time.perf_counter()
        g.exact_knn(x, x[s0:s0 + 4096], K, self_offset=s0)
        torch.cuda.synchronize()
        filt_s = time.perf_counter() - t0
        knn_h = knn.cpu().numpy().astype(np.uint32).reshape(-1)
        xh = x.cpu().numpy()
        koff = np.arange(n + 1, dtype=np.uint64) * np.uint64(K)
        t0 = time.perf_counter()
        off, nbr, on_host = g.build_graph_gd_device(koff, knn_h, xh, M)
        gd_s = time.perf_counter() - t0
        threads = cpu_threads_available()
        t0 = time.perf_counter()
        off2, nbr2 = g.build_graph_gd(koff, knn_h, xh, M, threads=threads)
        gd_host_s = time.perf_counter() - t0
        out.update({
            "measured": "live, this run",
            "n": n, "d_low": int(x.shape[1]), "knn_k": K, "M": M,
            "knn_s": round(knn_s, 3),
            "knn_rows_per_s": round(n / knn_s, 1),
            "knn_sample_4096_rows_s": {"matrix_core_filter": round(filt_s, 4), "exact_scan": round(scan_s, 4)},
            "knn_exact_scan_s": profile_figure("profiles/r04_knn_summary.txt", r"exact scan of every row \(rounds 1-3\): n=1000000 d=32 k=48\s+([0-9.]+) s"),
            "knn_exact_scan_s_source": "profiles/r04_knn_summary.txt (the whole 10^6-row job on the exact scan, builder-run)",
            "knn_sample_identical_to_exact_scan": same,
            "gd_s": round(gd_s, 3),
            "gd_nodes_finished_on_host": int(on_host),
            "gd_host_s": round(gd_host_s, 2),
            "gd_host_threads": threads,
            "gd_device_graph_identical_to_host": bool(np.array_equal(off, off2) and np.array_equal(nbr, nbr2)),
            "avg_degree": round(len(nbr) / n, 2),
        })
    except Exception as e:  # reported, never fatal
        out["error"] = repr(e)
    return out


def rccl_single_rank(g, ds, q, ef, metric_id, ref_ids, nq):
    out = {"form": "gbnns_multi_create(devices = [0]) + gbnns_multi_rccl_single_rank + gbnns_multi_search_device: search, "
                   "ncclAllGather on a one-rank communicator, unpadding copy"}
    # (librccl greets on stdout when it initialises a communicator; this process's stdout carries the ONE json line: the
    # greeting goes to stderr)
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        mi = g.MultiIndex(ds.base.cpu().numpy(), ds.graph_off, ds.graph_nbr, db_low=ds.db_low.cpu().numpy(),
                          net=tuple(t.cpu().numpy() for t in ds.net), metric=metric_id, devices=[0])
        mi.rccl_single_rank(True)
        outs = mi.search_device([q], ef, nq)
        mi.synchronize()
        t1 = time.perf_counter()
        for _ in range(10):
            outs = mi.search_device([q], ef, nq)
        mi.synchronize()
        out.update({"rccl_version": mi.rccl_version(), "ms_per_step": round((time.perf_counter() - t1) * 100, 4),
                    "ids_identical_to_single_handle": bool(torch.equal(outs[0].view(ref_ids.dtype), ref_ids))})
        mi.close()
    except Exception as e:  # reported, never fatal
        out["failed"] = str(e)[-300:]
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)
    return out


def capi_multi_parent(args, argv, world, pieces, rehearsal):
    """Rank 0, everyone else parked: starts the child below and relays its JSON."""
    import tempfile
    ref_path = ""
    if pieces is not None:
        fd, ref_path = tempfile.mkstemp(suffix=".npy", prefix="gbnns_rank_ids_")
        os.close(fd)
        np.save(ref_path, torch.cat(pieces).cpu().numpy())
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "GBNNS_BENCH_ARGV", "MASTER_PORT",
                                                            "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "TORCHELASTIC_RUN_ID")}
    child_argv = [a for a in argv if a != "--capi-multi-child"]
    cmd = [sys.executable, os.path.abspath(__file__)] + child_argv + ["--capi-multi-child", ref_path or "-"]
    try:
        pr = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
        line = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
        if pr.returncode != 0 or not line:
            return {"failed": "child rc %d: %s" % (pr.returncode, (pr.stderr or "")[-400:])}
        return json.loads(line[-1])
    except subprocess.TimeoutExpired:
        return {"failed": "no answer from the child within 300 s"}
    finally:
        if ref_path:
            try:
                os.unlink(ref_path)
            except OSError:
                pass


def capi_multi_child(args, cfg):
    """`--capi-multi-child <ids.npy>`: one process, N = --gpus replicas through gbnns_multi_* (DESIGN.md 7).  The batch is
    the concatenation of the ranks' batches 0 (weak) / the one batch (strong), block r resident on device r; K steps of
    gbnns_multi_search_device (search + one RCCL all-gather of the ids each), synchronised at the end."""
    import gbnns_dim_red_amd as g
    from gbnns_dim_red_amd import synth
    g.load_library()
    world = args.gpus
    rehearsal = os.environ.get("GBNNS_BENCH_REHEARSAL") == "1"
    out = {"ranks": world, "form": "device blocks + RCCL all-gather (gbnns_multi_search_device)", "rccl_loaded": None}
    try:
        kw = dict(n=cfg["n"], nq=cfg["nq"], d=cfg["d"], d_low=cfg["d_low"], d_hidden=cfg["d_hidden"], seed=1234,
                  cache_dir=args.cache_dir, native_knn=bool(cfg.get("native_knn")) and cfg["n"] > 2_000_000)
        if cfg.get("unit_norm"):
            kw["unit_norm"] = True
        if args.graph_M:
            kw["M"] = args.graph_M
        if cfg.get("strong") and cfg["nq"] > 200_000:
            kw["gt_queries"] = 20_000
        if cfg.get("recipe"):
            kw.update(cfg["recipe"])
        ds = synth.make_dataset(device="cuda:0", **kw)
        ef = args.ef or cfg["ef"]
        metric_id = g.METRIC_NEG_DOT if cfg.get("negdot") else g.METRIC_L2
        strong = bool(cfg.get("strong"))
        if strong:
            whole = ds.queries
        else:
            whole = torch.cat([torch.roll(ds.queries, shifts=-((r * ds.nq) // world), dims=0) for r in range(world)])
        n_q = int(whole.shape[0])
        devs = [0] * world if rehearsal else list(range(world))
        mi = g.MultiIndex(ds.base.cpu().numpy(), ds.graph_off, ds.graph_nbr, db_low=ds.db_low.cpu().numpy(),
                          net=tuple(t.cpu().numpy() for t in ds.net), metric=metric_id, devices=devs)
        out["devices"] = mi.devices
        ref = None if args.capi_multi_child in ("", "-") else np.load(args.capi_multi_child)
        steps = max(1, min(args.steps or 20, 50))
        if rehearsal:
            # one GPU: the replicas share it, RCCL needs distinct devices -> the host form (gbnns_multi_search_ex, no collective)
            out["form"] = "host buffers, no collective (gbnns_multi_search_ex): rehearsal on one GPU"
            wq = whole.cpu().numpy()
            r = mi.search(wq, ef, want=())
            t1 = time.perf_counter()
            for _ in range(steps):
                r = mi.search(wq, ef, want=())
            dt = time.perf_counter() - t1
            got = r["ids"].astype(np.int64)
        else:
            blocks = []
            for r_ in range(world):
                lo, hi = mi.shard_bounds(n_q, r_)
                blocks.append(whole[lo:hi].to("cuda:%d" % mi.devices[r_]).contiguous())
            for r_ in range(world):
                torch.cuda.synchronize(mi.devices[r_])
            outs = mi.search_device(blocks, ef, n_q)
            mi.synchronize()
            out["rccl_loaded"] = True
            t1 = time.perf_counter()
            for _ in range(steps):
                outs = mi.search_device(blocks, ef, n_q)
            mi.synchronize()
            dt = time.perf_counter() - t1
            got = outs[0].cpu().numpy().astype(np.int64)
            out["every_replica_holds_all_ids"] = all(bool((o.cpu().numpy().astype(np.int64) == got).all()) for o in outs)
        out.update({"queries_per_s": round(steps * n_q / dt, 1), "steps": steps, "ms_per_step": round(dt * 1e3 / steps, 4),
                    "batch": n_q, "ef": ef,
                    "ids_identical_to_rank_results": None if ref is None else bool((got == ref.astype(np.int64)).all())})
        mi.close()
    except Exception as e:  # RCCL missing / refused, a device out of reach, ...: reported, never fatal
        msg = str(e)
        if "RCCL" in msg or "rccl" in msg or "nccl" in msg:
            out["rccl_loaded"] = False
        out["failed"] = msg[-400:]
    print(json.dumps(out), flush=True)


def other_configs(args, t_start):
    """BASELINE.json configurations 3 - 5 as short runs of their own (`python bench.py --config <name> --no-extras
    --cpu-sample 1000`), each in a fresh child process started after this one has released its index: GIST1M-shaped
    (ef 200, the reference's 1 000-query batch), GloVe-shaped with the L2 and the negative-dot metric (ef 64) and,
    while the wall-clock budget lasts, DEEP10M-shaped at full size (n = 10^7, ONE 1 M-query batch).  Per configuration:
    the same figures as the headline line's (in-flight value, serial rate, first-pass kernel, its time, algorithmic
    bytes, roofline fraction, recall) and the answers of a 1 000-query sample compared with the compiled reference."""
    # ("sift-M30": the headline workload on the graph of prepare_graph.cpp's M = 30 -- adjacency rows of up to 60 slots, the
    # walk_hotw* instances: the library's tuning constants on a second degree distribution, every round)
    # ("deep1m" / "glove1m": the two rows of the reference's own parameter file that BASELINE.json does not name -- deep 96 -> 48 at
    # ef 40, glove 300 -> 144 at ef 300: 192- and 576-byte walked rows)
    plan = [("gist", 20, 150), ("glove", 20, 120), ("glove-dot", 20, 90), ("sift-M30", 20, 120), ("deep1m", 20, 100), ("glove1m", 10, 120),
            ("deep", 3, 200),
            # (round 6, after DEEP10M so that they never cost it its slot) the headline shape on a harder data recipe, at its own recall
            # gate, and on a GD(M = 24) graph -- adjacency rows of up to 48 slots, what hnswlib's M = 18 level-0 lists look like
            ("sift-hard", 20, 80), ("sift-M24", 20, 80)]
    out = {}
    for name, steps, need_s in plan:
        left = args.budget_s - (time.time() - t_start)
        if left < need_s:
            out[name] = {"skipped": "%.0f s of the %.0f s budget left, this configuration is given %d s" % (left, args.budget_s, need_s)}
            continue
        cmd = [sys.executable, os.path.abspath(__file__), "--config", name.split("-M")[0] if "-M" in name else name, "--steps", str(steps),
               "--warmup", "3" if steps > 3 else "1", "--no-extras", "--cpu-sample", "1000", "--cache-dir", args.cache_dir, "--full-line"]
        if "-M" in name:
            cmd += ["--graph-M", name.split("-M")[1]]
        t1 = time.time()
        try:
            pr = subprocess.run(cmd, capture_output=True, text=True, timeout=need_s)
            line = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
            if pr.returncode != 0 or not line:
                out[name] = {"failed": "rc %d: %s" % (pr.returncode, (pr.stderr or "")[-300:])}
                continue
            j = json.loads(line[-1])
        except subprocess.TimeoutExpired:
            out[name] = {"failed": "no line within %d s" % need_s}
            continue
        rl, cb = j["roofline"], j.get("cpu_baseline", {})
        out[name] = {
            "workload": j["config"]["workload"], "ef": j["config"]["ef"], "recall_at_1": j["config"]["recall_at_1"],
            "value": j["value"], "serial": j["serial"]["queries_per_s"], "steps": j["steps"], "ms_per_step": j["ms_per_step"],
            "kernel": rl["kernel"], "kernel_ms": rl["kernel_ms"], "algorithmic_bytes_per_launch": rl["algorithmic_bytes_per_launch"],
            "achieved_GBps": rl["achieved"], "frac": rl["frac"], "project_ms": j["kernels_ms"]["project"],
            "gpu_ids_identical": cb.get("gpu_ids_identical"), "cpu_sample": cb.get("sample"), "cpu_value": cb.get("value"),
            "cpu_cores": cb.get("cores"), "cpu_kind": cb.get("kind"), "wall_s": round(time.time() - t1, 1),
        }
        if j.get("recall_gate_failed"):
            out[name]["recall_gate_failed"] = True
    return out


def counters_for(config, ef):
    """Committed PMC figures of this configuration's dominant kernel (profiles/counters_latest.json, written from
    the rocprofv3 --pmc passes by tools/digest_profile.py); bench.py cannot read hardware counters itself."""
    path = os.path.join(ROOT, "profiles", "counters_latest.json")
    try:
        return json.load(open(path)).get("%s:ef%d" % (config, ef))
    except Exception:
        return None


def roofline_of(ds, res, prof, ef, nq, max_degree, cfg, rank, launches=1):
    """`res` holds the walk counters of `launches` batches of nq queries each (concatenated)."""
    dc = res["dist_calc"].double()
    hops = res["hops"].double()
    edges = res["edges"].double()
    if rank == 0:
        qs = torch.tensor([0.5, 0.9, 0.99, 0.999, 1.0], dtype=torch.float64, device=dc.device)
        log("ef", ef, "dist_calc quantiles 50/90/99/99.9/max:", [int(v) for v in torch.quantile(dc[:1 << 20], qs).tolist()],
            "hops max", int(hops.max().item()))
    d_low, d = ds.d_low, ds.d
    calls = max(prof["calls"], 1)
    walk_bytes = (dc * 4 * d_low + edges * 4 + hops * 8 + 4 * d_low + 4 * ef).sum().item() / launches
    rerank_bytes = nq * (ef * 4.0 * d + 4 * d + 4 + 4 * ef)
    walk_ms = prof["walk_ms"] / calls
    rerank_ms = prof["rerank_ms"] / calls
    project_ms = prof["project_ms"] / calls
    general_ms = prof["walk_general_ms"] / calls
    # the register-list / two-list first passes (ef <= 1024) re-rank at the end of the walk: then no re-rank kernel is
    # launched and the library's re-rank interval is an empty pair of events (~0.006 ms)
    # (L2: d % 8 == 4 too -- glove's 300 -- since round 5)
    fused = (ds.d % 8 == 0 or (ds.d % 4 == 0 and not cfg.get("negdot"))) and rerank_ms < 0.02
    kernel_bytes = walk_bytes + (rerank_bytes if fused else 0.0)
    achieved = kernel_bytes / (walk_ms * 1e-3) / 1e9 if walk_ms > 0 else 0.0
    pmc = counters_for(cfg["name"], ef)
    roof = {
        "bound": "hbm",
        # the walked tables (db_low + adjacency) of the 10^6-node shapes fit the 256 MB Infinity Cache: the peak the
        # fraction is taken against is still the HBM spec figure (8 TB/s), i.e. a ceiling for any mix of the two
        "served_from": "HBM + Infinity Cache (MALL)",
        # the first-pass kernel the library launched (gbnns_profile.walk_kernel), template arguments included
        "kernel": prof["walk_kernel"] + (" (walk + fused re-rank)" if fused else ""),
        "achieved": round(achieved, 1),
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4),
        # PMC bytes per launch: FETCH_SIZE x2 + WRITE_SIZE (the guide's gfx950 correction, calibrated for wide
        # coalesced streams only); the raw sum is given beside it -- the truth lies between the two
        "traffic": pmc.get("hbm_bytes_corrected") if pmc else None,
        "traffic_raw": pmc.get("hbm_bytes_raw") if pmc else None,
        "traffic_source": pmc.get("source") if pmc else None,
        "algorithmic_bytes_per_launch": round(kernel_bytes),
        "algorithmic_bytes_walk_part": round(walk_bytes),
        "algorithmic_bytes_rerank_part": round(rerank_bytes if fused else 0),
        "kernel_ms": round(walk_ms, 4),
        "kernel_ms_mode": "serialised steps (one stream, kernels back to back), hipEvent pairs on the launch stream; a launch that runs "
                          "alone lets its last, partial round of wavefronts request rows before the visited test (library knob "
                          "spec_tail; batches in flight -- `value` -- do not)",
    }
    if pmc and pmc.get("valu_insts"):
        # second roofline: vector-instruction issue.  One VALU instruction occupies its SIMD's issue port for 4
        # cycles (MI355X_MICROARCH.md, "vector-instruction ISSUE cost"); 1024 SIMDs; cycles from the kernel time
        # at the clock the PMC pass observed (GRBM_GUI_ACTIVE / 8 per launch).
        cyc = pmc.get("gpu_cycles") or walk_ms * 1e-3 * 2.4e9
        roof["valu_issue"] = {"insts_per_launch": pmc["valu_insts"], "issue_cycles_each": 4,
                              "busy_frac": round(pmc["valu_insts"] * 4.0 / (1024.0 * cyc), 4),
                              "kernel_cycles": round(cyc)}
    kernels = {"project": round(project_ms, 4), "walk": round(walk_ms, 4), "walk_general": round(general_ms, 4),
               "rerank": None if fused else round(rerank_ms, 4),  # fused: inside the walk kernel
               "rerank_GBps": (round(rerank_bytes / (rerank_ms * 1e-3) / 1e9, 1) if rerank_ms > 0 and not fused else None),
               "general_queries": prof["general_queries"]}
    return dict(roofline=roof, kernels_ms=kernels, fused=fused, walk_bytes=walk_bytes, rerank_bytes=rerank_bytes)


def ef_sweep(ix, ds, q, cfg, ef0, recalls, recall_of, nq, max_degree, rank, batches=None, depth=1):
    """Every ef of the configuration (and, for sift, the reference's top efs): wall time per step, kernel time,
    algorithmic bytes and the roofline fraction -- the same derivation as the headline's, 3 + 8 serialised steps each
    -- and the rate with `depth` batches in flight over the rotating batches (as the headline's `value`)."""
    import gbnns_dim_red_amd as g
    out = []
    reps = 8 if nq <= 20_000 else 2
    for e in sorted(set([ef0] + cfg["efs"])):
        for _ in range(3 if nq <= 20_000 else 1):
            ix.search(q, e, want=())
        torch.cuda.synchronize()
        ix.profile_read(reset=True)
        ix.profile_enable(True)
        t1 = time.perf_counter()
        for _ in range(reps):
            r = ix.search(q, e, want=("hops", "dist_calc", "edges"))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / reps
        p = ix.profile_read(reset=True)
        ix.profile_enable(False)
        rl = roofline_of(ds, r, p, e, nq, max_degree, cfg, rank)
        rec = recalls.get(e)
        if rec is None:
            rec = recall_of(r["ids"])
        flight = None
        if batches and depth > 1:
            bufs = [{} for _ in range(depth)]
            # (an untimed stretch of ~30 ms first, as in front of the headline's steps: the device's first milliseconds with
            # batches in flight are 10 % slower than its steady state; then at least ~20 ms of timed steps)
            for i in range(max(2 * depth, min(200, int(0.03 / dt) + 1))):
                ix.search(batches[i % len(batches)], e, want=(), out=bufs[i % depth], flags=g.FLAG_DEFER_JOIN, defer_depth=depth)
            ix.join()
            torch.cuda.synchronize()
            nrep = max(4 * reps, min(200, int(0.02 / dt) + 1)) if nq <= 20_000 else 4 * reps
            t2 = time.perf_counter()
            for i in range(nrep):
                ix.search(batches[i % len(batches)], e, want=(), out=bufs[i % depth], flags=g.FLAG_DEFER_JOIN, defer_depth=depth)
            ix.join()
            torch.cuda.synchronize()
            flight = nrep * nq / (time.perf_counter() - t2)
        out.append({"ef": e, "recall_at_1": round(rec, 4), "queries_per_s": round(nq / dt, 1),
                    "queries_per_s_in_flight": round(flight, 1) if flight else None,
                    "ms_per_step": round(dt * 1e3, 4), "kernel": rl["roofline"]["kernel"],
                    "kernel_ms": rl["roofline"]["kernel_ms"], "rerank_ms": rl["kernels_ms"]["rerank"],
                    "algorithmic_bytes_per_launch": rl["roofline"]["algorithmic_bytes_per_launch"],
                    "achieved_GBps": rl["roofline"]["achieved"], "frac": rl["roofline"]["frac"],
                    "mean_dist_calc": round(r["dist_calc"].double().mean().item(), 1),
                    "mean_hops": round(r["hops"].double().mean().item(), 1)})
    return out


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(ds, q, ef, gpu_ids, metric_id, sample=0):
    """Times the reference's per-query body on host cores over the SAME batch and checks that the
    GPU answers are identical.  oracle/ is used here only as the reported baseline / checker.
    sample > 0: the first `sample` queries on the widest thread count only (the other configurations' child runs)."""
    import oracle
    base = ds.base.cpu().numpy()
    dbl = ds.db_low.cpu().numpy()
    net = tuple(t.cpu().numpy() for t in ds.net)
    qh = q.cpu().numpy()
    off, nbr = ds.graph_off, ds.graph_nbr
    if oracle.have_ref():
        impl, kind = oracle.Ref(), "reference"
        impl.prepare(base)
        impl.search_batch(oracle.MODE_NET, qh[:8], base, off, nbr, ef, db_low=dbl, net=net, metric=metric_id)  # graph conv
        flags = "g++ -O2 -std=c++11 -fopenmp -mavx2 -mfma -ffp-contract=off (oracle/Makefile REF_FLAGS: the strict-IEEE " \
                "build the golden vectors come from; ~10 % slower than the README's -Ofast -march=native, BASELINE.md 2)"
    else:
        impl, kind = oracle.Oracle(), "port"
        flags = "g++ -O2 -std=c++17 -fopenmp -ffp-contract=off -fno-fast-math (oracle/Makefile ORACLE_FLAGS)"
    omp_max = impl.max_threads()   # honours OMP_NUM_THREADS (the GPU boxes export 1); explicit counts below override it
    affinity, quota, avail = cpu_threads_available(detail=True)

    def run(nqs, threads):
        t0 = time.perf_counter()
        r = impl.search_batch(oracle.MODE_NET, qh[:nqs], base, off, nbr, ef, db_low=dbl, net=net, threads=threads,
                              metric=metric_id)
        return time.perf_counter() - t0, r

    if sample > 0:
        ns = min(len(qh), sample)
        tn, r = run(ns, avail)
        same = int((r["ids"].astype(np.int64) == gpu_ids[:ns].cpu().numpy().astype(np.int64)).sum())
        return {"value": round(ns / tn, 1), "unit": "queries/s", "cores": avail, "kind": kind,
                "sample": "the first %d queries of the batch at ef=%d, OpenMP over queries (search_function.h:152) on %d thread(s)" % (ns, ef, avail),
                "cpu_model": cpu_model(), "build_flags": flags, "gpu_ids_identical": same == ns, "gpu_id_mismatches": ns - same}
    # 1 thread (what final_test.cpp ships, :71) on a bounded sample sized from a probe (~4 s of work) ...
    probe = min(len(qh), 200)
    t_probe, _ = run(probe, 1)
    per_q = t_probe / probe
    ns = int(max(probe, min(len(qh), 4.0 / max(per_q, 1e-9))))
    t1, _ = run(ns, 1)
    # ... then the OpenMP form (search_function.h:152) at 2, 4, ... up to the cores this process may use, each on a
    # sample of ~2.5 s; the widest run covers the whole batch when that fits ~10 s and is the id check
    scaling = {"1": round(ns / t1, 1)}
    th = 2
    while th < avail:
        nst = int(max(probe, min(len(qh), 2.5 * th / max(per_q, 1e-9))))
        tt, _ = run(nst, th)
        scaling[str(th)] = round(nst / tt, 1)
        th *= 2
    nfull = int(max(probe, min(len(qh), 10.0 * avail / max(per_q, 1e-9))))
    tn, r = run(nfull, avail)
    scaling[str(avail)] = round(nfull / tn, 1)
    same = int((r["ids"].astype(np.int64) == gpu_ids[:nfull].cpu().numpy().astype(np.int64)).sum())
    # the same sample on the reference built with the README's own flags (README.md:33: -Ofast -march=native; built in the
    # build container, so first probed in a child process: an instruction this host lacks must not take the bench down)
    readme = {"value_readme_flags": None}
    if kind == "reference" and oracle.have_ref_fast():
        probe_rc = subprocess.run([sys.executable, "-c",
                                   "import sys; sys.path.insert(0, %r); import oracle, numpy as np; "
                                   "r = oracle.Ref(oracle.REF_FAST_SO); "
                                   "a = np.arange(64, dtype=np.float32); print(float(r.l2(a, a[::-1].copy())))" % ROOT],
                                  capture_output=True).returncode
        if probe_rc == 0:
            fast = oracle.Ref(oracle.REF_FAST_SO)
            fast.prepare(base)
            fast.search_batch(oracle.MODE_NET, qh[:8], base, off, nbr, ef, db_low=dbl, net=net, metric=metric_id)
            t0 = time.perf_counter()
            rf = fast.search_batch(oracle.MODE_NET, qh[:nfull], base, off, nbr, ef, db_low=dbl, net=net, threads=avail,
                                   metric=metric_id)
            tf = time.perf_counter() - t0
            t0 = time.perf_counter()
            fast.search_batch(oracle.MODE_NET, qh[:ns], base, off, nbr, ef, db_low=dbl, net=net, threads=1, metric=metric_id)
            tf1 = time.perf_counter() - t0
            readme = {"value_readme_flags": round(nfull / tf, 1), "value_readme_flags_1thread": round(ns / tf1, 1),
                      "readme_flags": "g++ -Ofast -std=c++11 -fopenmp -march=native -ftree-vectorize (README.md:33), built on "
                                      "the build container's CPU; host-dependent arithmetic (SURVEY F5), timing only",
                      "readme_flags_ids_identical_to_strict": bool((rf["ids"] == r["ids"]).all())}
        else:
            readme["readme_flags"] = "the -march=native build of the build container does not run on this host (probe rc %d)" % probe_rc
    return {
        **readme,
        "value": round(nfull / tn, 1),
        "unit": "queries/s",
        "cores": avail,
        "kind": kind,
        "sample": "the first %d queries of the batch at ef=%d, OpenMP over queries (search_function.h:152) on %d "
                  "thread(s); 1-thread figure on the first %d queries" % (nfull, ef, avail, ns),
        "value_1thread": round(ns / t1, 1),
        "queries_per_s_by_threads": scaling,
        "cpu_model": cpu_model(),
        "os_cpu_count": os.cpu_count(),
        "sched_affinity_cpus": affinity,
        "omp_max_threads": omp_max,
        "env_OMP_NUM_THREADS": os.environ.get("OMP_NUM_THREADS"),
        "cgroup_cpu_quota_cores": quota,
        "cores_available": avail,
        "cores_rule": "min(sched affinity, cgroup cpu.max, 16 = the GPU pool's CPU share per GPU); thread counts are set "
                      "explicitly (omp_set_num_threads), whatever OMP_NUM_THREADS says",
        "build_flags": flags,
        "gpu_ids_identical": same == nfull,
        "gpu_id_mismatches": nfull - same,
        "_ref_ids": r["ids"].astype(np.int64),   # (the reference's answers to the first nfull queries; dropped before printing)
    }


def emit(result, args):
    """Rank 0's output.  stdout gets ONE line: the compact record (every contract field, `roofline`, `cpu_baseline`, and one-row summaries
    of the diagnostic sections; under 8 KB so that a driver that keeps a bounded tail of stdout keeps all of it) -- or the complete record
    with --full-line.  The complete record always goes to stderr as one line prefixed `BENCH_FULL ` and, when GBNNS_BENCH_FULL names a file
    (or gpurun_out/ exists), into that file."""
    result = {k: v for k, v in result.items() if not k.startswith("_")}
    for sec in ("throughput_option", "cpu_baseline"):
        if isinstance(result.get(sec), dict):
            result[sec] = {k: v for k, v in result[sec].items() if not k.startswith("_")}
    full = json.dumps(result)
    log("BENCH_FULL " + full)
    path = os.environ.get("GBNNS_BENCH_FULL")
    if not path and os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        path = os.path.join(ROOT, "gpurun_out", "bench_full_%s.json" % result["config"]["name"])
    if path:
        try:
            with open(path, "w") as f:
                f.write(full + "\n")
        except OSError:
            pass
    if args.full_line or (args.no_extras and len(full) <= 8000):   # (a --no-extras record is small: tools read it whole)
        print(full, flush=True)
        return
    line = json.dumps(compact(result))
    if len(line) > 8000:   # never silently: drop the least essential summaries until it fits, and say so
        c = compact(result)
        for k in ("graph_prep", "rccl_single_rank", "cold", "separate_stages", "host", "throughput_option"):
            if len(json.dumps(c)) <= 8000:
                break
            c.pop(k, None)
            c.setdefault("dropped_for_size", []).append(k)
        line = json.dumps(c)
    print(line, flush=True)


def compact(r):
    """The record in under 8 KB: contract fields verbatim; sections as rows (the column names ride along once)."""
    def pick(d, keys):
        return {k: d[k] for k in keys if isinstance(d, dict) and k in d}
    mq = lambda v: None if v is None else round(v / 1e6, 3)
    c = pick(r, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "settle_steps_before_warmup", "ms_per_step", "higher_is_better",
                 "scaling", "vs_baseline", "dtype", "data"))
    cfg = r["config"]
    c["config"] = pick(cfg, ("workload", "name", "ef", "recall_at_1", "recall_scored_queries", "mean_hops", "mean_dist_calc", "graph", "parallelism"))
    c["config"]["recall_sweep"] = cfg.get("recall_sweep")
    c["config"]["recipe"] = cfg.get("recipe")
    rl = dict(r["roofline"])
    rl.pop("kernel_ms_mode", None)
    if isinstance(rl.get("traffic_source"), str):
        rl["traffic_source"] = rl["traffic_source"].split(" (")[0]
    c["roofline"] = rl
    c["kernels_ms"] = r.get("kernels_ms")
    c.update(pick(r, ("ranks_seen", "gather_self_check", "batches_rotated", "pipelined", "batches_in_flight", "recall_gate_failed",
                      "per_rank", "strong_batch")))
    if "serial" in r:
        c["serial"] = pick(r["serial"], ("ms_per_step", "queries_per_s"))
    c["value_definition"] = "all ranks' queries / wall time of K steps; inputs, outputs and index resident in HBM; %s" % (
        "%d batches in flight inside the library (GBNNS_FLAG_DEFER_JOIN), joins and drains timed" % r["batches_in_flight"]
        if r.get("pipelined") else "one batch at a time")
    cb = r.get("cpu_baseline")
    if cb:
        c["cpu_baseline"] = pick(cb, ("value", "unit", "cores", "kind", "sample", "value_1thread", "queries_per_s_by_threads", "cpu_model",
                                      "value_readme_flags", "value_readme_flags_1thread", "readme_flags_ids_identical_to_strict",
                                      "gpu_ids_identical", "gpu_id_mismatches"))
    if "cold" in r:
        c["cold"] = pick(r["cold"], ("first_call_ms", "second_call_ms", "calls_until_steady", "steady_call_ms"))
    if "best_ef_at_recall_gate" in r:
        b = r["best_ef_at_recall_gate"]
        c["best_ef_at_recall_gate"] = pick(b, ("ef", "recall_at_1", "queries_per_s", "queries_per_s_in_flight", "kernel_ms", "achieved_GBps", "frac"))
    if "ef_sweep" in r:
        c["ef_sweep"] = {"columns": ["ef", "recall_at_1", "Mq_per_s_serial", "Mq_per_s_in_flight", "kernel", "kernel_ms", "frac", "frac_whole_step_in_flight"],
                         "rows": [[e["ef"], e["recall_at_1"], mq(e["queries_per_s"]), mq(e["queries_per_s_in_flight"]),
                                   e["kernel"].split("<")[0].split(" (")[0], e["kernel_ms"], e["frac"],
                                   None if not e.get("queries_per_s_in_flight") else
                                   round(e["algorithmic_bytes_per_launch"] * e["queries_per_s_in_flight"] / r["config_nq"] / 1e9 / HBM_PEAK_GBS, 4)]
                                  for e in r["ef_sweep"]]}
    if "separate_stages" in r:
        c["separate_stages"] = r["separate_stages"]
    host = pick(r, ("survey_8d_value", "survey_8d_in_flight", "host_buffers_pinned_qps", "host_batches_in_flight_ids_identical",
                    "survey_8d_blocks"))
    if host:
        c["host"] = host
        c["survey_8d_value"] = r.get("survey_8d_value")
        c["survey_8d_in_flight"] = r.get("survey_8d_in_flight")
    if "throughput_option" in r:
        c["throughput_option"] = pick(r["throughput_option"], ("kernel", "project_ms", "mfma_util", "mfma_util_source", "value_in_flight",
                                                               "max_abs_q_low_err", "id_mismatches_vs_exact_path", "id_mismatches_vs_reference",
                                                               "reference_sample", "batch", "failed", "project_ms_exact"))
        if "kernel" in c["throughput_option"]:
            c["throughput_option"]["kernel"] = c["throughput_option"]["kernel"].split(" (")[0]
    if "graph_prep" in r:
        c["graph_prep"] = pick(r["graph_prep"], ("measured", "knn_s", "gd_s", "gd_host_s", "gd_host_threads", "knn_mfma_util",
                                                 "knn_sample_identical_to_exact_scan", "gd_device_graph_identical_to_host", "error"))
    if "rccl_single_rank" in r:
        c["rccl_single_rank"] = pick(r["rccl_single_rank"], ("rccl_version", "ms_per_step", "ids_identical_to_single_handle", "failed"))
    if "capi_multi" in r:
        c["capi_multi"] = pick(r["capi_multi"], ("ranks", "queries_per_s", "ms_per_step", "rccl_loaded", "ids_identical_to_rank_results",
                                                 "every_replica_holds_all_ids", "failed"))
    oc = r.get("other_configs")
    if oc:
        cols = ["ef", "recall_at_1", "Mq_per_s_in_flight", "Mq_per_s_serial", "kernel", "kernel_ms", "frac", "project_ms", "gpu_ids_identical",
                "cpu_q_per_s", "cpu_cores"]
        rows = {}
        for name, o in oc.items():
            if "value" not in o:
                rows[name] = o.get("skipped") or o.get("failed")
                continue
            rows[name] = [o["ef"], o["recall_at_1"], mq(o["value"]), mq(o["serial"]), o["kernel"].split(" (")[0].replace("false", "0").replace("true", "1"),
                          o["kernel_ms"], o["frac"], o["project_ms"], o["gpu_ids_identical"], o["cpu_value"], o["cpu_cores"]]
            if o.get("recall_gate_failed"):
                rows[name].append("recall_gate_failed")
        c["other_configs"] = {"columns": cols, "rows": rows}
    c["full_record"] = "stderr line `BENCH_FULL {...}` of this run (and gpurun_out/bench_full_<config>.json where that directory exists)"
    return c


if __name__ == "__main__":
    main()
