"""Synthetic benchmark inputs shaped like the reference's datasets (no datasets / network here).

Recipe (constants recorded in `Dataset.recipe` and printed by bench.py):

  base / queries  low-intrinsic-dimension mixture on a sphere, embedded in R^d with noise:
                  x = normalise(c_j + s * g) @ A + sigma * eps,  g in R^m, A in R^{m x d} with
                  orthonormal rows, j a random cluster.
  net             the 3-layer ReLU MLP of the reference (support_func.h:645-658) with weights that
                  realise the linear map P = [A; B] (B: d_low - m further orthonormal directions)
                  exactly: layer 1 rows [P; -P], layer 2 identity on those 2*d_low units, layer 3
                  [I, -I].  Needs d_hidden >= 2*d_low.  It is applied by the product's own
                  projection kernel (bit-identical to the reference's GetLowQueryFromNet).
  db_low          net(base)           -- gbnns_project, i.e. the `_base_angular_optimal.fvecs` file
  graph           exact kNN (K) in the low-dim space -> GD pruning (hnswlikeGD, M, reverse edges),
                  i.e. what prepare_graph.cpp produces -- gbnns_build_graph_gd
  ground truth    exact nearest neighbour in the original space (top-2 kept for the duplicate rule)

torch (ROCm) is used here only to synthesise and hold data; nothing in this file is on the
timed search path.
"""
import hashlib
import os
import sys
import time

import numpy as np
import torch

from . import binding


class Dataset:
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def index(self, device_index=0, metric=binding.METRIC_L2):
        """Index over the resident tensors (borrowed when they are on the GPU)."""
        return binding.Index(self.base, self.graph_off, self.graph_nbr, db_low=self.db_low,
                             net=self.net, metric=metric, device=device_index)


def _orthonormal_rows(gen, rows, cols, device):
    a = torch.randn(cols, rows, generator=gen, device=device, dtype=torch.float64)
    q, _ = torch.linalg.qr(a)
    return q.t().contiguous().to(torch.float32)  # [rows x cols]


def make_net(P, d_hidden):
    """[W|b] layers (reference file layout, final_test.cpp:73-76) realising x -> P x."""
    dl, d = P.shape
    assert d_hidden >= 2 * dl, "projection net needs d_hidden >= 2*d_low"
    dev = P.device
    l1 = torch.zeros(d_hidden, d + 1, device=dev)
    l1[:dl, :d] = P
    l1[dl:2 * dl, :d] = -P
    l2 = torch.zeros(d_hidden, d_hidden + 1, device=dev)
    idx = torch.arange(2 * dl, device=dev)
    l2[idx, idx] = 1.0
    l3 = torch.zeros(dl, d_hidden + 1, device=dev)
    j = torch.arange(dl, device=dev)
    l3[j, j] = 1.0
    l3[j, j + dl] = -1.0
    return l1.contiguous(), l2.contiguous(), l3.contiguous()


def project(net, x):
    """Applies the net with the product's projection kernel (gbnns_project)."""
    dev = x.is_cuda
    n1 = x[:1].contiguous()
    off = np.array([0, 0], np.uint64)
    nbr = np.zeros(0, np.uint32)
    dl = net[2].shape[0]
    low = torch.zeros(1, dl, device=x.device) if dev else np.zeros((1, dl), np.float32)
    if not dev:
        n1 = n1.numpy()
        net = tuple(t.numpy() for t in net)
        x = x.numpy()
    ix = binding.Index(n1, off, nbr, db_low=low, net=net,
                       device=x.device.index or 0 if dev else 0)
    try:
        out = ix.project(x)
        if dev:
            torch.cuda.synchronize()
    finally:
        ix.close()
    return out if dev else torch.from_numpy(out)


def knn_exact(x, k, chunk=4096):
    """Exact k nearest neighbours of every row of x within x (self excluded), squared L2."""
    n = x.shape[0]
    sq = (x * x).sum(1)
    out = torch.empty(n, k, dtype=torch.int32, device=x.device)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        dm = sq[s:e, None] + sq[None, :] - 2.0 * (x[s:e] @ x.t())
        dm[torch.arange(e - s, device=x.device), torch.arange(s, e, device=x.device)] = float("inf")
        out[s:e] = dm.topk(k, dim=1, largest=False, sorted=True).indices.to(torch.int32)
        del dm
    return out


def ground_truth(base, queries, k=2, chunk=2048):
    sq = (base * base).sum(1)
    out = torch.empty(queries.shape[0], k, dtype=torch.int64, device=base.device)
    for s in range(0, queries.shape[0], chunk):
        q = queries[s:s + chunk]
        dm = sq[None, :] - 2.0 * (q @ base.t())
        out[s:s + chunk] = dm.topk(k, dim=1, largest=False, sorted=True).indices
        del dm
    return out


def more_queries(ds, count, batch=0):
    """`count` further query vectors from the dataset's distribution (same clusters, same embedding, an
    independent random stream per `batch`): distinct batches for a timed loop.  No ground truth."""
    r = ds.recipe
    dev = ds.base.device
    gen = torch.Generator(device=dev)
    gen.manual_seed(r["seed"])
    m = r["intrinsic"]
    A = _orthonormal_rows(gen, r["d_low"], r["d"], dev)  # the first two draws of make_dataset, replayed
    centers = torch.randn(r["n_clusters"], m, generator=gen, device=dev)
    centers = centers / centers.norm(dim=1, keepdim=True)
    g2 = torch.Generator(device=dev)
    g2.manual_seed(r["seed"] + 7919 * (batch + 1))
    j = torch.randint(0, r["n_clusters"], (count,), generator=g2, device=dev)
    z = centers[j] + r["cluster_scale"] * torch.randn(count, m, generator=g2, device=dev)
    z = z / z.norm(dim=1, keepdim=True)
    x = z @ A[:m] + r["sigma"] * torch.randn(count, r["d"], generator=g2, device=dev)
    if r.get("unit_norm"):
        x = x / x.norm(dim=1, keepdim=True)
    return x.contiguous()


def make_dataset(n=1_000_000, nq=10_000, d=128, d_low=32, d_hidden=256, seed=1234,
                 device="cuda:0", intrinsic=16, n_clusters=1000, cluster_scale=0.5, sigma=0.03,
                 knn_k=48, M=16, threads=0, cache_dir=None, projector=None, verbose=False, native_knn=False,
                 gt_queries=None, unit_norm=False):
    """Builds (or loads from `cache_dir`) the synthetic workload.  Returns a Dataset whose tensors
    live on `device`; graph arrays are numpy (host), as gbnns_index_create wants them."""
    recipe = dict(n=n, nq=nq, d=d, d_low=d_low, d_hidden=d_hidden, seed=seed, intrinsic=intrinsic,
                  n_clusters=n_clusters, cluster_scale=cluster_scale, sigma=sigma, knn_k=knn_k, M=M)
    if gt_queries is not None and gt_queries < nq:
        recipe["gt_queries"] = gt_queries  # exact ground truth for the first gt_queries queries only (huge batches)
    if unit_norm:
        recipe["unit_norm"] = True  # vectors scaled to unit length (the reference does this to GloVe): L2 and dot orders agree
    if native_knn:
        # kNN lists and ground truth from gbnns_exact_knn (the reference's distance arithmetic) instead of torch's
        # formula-based top-k: what large n needs (torch's n x chunk distance matrices do not scale to 10^7)
        recipe["knn"] = "gbnns_exact_knn"
    key = hashlib.sha1(repr(sorted(recipe.items())).encode()).hexdigest()[:16]
    dev = torch.device(device)
    path = os.path.join(cache_dir, f"gbnns_synth_{key}.pt") if cache_dir else None
    if path and os.path.exists(path):
        blob = torch.load(path, map_location=dev)
        blob["graph_off"] = blob["graph_off"].cpu().numpy().astype(np.uint64)
        blob["graph_nbr"] = blob["graph_nbr"].cpu().numpy().astype(np.uint32)
        blob["net"] = tuple(blob["net"])
        return Dataset(recipe=recipe, n=n, nq=nq, d=d, d_low=d_low, d_hidden=d_hidden, timings={},
                       **blob)

    t0 = time.time()
    timings = {}
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    m = intrinsic
    A = _orthonormal_rows(gen, d_low, d, dev)       # rows 0..m-1 span the data, the rest is B
    centers = torch.randn(n_clusters, m, generator=gen, device=dev)
    centers = centers / centers.norm(dim=1, keepdim=True)

    def sample(count):
        j = torch.randint(0, n_clusters, (count,), generator=gen, device=dev)
        z = centers[j] + cluster_scale * torch.randn(count, m, generator=gen, device=dev)
        z = z / z.norm(dim=1, keepdim=True)
        x = z @ A[:m] + sigma * torch.randn(count, d, generator=gen, device=dev)
        if unit_norm:
            x = x / x.norm(dim=1, keepdim=True)
        return x.contiguous()

    base = sample(n)
    queries = sample(nq)
    net = make_net(A, d_hidden)
    timings["vectors_s"] = time.time() - t0

    t1 = time.time()
    db_low = (projector or project)(net, base)
    timings["project_base_s"] = time.time() - t1

    t1 = time.time()
    if native_knn:
        parts = []
        step = 1 << 20
        for s0 in range(0, n, step):
            parts.append(binding.exact_knn(db_low, db_low[s0:s0 + step], knn_k, self_offset=s0))
            if verbose:
                print("synth: kNN rows", min(n, s0 + step), "of", n, "at %.1fs" % (time.time() - t1), file=sys.stderr, flush=True)
        knn = torch.cat(parts)
        del parts
    else:
        knn = knn_exact(db_low, knn_k)
    timings["knn_s"] = time.time() - t1

    t1 = time.time()
    knn_h = knn.cpu().numpy().astype(np.uint32)
    koff = np.arange(n + 1, dtype=np.uint64) * np.uint64(knn_k)
    goff, gnbr = binding.build_graph_gd(koff, knn_h.reshape(-1), db_low.cpu().numpy(), M,
                                        reverse=True, threads=threads)
    del knn, knn_h
    timings["gd_s"] = time.time() - t1

    t1 = time.time()
    gq = queries if gt_queries is None or gt_queries >= nq else queries[:gt_queries]
    if native_knn:
        gt2 = binding.exact_knn(base, gq, 2).to(torch.int64)
    else:
        gt2 = ground_truth(base, gq, 2)
    timings["gt_s"] = time.time() - t1
    timings["total_s"] = time.time() - t0
    if verbose:
        print("synth:", {k: round(v, 2) for k, v in timings.items()}, file=sys.stderr, flush=True)

    ds = Dataset(recipe=recipe, n=n, nq=nq, d=d, d_low=d_low, d_hidden=d_hidden, base=base,
                 queries=queries, net=net, db_low=db_low, graph_off=goff, graph_nbr=gnbr,
                 gt=gt2[:, 0].contiguous(), gt2=gt2, timings=timings)
    if path:
        os.makedirs(cache_dir, exist_ok=True)
        tmp = path + f".tmp{os.getpid()}"
        torch.save(dict(base=base, queries=queries, net=list(net), db_low=db_low,
                        graph_off=torch.from_numpy(goff.astype(np.int64)),
                        graph_nbr=torch.from_numpy(gnbr.astype(np.int64)),
                        gt=ds.gt, gt2=gt2), tmp)
        os.replace(tmp, path)
    return ds
