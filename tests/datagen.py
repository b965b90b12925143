"""Deterministic, bit-portable test inputs.

Everything here is built from numpy's PCG64 integer stream and exact integer->float32
conversions (no libm, no BLAS), so the same seed gives the same bytes on every host; the golden
fixtures store a sha256 of the regenerated inputs to prove it.
"""
import hashlib

import numpy as np


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _ints(rng, shape, lo, hi):
    return rng.integers(lo, hi, size=shape, dtype=np.int64)


def clustered(rng, n, d, n_centers=16, spread=40, scale=256.0):
    """n points around integer-grid centres (overlapping clusters, so kNN graphs stay connected);
    every coordinate is an exact multiple of 1/scale."""
    centers = _ints(rng, (n_centers, d), -12, 13)
    assign = _ints(rng, (n,), 0, n_centers)
    noise = _ints(rng, (n, d), -spread, spread + 1)
    return ((centers[assign] * 4 + noise).astype(np.float32) / np.float32(scale)).astype(np.float32)


def lattice(rng, n, d, levels=3, dup=3):
    """Small-integer coordinates with every vector repeated `dup` times: distances are small
    exact integers, so equal-distance ties (and exact duplicates) are everywhere."""
    m = (n + dup - 1) // dup
    base = _ints(rng, (m, d), 0, levels).astype(np.float32)
    out = np.repeat(base, dup, axis=0)[:n]
    perm = np.argsort(_ints(rng, (n,), 0, 1 << 40), kind="stable")
    return np.ascontiguousarray(out[perm])


def net_layers(rng, d, dh, dlow):
    """Three [d_out x (d_in+1)] layers (weights | bias), dyadic-rational entries."""
    def layer(dout, din, shift):
        w = _ints(rng, (dout, din + 1), -128, 129).astype(np.float32)
        return (w / np.float32(1 << shift)).astype(np.float32)
    # shifts keep activations O(1) for inputs of magnitude ~1
    s1 = 7 + max(0, int(np.log2(max(d, 2))) // 2)
    s2 = 7 + max(0, int(np.log2(max(dh, 2))) // 2)
    return layer(dh, d, s1), layer(dh, dh, s2), layer(dlow, dh, s2)


def knn_bruteforce(x, k, block=512):
    """Exact-enough kNN lists (float64 distances, ties by id); excludes self.  Generator-side
    helper only: its output is an *input* (and is committed where bit-portability matters)."""
    x64 = x.astype(np.float64)
    n = x.shape[0]
    sq = (x64 * x64).sum(1)
    out = np.empty((n, k), np.uint32)
    for s in range(0, n, block):
        e = min(n, s + block)
        dmat = sq[s:e, None] + sq[None, :] - 2.0 * (x64[s:e] @ x64.T)
        dmat[np.arange(e - s), np.arange(s, e)] = np.inf
        idx = np.argsort(dmat, axis=1, kind="stable")[:, :k]
        out[s:e] = idx.astype(np.uint32)
    return out


def lists_to_csr(lists):
    off = np.zeros(len(lists) + 1, np.uint64)
    off[1:] = np.cumsum([len(l) for l in lists])
    nbr = np.concatenate([np.asarray(l, np.uint32) for l in lists]) if len(lists) else \
        np.zeros(0, np.uint32)
    return off, np.ascontiguousarray(nbr, np.uint32)


def dense_to_csr(knn):
    n, k = knn.shape
    off = (np.arange(n + 1, dtype=np.uint64) * np.uint64(k)).astype(np.uint64)
    return off, np.ascontiguousarray(knn.reshape(-1), np.uint32)


def random_graph(rng, n, deg_lo, deg_hi):
    """Ragged random adjacency (no self loops, no duplicate neighbours inside a list)."""
    lists = []
    for i in range(n):
        k = int(_ints(rng, (), deg_lo, deg_hi + 1))
        c = np.unique(_ints(rng, (k * 2 + 2,), 0, n))
        c = c[c != i]
        rng.shuffle(c)
        lists.append(c[:k].astype(np.uint32))
    return lists_to_csr(lists)


KAT_DIMS = list(range(0, 41)) + [45, 96, 128, 200, 257, 960]


def kat_pairs():
    """Vector pairs for the scalar distance known-answer tests (4 per dimension in KAT_DIMS)."""
    rng = np.random.Generator(np.random.PCG64(4242))
    pairs = []
    for d in KAT_DIMS:
        for _ in range(4):
            a = rng.integers(-2**20, 2**20, size=d).astype(np.float32) / np.float32(2**18)
            b = rng.integers(-2**20, 2**20, size=d).astype(np.float32) / np.float32(2**18)
            pairs.append((a.astype(np.float32), b.astype(np.float32)))
    return pairs


class Case:
    """One seeded configuration: base, queries, net (all bit-portable)."""

    def __init__(self, name, seed, n, nq, d, dlow, dh, kind="clustered", metric=0):
        self.name, self.seed, self.n, self.nq = name, seed, n, nq
        self.d, self.dlow, self.dh, self.kind, self.metric = d, dlow, dh, kind, metric
        rng = np.random.Generator(np.random.PCG64(seed))
        if kind == "lattice":
            self.base = lattice(rng, n, d)
            self.queries = lattice(rng, nq, d, dup=1)
        else:
            self.base = clustered(rng, n, d)
            self.queries = clustered(rng, nq, d)
        self.net = net_layers(rng, d, dh, dlow)
        self.rng = rng

    def input_hash(self):
        h = hashlib.sha256()
        for a in (self.base, self.queries) + tuple(self.net):
            h.update(np.ascontiguousarray(a).tobytes())
        return h.hexdigest()


# Golden configurations (tests/golden/make_golden.py); shapes follow BASELINE.json's configs at
# toy size: SIFT 128->32 w256, GIST 960->64, GloVe 200->32 (neg-dot metric), DEEP 96->32, a tail
# case with d % 8 != 0, d % 4 != 0, d_low % 4 != 0, and a tie-heavy lattice case.
GOLDEN_CASES = [
    dict(name="sift_toy", seed=101, n=4096, nq=256, d=128, dlow=32, dh=256, efs=[1, 8, 64]),
    dict(name="gist_toy", seed=102, n=2048, nq=64, d=960, dlow=64, dh=128, efs=[8, 200]),
    dict(name="glove_toy", seed=103, n=2048, nq=128, d=200, dlow=32, dh=64, efs=[8, 64],
         metric=1),
    dict(name="deep_toy", seed=104, n=2048, nq=128, d=96, dlow=32, dh=64, efs=[1, 40]),
    dict(name="tail_toy", seed=105, n=1024, nq=64, d=45, dlow=14, dh=27, efs=[1, 8, 33]),
    dict(name="ties_toy", seed=106, n=3072, nq=128, d=16, dlow=8, dh=16, efs=[1, 2, 8, 64],
         kind="lattice"),
]
