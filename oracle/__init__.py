"""ctypes bindings for the CPU checker libraries (TEST INFRASTRUCTURE ONLY).

`oracle.Oracle()`  -> liboracle.so, the restatement in gbnns_oracle.cpp.
`oracle.Ref()`     -> _ref/libgbnns_ref.so, the compiled reference (present only where it was
                      built from /root/reference; `oracle.have_ref()` tells).

Both expose the same numpy-level API (l2, negdot, project, walk, rerank via search_batch,
hnswlike_gd) so tests can run one body against either.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this package; the product package
(gbnns_dim_red_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "liboracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libgbnns_ref.so")
REF_FAST_SO = os.path.join(_HERE, "_ref", "libgbnns_ref_fast.so")  # the README's -Ofast -march=native build (timing only)

L2, NEG_DOT = 0, 1
MODE_NET, MODE_LOWQ, MODE_PLAIN = 0, 1, 2

_f32p = C.POINTER(C.c_float)
_u32p = C.POINTER(C.c_uint32)
_i32p = C.POINTER(C.c_int32)
_u64p = C.POINTER(C.c_uint64)


def build(force=False):
    """Compile liboracle.so (and _ref when the reference sources are present)."""
    if force and os.path.exists(ORACLE_SO):
        os.remove(ORACLE_SO)
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])


def have_ref():
    return os.path.exists(REF_SO)


def have_ref_fast():
    return os.path.exists(REF_FAST_SO)


def _p(a, typ):
    if a is None:
        return None
    return a.ctypes.data_as(typ)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


class _Base:
    prefix = None

    def _fn(self, name, restype, argtypes):
        f = getattr(self.lib, self.prefix + name)
        f.restype = restype
        f.argtypes = argtypes
        return f

    def l2(self, a, b):
        a, b = _f32(a), _f32(b)
        return np.float32(self._l2(_p(a, _f32p), _p(b, _f32p), a.size))

    def negdot(self, a, b):
        a, b = _f32(a), _f32(b)
        return np.float32(self._negdot(_p(a, _f32p), _p(b, _f32p), a.size))

    def max_threads(self):
        return int(self._max_threads())


class Oracle(_Base):
    prefix = "gbo_"

    def __init__(self, so=None):
        so = so or os.environ.get("GBNNS_ORACLE_SO") or ORACLE_SO
        if so == ORACLE_SO and not os.path.exists(ORACLE_SO):
            build()
        self.lib = C.CDLL(so)
        self._l2 = self._fn("l2", C.c_float, [_f32p, _f32p, C.c_uint64])
        self._negdot = self._fn("negdot", C.c_float, [_f32p, _f32p, C.c_uint64])
        self._project = self._fn("project", None, [_f32p] * 5 + [C.c_uint64] + [C.c_int] * 4)
        self._walk = self._fn(
            "walk", None,
            [_f32p, C.c_uint64, _f32p, C.c_uint64, C.c_int, _u64p, _u32p, C.c_int, C.c_int, _u32p,
             C.c_int, C.c_int, _u32p, _f32p, _i32p, _i32p, _i32p, C.c_int])
        self._walk_aux = self._fn(
            "walk_aux", None,
            [_f32p, C.c_uint64, _f32p, C.c_uint64, C.c_int, _u64p, _u32p, C.c_int, C.c_int, _u32p,
             C.c_int, C.c_int, _u32p, _f32p, _i32p, _i32p, _i32p, C.c_int, _u64p, _u32p, C.c_int,
             C.c_uint32])
        self._search_aux = self._fn(
            "search_batch_aux", None,
            [C.c_int, _f32p, _f32p, C.c_uint64, _f32p, _f32p, C.c_uint64, C.c_int, C.c_int,
             C.c_int, _f32p, _f32p, _f32p, _u64p, _u32p, C.c_int, C.c_int, _u32p, C.c_int, _u32p,
             _i32p, _i32p, C.c_int, _u64p, _u32p, C.c_int, C.c_uint32])
        self._rerank = self._fn(
            "rerank", None,
            [_f32p, C.c_uint64, C.c_int, _u32p, C.c_int, _i32p, _f32p, C.c_int, _u32p, C.c_int])
        self._search = self._fn(
            "search_batch", None,
            [C.c_int, _f32p, _f32p, C.c_uint64, _f32p, _f32p, C.c_uint64, C.c_int, C.c_int,
             C.c_int, _f32p, _f32p, _f32p, _u64p, _u32p, C.c_int, C.c_int, _u32p, C.c_int, _u32p,
             _i32p, _i32p, C.c_int])
        self._gd = self._fn("hnswlike_gd", C.c_uint64,
                            [_u64p, _u32p, _f32p, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int,
                             C.c_int])
        self._gd_fetch = self._fn("hnswlike_gd_fetch", None, [_u64p, _u32p])
        self._max_threads = self._fn("max_threads", C.c_int, [])
        self._knn = self._fn("exact_knn", None,
                             [_f32p, C.c_uint64, _f32p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int64,
                              _u32p, _f32p, C.c_int])

    def exact_knn(self, base, queries, k, metric=L2, self_offset=-1, threads=0):
        """(ids, dists) [nq x k]: the k smallest (Dist(base_j, q_i), j) pairs, ascending."""
        base, queries = _f32(base), _f32(queries)
        n, d = base.shape
        nq = queries.shape[0]
        ids = np.empty((nq, k), np.uint32)
        dist = np.empty((nq, k), np.float32)
        self._knn(_p(base, _f32p), n, _p(queries, _f32p), nq, d, k, metric, self_offset, _p(ids, _u32p),
                  _p(dist, _f32p), threads)
        return ids, dist

    def project(self, net, q, threads=1):
        l1, l2, l3 = (_f32(x) for x in net)
        q = _f32(q)
        nq, d = q.shape
        dh, dlow = l1.shape[0], l3.shape[0]
        assert l1.shape == (dh, d + 1) and l2.shape == (dh, dh + 1) and l3.shape == (dlow, dh + 1)
        out = np.empty((nq, dlow), np.float32)
        self._project(_p(l1, _f32p), _p(l2, _f32p), _p(l3, _f32p), _p(q, _f32p), _p(out, _f32p),
                      nq, d, dh, dlow, threads)
        return out

    def walk(self, q, db, off, nbr, ef, k=None, entries=None, metric=L2, threads=1, aux=None,
             llf=False, hops_bound=50):
        """Returns dict(ids [nq x min(k,ef)] pop order, dists, count, hops, dist_calc).
        aux = (off, nbr) of an auxiliary graph: the reference's use_second_graph walk."""
        q, db, off, nbr = _f32(q), _f32(db), _u64(off), _u32(nbr)
        aoff, anbr = (None, None) if aux is None else (_u64(aux[0]), _u32(aux[1]))
        nq, d = q.shape
        n = db.shape[0]
        k = ef if k is None else k
        stride = min(k, ef)
        n_entries = 1
        if entries is not None:
            entries = _u32(entries).reshape(nq, -1)
            n_entries = entries.shape[1]
        ids = np.empty((nq, stride), np.uint32)
        dists = np.empty((nq, stride), np.float32)
        count = np.empty(nq, np.int32)
        hops = np.empty(nq, np.int32)
        dc = np.empty(nq, np.int32)
        self._walk_aux(_p(q, _f32p), nq, _p(db, _f32p), n, d, _p(off, _u64p), _p(nbr, _u32p), ef, k,
                       _p(entries, _u32p), n_entries, metric, _p(ids, _u32p), _p(dists, _f32p),
                       _p(count, _i32p), _p(hops, _i32p), _p(dc, _i32p), threads, _p(aoff, _u64p),
                       _p(anbr, _u32p), int(llf), hops_bound)
        return dict(ids=ids, dists=dists, count=count, hops=hops, dist_calc=dc)

    def rerank(self, q, cand, count, db, metric=L2, threads=1):
        q, db, cand = _f32(q), _f32(db), _u32(cand)
        nq, d = q.shape
        count = None if count is None else np.ascontiguousarray(count, np.int32)
        out = np.empty(nq, np.uint32)
        self._rerank(_p(q, _f32p), nq, d, _p(cand, _u32p), cand.shape[1], _p(count, _i32p),
                     _p(db, _f32p), metric, _p(out, _u32p), threads)
        return out

    def search_batch(self, mode, queries, db, off, nbr, ef, k=1, db_low=None, net=None,
                     q_low=None, entries=None, metric=L2, threads=1, aux=None, llf=False,
                     hops_bound=50):
        queries, db, off, nbr = _f32(queries), _f32(db), _u64(off), _u32(nbr)
        aoff, anbr = (None, None) if aux is None else (_u64(aux[0]), _u32(aux[1]))
        nq, d = queries.shape
        n = db.shape[0]
        dlow = dh = 0
        l1 = l2 = l3 = None
        if mode != MODE_PLAIN:
            db_low = _f32(db_low)
            dlow = db_low.shape[1]
        if mode == MODE_NET:
            l1, l2, l3 = (_f32(x) for x in net)
            dh = l1.shape[0]
        if mode == MODE_LOWQ:
            q_low = _f32(q_low)
        entries = None if entries is None else _u32(entries)
        ids = np.empty(nq, np.uint32)
        hops = np.empty(nq, np.int32)
        dc = np.empty(nq, np.int32)
        self._search_aux(mode, _p(queries, _f32p), _p(q_low, _f32p), nq, _p(db, _f32p),
                         _p(db_low, _f32p), n, d, dlow, dh, _p(l1, _f32p), _p(l2, _f32p),
                         _p(l3, _f32p), _p(off, _u64p), _p(nbr, _u32p), ef, k, _p(entries, _u32p),
                         metric, _p(ids, _u32p), _p(hops, _i32p), _p(dc, _i32p), threads,
                         _p(aoff, _u64p), _p(anbr, _u32p), int(llf), hops_bound)
        return dict(ids=ids, hops=hops, dist_calc=dc)

    def hnswlike_gd(self, koff, knbr, ds, M, metric=L2, reverse=True, threads=0):
        koff, knbr, ds = _u64(koff), _u32(knbr), _f32(ds)
        n, d = ds.shape
        total = self._gd(_p(koff, _u64p), _p(knbr, _u32p), _p(ds, _f32p), M, n, d, metric,
                         int(reverse), threads)
        off = np.empty(n + 1, np.uint64)
        nbr = np.empty(max(int(total), 1), np.uint32)
        self._gd_fetch(_p(off, _u64p), _p(nbr, _u32p))
        return off, nbr[:int(total)]


class Ref(_Base):
    """The compiled reference (oracle/_ref).  Same numpy API as Oracle."""
    prefix = "ref_"

    def __init__(self, so=None):
        so = so or os.environ.get("GBNNS_REF_SO") or REF_SO
        if not os.path.exists(so):
            raise FileNotFoundError(so)
        self.lib = C.CDLL(so)
        self._l2 = self._fn("l2", C.c_float, [_f32p, _f32p, C.c_uint64])
        self._negdot = self._fn("negdot", C.c_float, [_f32p, _f32p, C.c_uint64])
        self._graph_create = self._fn("graph_create", C.c_void_p, [_u64p, _u32p, C.c_uint64])
        self._graph_destroy = self._fn("graph_destroy", None, [C.c_void_p])
        self._project = self._fn("project", None, [_f32p] * 5 + [C.c_uint64] + [C.c_int] * 3)
        self._walk = self._fn(
            "walk", None,
            [_f32p, C.c_uint64, _f32p, C.c_uint64, C.c_int, C.c_void_p, C.c_int, C.c_int, _u32p,
             C.c_int, C.c_int, _u32p, _f32p, _i32p, _i32p, _i32p, C.c_int])
        self._search = self._fn(
            "search_batch", None,
            [C.c_int, _f32p, _f32p, C.c_uint64, _f32p, _f32p, C.c_uint64, C.c_int, C.c_int,
             C.c_int, _f32p, _f32p, _f32p, C.c_void_p, C.c_int, C.c_int, _u32p, C.c_int, _u32p,
             _i32p, _i32p, C.c_int])
        self._walk_aux = self._fn(
            "walk_aux", None,
            [_f32p, C.c_uint64, _f32p, C.c_uint64, C.c_int, C.c_void_p, C.c_int, C.c_int, _u32p,
             C.c_int, C.c_int, _u32p, _f32p, _i32p, _i32p, _i32p, C.c_int, C.c_void_p, C.c_int,
             C.c_uint32])
        self._search_aux = self._fn(
            "search_batch_aux", None,
            [C.c_int, _f32p, _f32p, C.c_uint64, _f32p, _f32p, C.c_uint64, C.c_int, C.c_int,
             C.c_int, _f32p, _f32p, _f32p, C.c_void_p, C.c_int, C.c_int, _u32p, C.c_int, _u32p,
             _i32p, _i32p, C.c_int, C.c_void_p, C.c_int, C.c_uint32])
        self._prepare = self._fn("prepare_db_cache", None, [_f32p, C.c_uint64, C.c_int])
        self._gd = self._fn("hnswlike_gd", C.c_uint64,
                            [_u64p, _u32p, _f32p, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int,
                             C.c_int])
        self._gd_fetch = self._fn("hnswlike_gd_fetch", None, [_u64p, _u32p])
        self._net_tests = self._fn(
            "perform_real_net_tests", None,
            [C.c_int] * 5 + [_i32p, C.c_int, C.c_void_p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p,
                             C.c_int, _u32p, C.c_char_p, C.c_char_p, C.c_int, C.c_int])
        self._real_tests = self._fn(
            "perform_real_tests", None,
            [C.c_int] * 5 + [_i32p, C.c_int, C.c_void_p, _f32p, _f32p, _f32p, _f32p, _u32p,
                             C.c_char_p, C.c_char_p, C.c_int, C.c_int])
        self._real_tests_aux = self._fn(
            "perform_real_tests_aux", None,
            [C.c_int] * 5 + [_i32p, C.c_int, C.c_void_p, _f32p, _f32p, _f32p, _f32p, _u32p,
                             C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_uint32])
        self._max_threads = self._fn("max_threads", C.c_int, [])
        self._truth = self._fn("get_truth", None, [_f32p, C.c_uint64, _f32p, C.c_uint64, C.c_int, C.c_int,
                                                   _u32p, C.c_int])
        self._graphs = {}

    def get_truth(self, base, queries, metric=L2, threads=4):
        """The reference's getTruth: id of the nearest base row of every query."""
        base, queries = _f32(base), _f32(queries)
        out = np.empty(queries.shape[0], np.uint32)
        self._truth(_p(base, _f32p), base.shape[0], _p(queries, _f32p), queries.shape[0], base.shape[1],
                    metric, _p(out, _u32p), threads)
        return out

    def _graph(self, off, nbr):
        off, nbr = _u64(off), _u32(nbr)
        key = (off.ctypes.data, nbr.ctypes.data, off.size)
        if key not in self._graphs:
            h = self._graph_create(_p(off, _u64p), _p(nbr, _u32p), off.size - 1)
            self._graphs[key] = (h, off, nbr)  # keep arrays alive so the key stays unique
        return self._graphs[key][0]

    def close(self):
        for h, _, _ in self._graphs.values():
            self._graph_destroy(h)
        self._graphs = {}

    def project(self, net, q, threads=1):
        l1, l2, l3 = (_f32(x) for x in net)
        q = _f32(q)
        nq, d = q.shape
        dh, dlow = l1.shape[0], l3.shape[0]
        out = np.empty((nq, dlow), np.float32)
        self._project(_p(l1, _f32p), _p(l2, _f32p), _p(l3, _f32p), _p(q, _f32p), _p(out, _f32p),
                      nq, d, dh, dlow)
        return out

    def walk(self, q, db, off, nbr, ef, k=None, entries=None, metric=L2, threads=1, aux=None,
             llf=False, hops_bound=50):
        q, db = _f32(q), _f32(db)
        g = self._graph(off, nbr)
        ga = None if aux is None else self._graph(aux[0], aux[1])
        nq, d = q.shape
        n = db.shape[0]
        k = ef if k is None else k
        stride = min(k, ef)
        n_entries = 1
        if entries is not None:
            entries = _u32(entries).reshape(nq, -1)
            n_entries = entries.shape[1]
        ids = np.empty((nq, stride), np.uint32)
        dists = np.empty((nq, stride), np.float32)
        count = np.empty(nq, np.int32)
        hops = np.empty(nq, np.int32)
        dc = np.empty(nq, np.int32)
        self._walk_aux(_p(q, _f32p), nq, _p(db, _f32p), n, d, g, ef, k, _p(entries, _u32p), n_entries,
                       metric, _p(ids, _u32p), _p(dists, _f32p), _p(count, _i32p), _p(hops, _i32p),
                       _p(dc, _i32p), threads, ga, int(llf), hops_bound)
        return dict(ids=ids, dists=dists, count=count, hops=hops, dist_calc=dc)

    def prepare(self, db):
        db = _f32(db)
        self._prepare(_p(db, _f32p), db.shape[0], db.shape[1])

    def search_batch(self, mode, queries, db, off, nbr, ef, k=1, db_low=None, net=None,
                     q_low=None, entries=None, metric=L2, threads=1, aux=None, llf=False,
                     hops_bound=50):
        queries, db = _f32(queries), _f32(db)
        g = self._graph(off, nbr)
        ga = None if aux is None else self._graph(aux[0], aux[1])
        nq, d = queries.shape
        n = db.shape[0]
        dlow = dh = 0
        l1 = l2 = l3 = None
        if mode != MODE_PLAIN:
            db_low = _f32(db_low)
            dlow = db_low.shape[1]
        if mode == MODE_NET:
            l1, l2, l3 = (_f32(x) for x in net)
            dh = l1.shape[0]
        if mode == MODE_LOWQ:
            q_low = _f32(q_low)
        entries = None if entries is None else _u32(entries)
        ids = np.empty(nq, np.uint32)
        hops = np.empty(nq, np.int32)
        dc = np.empty(nq, np.int32)
        self._search_aux(mode, _p(queries, _f32p), _p(q_low, _f32p), nq, _p(db, _f32p),
                     _p(db_low, _f32p), n, d, dlow, dh, _p(l1, _f32p), _p(l2, _f32p),
                     _p(l3, _f32p), g, ef, k, _p(entries, _u32p), metric, _p(ids, _u32p),
                     _p(hops, _i32p), _p(dc, _i32p), threads, ga, int(llf), hops_bound)
        return dict(ids=ids, hops=hops, dist_calc=dc)

    def hnswlike_gd(self, koff, knbr, ds, M, metric=L2, reverse=True, threads=0):
        koff, knbr, ds = _u64(koff), _u32(knbr), _f32(ds)
        n, d = ds.shape
        total = self._gd(_p(koff, _u64p), _p(knbr, _u32p), _p(ds, _f32p), M, n, d, metric,
                         int(reverse), threads)
        off = np.empty(n + 1, np.uint64)
        nbr = np.empty(max(int(total), 1), np.uint32)
        self._gd_fetch(_p(off, _u64p), _p(nbr, _u32p))
        return off, nbr[:int(total)]

    # ---- graph utilities around the search path (golden vectors for the drop-in's host helpers) ----
    def _fetch(self, n, total):
        off = np.empty(n + 1, np.uint64)
        nbr = np.empty(max(int(total), 1), np.uint32)
        self._gd_fetch(_p(off, _u64p), _p(nbr, _u32p))
        return off, nbr[:int(total)]

    def _util(self, name, argtypes):
        f = getattr(self.lib, "ref_" + name)
        f.restype = C.c_uint64
        f.argtypes = argtypes
        return f

    def hnswlike_gd_const(self, koff, knbr, ds, M, metric=L2, reverse=True):
        koff, knbr, ds = _u64(koff), _u32(knbr), _f32(ds)
        n, d = ds.shape
        f = self._util("hnswlike_gd_const", [_u64p, _u32p, _f32p, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int])
        return self._fetch(n, f(_p(koff, _u64p), _p(knbr, _u32p), _p(ds, _f32p), M, n, d, metric, int(reverse)))

    def cut_knn_by_k(self, koff, knbr, ds, k, metric=L2):
        koff, knbr, ds = _u64(koff), _u32(knbr), _f32(ds)
        n, d = ds.shape
        f = self._util("cut_knn_by_k", [_u64p, _u32p, _f32p, C.c_int, C.c_uint64, C.c_int, C.c_int])
        return self._fetch(n, f(_p(koff, _u64p), _p(knbr, _u32p), _p(ds, _f32p), k, n, d, metric))

    def cut_knn_by_threshold(self, koff, knbr, ds, thr, metric=L2):
        koff, knbr, ds = _u64(koff), _u32(knbr), _f32(ds)
        n, d = ds.shape
        f = self._util("cut_knn_by_threshold", [_u64p, _u32p, _f32p, C.c_float, C.c_uint64, C.c_int, C.c_int])
        return self._fetch(n, f(_p(koff, _u64p), _p(knbr, _u32p), _p(ds, _f32p), thr, n, d, metric))

    def merge_graph(self, aoff, anbr, boff, bnbr):
        aoff, anbr, boff, bnbr = _u64(aoff), _u32(anbr), _u64(boff), _u32(bnbr)
        n = len(aoff) - 1
        f = self._util("merge_graph", [_u64p, _u32p, _u64p, _u32p, C.c_uint64])
        return self._fetch(n, f(_p(aoff, _u64p), _p(anbr, _u32p), _p(boff, _u64p), _p(bnbr, _u32p), n))

    def fill_const_degree(self, aoff, anbr, boff, bnbr, degree):
        aoff, anbr, boff, bnbr = _u64(aoff), _u32(anbr), _u64(boff), _u32(bnbr)
        n = len(aoff) - 1
        f = self._util("fill_const_degree", [_u64p, _u32p, _u64p, _u32p, C.c_uint64, C.c_int])
        return self._fetch(n, f(_p(aoff, _u64p), _p(anbr, _u32p), _p(boff, _u64p), _p(bnbr, _u32p), n, degree))

    def make_step(self, db, query, nb, ef, top, cand, visited, metric=L2):
        """The reference's makeStep once: top / cand = (keys f32, ids u32) heap contents, visited = marked ids.
        Returns dict(dist_calc, found, marked, top=(keys, ids) in pop order, cand=(keys, ids) in pop order)."""
        db, query = _f32(db), _f32(query)
        nb, visited = _u32(nb), _u32(visited)
        tk, ti, ck, ci = _f32(top[0]), _u32(top[1]), _f32(cand[0]), _u32(cand[1])
        cap = len(tk) + len(ck) + len(nb) + 1
        info = np.zeros(5, np.int32)
        otk, oti = np.zeros(cap, np.float32), np.zeros(cap, np.uint32)
        ock, oci = np.zeros(cap, np.float32), np.zeros(cap, np.uint32)
        f = self.lib.ref_make_step
        f.restype = None
        f.argtypes = [_f32p, C.c_uint64, C.c_int, _f32p, _u32p, C.c_int, C.c_int, _f32p, _u32p, C.c_int, _f32p, _u32p,
                      C.c_int, _u32p, C.c_int, C.c_int, _i32p, _f32p, _u32p, _f32p, _u32p]
        f(_p(db, _f32p), db.shape[0], db.shape[1], _p(query, _f32p), _p(nb, _u32p), len(nb), ef, _p(tk, _f32p),
          _p(ti, _u32p), len(tk), _p(ck, _f32p), _p(ci, _u32p), len(ck), _p(visited, _u32p), len(visited), metric,
          _p(info, _i32p), _p(otk, _f32p), _p(oti, _u32p), _p(ock, _f32p), _p(oci, _u32p))
        nt, nc = int(info[2]), int(info[3])
        return dict(dist_calc=int(info[0]), found=bool(info[1]), marked=int(info[4]), top=(otk[:nt], oti[:nt]),
                    cand=(ock[:nc], oci[:nc]))

    def kl_build(self, which, l, ds, sqrt_n, seed, metric=L2):
        ds = _f32(ds)
        n, d = ds.shape
        f = self._util("kl_build", [C.c_int, C.c_int, _f32p, C.c_uint64, C.c_int, C.c_uint64, C.c_uint32, C.c_int])
        return self._fetch(n, f(which, l, _p(ds, _f32p), n, d, sqrt_n, seed, metric))

    def create_uniform_data(self, n, d, seed):
        out = np.empty((n, d), np.float32)
        f = self.lib.ref_create_uniform_data
        f.restype = None
        f.argtypes = [C.c_int, C.c_int, C.c_uint32, _f32p]
        f(n, d, seed, _p(out, _f32p))
        return out

    def perform_real_net_tests(self, db, queries, db_low, net, off, nbr, truth, efs, out_path,
                               graph_name="hnsw_new_ar", number_exper=1, threads=1):
        db, queries, db_low = _f32(db), _f32(queries), _f32(db_low)
        l1, l2, l3 = (_f32(x) for x in net)
        truth = _u32(truth)
        efs = np.ascontiguousarray(efs, np.int32)
        g = self._graph(off, nbr)
        n, d = db.shape
        self._net_tests(n, d, db_low.shape[1], queries.shape[0], truth.shape[1], _p(efs, _i32p),
                        efs.size, g, _p(db, _f32p), _p(queries, _f32p), _p(db_low, _f32p),
                        _p(l1, _f32p), _p(l2, _f32p), _p(l3, _f32p), l1.shape[0],
                        _p(truth, _u32p), out_path.encode(), graph_name.encode(), number_exper,
                        threads)

    def perform_real_tests(self, db, queries, db_low, queries_low, off, nbr, truth, efs, out_path,
                           graph_name="hnsw", number_exper=1, threads=1, aux=None, llf=False, seed=1):
        db, queries, db_low, queries_low = _f32(db), _f32(queries), _f32(db_low), _f32(queries_low)
        truth = _u32(truth)
        efs = np.ascontiguousarray(efs, np.int32)
        g = self._graph(off, nbr)
        n, d = db.shape
        ga = None if aux is None else self._graph(aux[0], aux[1])
        self._real_tests_aux(n, d, db_low.shape[1], queries.shape[0], truth.shape[1], _p(efs, _i32p),
                             efs.size, g, _p(db, _f32p), _p(queries, _f32p), _p(db_low, _f32p),
                             _p(queries_low, _f32p), _p(truth, _u32p), out_path.encode(),
                             graph_name.encode(), number_exper, threads, ga, int(llf), seed)
