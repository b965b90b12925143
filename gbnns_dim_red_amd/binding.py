"""ctypes binding of include/gbnns.h.

Host buffers are numpy arrays; device buffers are torch CUDA(ROCm) tensors (torch is used only
as the owner of device memory and streams -- no torch op is on the search path).
"""
import collections
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "lib", "libgbnns_hip.so")

METRIC_L2, METRIC_NEG_DOT = 0, 1
MEM_HOST, MEM_DEVICE = 0, 1
MODE_NET, MODE_LOWQ, MODE_PLAIN = 0, 1, 2
FLAG_NO_FUSED_RERANK = 2
FLAG_AUX_GRAPH = 4
FLAG_LLF = 8
FLAG_WIDE_INDEX = 16
FLAG_BITMAP_PASS = 32
FLAG_SERIAL = 64
FLAG_DEFER_JOIN = 128
FLAG_MFMA_PROJECTION = 256

# every symbol include/gbnns.h declares (tests check the library exports all of them)
SYMBOLS = [
    "gbnns_index_create", "gbnns_index_destroy", "gbnns_index_set_aux_graph", "gbnns_search_ex", "gbnns_search_batch", "gbnns_index_join", "gbnns_index_wait", "gbnns_host_pin", "gbnns_host_unpin",
    "gbnns_project", "gbnns_rerank", "gbnns_debug_knob", "gbnns_index_knob", "gbnns_index_knob_get", "gbnns_profile_enable", "gbnns_profile_read", "gbnns_build_graph_gd", "gbnns_build_graph_gd_device",
    "gbnns_free", "gbnns_exact_knn", "gbnns_device_count", "gbnns_version", "gbnns_last_error",
    "gbnns_index_n", "gbnns_index_d", "gbnns_index_d_low", "gbnns_index_device",
    "gbnns_multi_create", "gbnns_multi_destroy", "gbnns_multi_size", "gbnns_multi_replica", "gbnns_multi_device_of",
    "gbnns_multi_stream", "gbnns_multi_set_aux_graph", "gbnns_shard_bounds", "gbnns_multi_search_ex",
    "gbnns_multi_search_device", "gbnns_multi_synchronize", "gbnns_multi_last_error",
    "gbnns_multi_rccl_single_rank", "gbnns_multi_rccl_version",
]


class GbnnsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"gbnns status {code}: {msg}")
        self.code = code


class _IndexDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("device", C.c_int32), ("metric", C.c_int32),
        ("mem_kind", C.c_int32), ("n", C.c_uint64), ("d", C.c_uint32), ("d_low", C.c_uint32),
        ("d_hidden", C.c_uint32), ("reserved0", C.c_uint32), ("db", C.c_void_p),
        ("db_low", C.c_void_p), ("graph_offsets", C.c_void_p), ("graph_nbrs", C.c_void_p),
        ("net_l1", C.c_void_p), ("net_l2", C.c_void_p), ("net_l3", C.c_void_p),
    ]


class _SearchArgs(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("mode", C.c_int32), ("ef", C.c_int32), ("k", C.c_int32),
        ("mem_kind", C.c_int32), ("hash_capacity", C.c_int32), ("n_q", C.c_uint64),
        ("queries", C.c_void_p), ("queries_low", C.c_void_p), ("entry_ids", C.c_void_p),
        ("out_ids", C.c_void_p), ("out_hops", C.c_void_p), ("out_dist_calc", C.c_void_p),
        ("out_cand", C.c_void_p), ("out_cand_dist", C.c_void_p), ("out_q_low", C.c_void_p),
        ("out_edges", C.c_void_p), ("stream", C.c_void_p), ("flags", C.c_uint32),
        ("hops_bound", C.c_uint32), ("n_entries", C.c_uint32), ("defer_depth", C.c_uint32),
    ]


class Profile(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("calls", C.c_uint32), ("project_ms", C.c_double),
        ("walk_ms", C.c_double), ("walk_general_ms", C.c_double), ("rerank_ms", C.c_double),
        ("total_ms", C.c_double), ("queries", C.c_uint64), ("general_queries", C.c_uint64),
        ("walk_kernel", C.c_char * 96), ("project_kernel", C.c_char * 32),
    ]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k != "struct_size"}
        d["walk_kernel"] = d["walk_kernel"].decode("ascii", "replace")
        d["project_kernel"] = d["project_kernel"].decode("ascii", "replace")
        return d


_lib = None


def lib_path():
    return _LIB_PATH


def load_library():
    """Load libgbnns_hip.so.  Fails loudly when it has not been built: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise ImportError(
            f"{_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc, gfx950).  gbnns_dim_red_amd has no CPU fallback.")
    lib = C.CDLL(_LIB_PATH)
    lib.gbnns_last_error.restype = C.c_char_p
    lib.gbnns_index_create.argtypes = [C.POINTER(_IndexDesc), C.POINTER(C.c_void_p)]
    lib.gbnns_index_destroy.argtypes = [C.c_void_p]
    lib.gbnns_index_set_aux_graph.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.gbnns_search_ex.argtypes = [C.c_void_p, C.POINTER(_SearchArgs)]
    lib.gbnns_index_join.argtypes = [C.c_void_p]
    lib.gbnns_index_wait.argtypes = [C.c_void_p, C.c_uint32]
    lib.gbnns_host_pin.argtypes = [C.c_void_p, C.c_size_t]
    lib.gbnns_host_unpin.argtypes = [C.c_void_p]
    lib.gbnns_search_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.gbnns_project.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int,
                                  C.c_void_p]
    lib.gbnns_rerank.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32,
                                 C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    if hasattr(lib, "gbnns_debug_knob"):  # (absent from the rounds 1-3 libraries that tools/ab4.sh may put in place of this one)
        lib.gbnns_debug_knob.argtypes = [C.c_char_p, C.c_int]
    if hasattr(lib, "gbnns_index_knob"):
        lib.gbnns_index_knob.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        lib.gbnns_index_knob_get.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]
    lib.gbnns_profile_enable.argtypes = [C.c_void_p, C.c_int]
    lib.gbnns_profile_read.argtypes = [C.c_void_p, C.POINTER(Profile), C.c_int]
    lib.gbnns_build_graph_gd.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                         C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    lib.gbnns_free.argtypes = [C.c_void_p]
    lib.gbnns_build_graph_gd_device.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                                C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int,
                                                C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                                C.POINTER(C.c_uint64)]
    lib.gbnns_exact_knn.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint32,
                                    C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.gbnns_multi_last_error.restype = C.c_char_p
    lib.gbnns_multi_create.argtypes = [C.POINTER(_IndexDesc), C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]
    lib.gbnns_multi_destroy.argtypes = [C.c_void_p]
    lib.gbnns_multi_size.argtypes = [C.c_void_p]
    lib.gbnns_multi_replica.argtypes = [C.c_void_p, C.c_int32]
    lib.gbnns_multi_replica.restype = C.c_void_p
    lib.gbnns_multi_stream.argtypes = [C.c_void_p, C.c_int32]
    lib.gbnns_multi_stream.restype = C.c_void_p
    lib.gbnns_multi_device_of.argtypes = [C.c_void_p, C.c_int32]
    lib.gbnns_multi_set_aux_graph.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.gbnns_multi_search_ex.argtypes = [C.c_void_p, C.POINTER(_SearchArgs)]
    lib.gbnns_multi_search_device.argtypes = [C.c_void_p, C.POINTER(_SearchArgs), C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.gbnns_multi_synchronize.argtypes = [C.c_void_p]
    lib.gbnns_multi_rccl_single_rank.argtypes = [C.c_void_p, C.c_int]
    lib.gbnns_multi_rccl_version.argtypes = [C.c_void_p]
    lib.gbnns_shard_bounds.argtypes = [C.c_uint64, C.c_int32, C.c_int32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.gbnns_shard_bounds.restype = None
    lib.gbnns_index_d_low.argtypes = [C.c_void_p]
    lib.gbnns_index_d_low.restype = C.c_uint32
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise GbnnsError(rc, load_library().gbnns_last_error().decode(errors="replace"))


def version():
    return load_library().gbnns_version()


def device_count():
    return load_library().gbnns_device_count()


def _is_dev(x):
    return hasattr(x, "data_ptr")


def _ptr(x):
    if x is None:
        return None
    if _is_dev(x):
        return x.data_ptr()
    return x.ctypes.data


def _host(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _prep(x, dtype, torch_dtype_name):
    """numpy -> contiguous numpy of dtype; torch tensor -> contiguous tensor (dtype checked)."""
    if x is None:
        return None
    if _is_dev(x):
        import torch
        want = getattr(torch, torch_dtype_name)
        if x.dtype != want:
            raise TypeError(f"expected torch.{torch_dtype_name}, got {x.dtype}")
        if not x.is_cuda and not x.is_pinned():
            raise TypeError("torch buffers must be CUDA/ROCm tensors (device buffers) or pinned CPU tensors "
                            "(page-locked host buffers); pass numpy arrays for pageable host memory")
        return x.contiguous()
    return _host(x, dtype)


def build_graph_gd(knn_offsets, knn_nbrs, ds, M, metric=METRIC_L2, reverse=True, threads=0):
    """hnswlikeGD + reverse edges (support_func.h:521-575, 402-445) on host arrays -> CSR."""
    lib = load_library()
    koff, knbr, ds = _host(knn_offsets, np.uint64), _host(knn_nbrs, np.uint32), _host(ds, np.float32)
    n, d = ds.shape
    po, pn = C.c_void_p(), C.c_void_p()
    _check(lib.gbnns_build_graph_gd(koff.ctypes.data, knbr.ctypes.data, ds.ctypes.data, n, d, M,
                                    metric, int(reverse), threads, C.byref(po), C.byref(pn)))
    try:
        off = np.ctypeslib.as_array(C.cast(po, C.POINTER(C.c_uint64)), shape=(n + 1,)).copy()
        total = int(off[n])
        nbr = (np.ctypeslib.as_array(C.cast(pn, C.POINTER(C.c_uint32)), shape=(max(total, 1),))
               [:total].copy())
    finally:
        lib.gbnns_free(po)
        lib.gbnns_free(pn)
    return off, nbr


def build_graph_gd_device(knn_offsets, knn_nbrs, ds, M, metric=METRIC_L2, reverse=True, threads=0, device=0):
    """gbnns_build_graph_gd_device: per-node pruning on the device (ties finished on the host), same result as
    build_graph_gd.  Returns (offsets, nbrs, nodes_finished_on_host)."""
    lib = load_library()
    koff, knbr, ds = _host(knn_offsets, np.uint64), _host(knn_nbrs, np.uint32), _host(ds, np.float32)
    n, d = ds.shape
    po, pn, hn = C.c_void_p(), C.c_void_p(), C.c_uint64(0)
    _check(lib.gbnns_build_graph_gd_device(device, koff.ctypes.data, knbr.ctypes.data, ds.ctypes.data, n, d, M,
                                           metric, int(reverse), threads, C.byref(po), C.byref(pn), C.byref(hn)))
    try:
        off = np.ctypeslib.as_array(C.cast(po, C.POINTER(C.c_uint64)), shape=(n + 1,)).copy()
        total = int(off[n])
        nbr = (np.ctypeslib.as_array(C.cast(pn, C.POINTER(C.c_uint32)), shape=(max(total, 1),))
               [:total].copy())
    finally:
        lib.gbnns_free(po)
        lib.gbnns_free(pn)
    return off, nbr, int(hn.value)


def exact_knn(base, queries, k, metric=METRIC_L2, self_offset=-1, want_dist=False, device=0, stream=None):
    """gbnns_exact_knn: ids [n_q x k] (and distances) of the k nearest base rows of every query, in the
    reference's distance arithmetic, ascending (distance, id).  numpy in -> numpy out; torch CUDA in -> torch
    out.  self_offset >= 0: query i is base row i + self_offset (excluded from its own list)."""
    lib = load_library()
    dev = _is_dev(base)
    if _is_dev(queries) != dev:
        raise TypeError("base and queries must live in the same memory kind")
    base = _prep(base, np.float32, "float32")
    queries = _prep(queries, np.float32, "float32")
    n, d = int(base.shape[0]), int(base.shape[1])
    nq = int(queries.shape[0])
    if int(queries.shape[1]) != d:
        raise ValueError("dimension mismatch")
    if dev:
        import torch
        ids = torch.empty((nq, k), dtype=torch.int32, device=base.device)
        dist = torch.empty((nq, k), dtype=torch.float32, device=base.device) if want_dist else None
        if stream is None:
            stream = torch.cuda.current_stream(base.device)
        sptr = stream.cuda_stream
        device = base.device.index or 0
    else:
        ids = np.empty((nq, k), np.uint32)
        dist = np.empty((nq, k), np.float32) if want_dist else None
        sptr = None
    _check(lib.gbnns_exact_knn(device, _ptr(base), n, _ptr(queries), nq, d, k, metric, self_offset, _ptr(ids),
                               _ptr(dist), MEM_DEVICE if dev else MEM_HOST, sptr))
    return (ids, dist) if want_dist else ids


def _check_multi(rc):
    if rc != 0:
        raise GbnnsError(rc, load_library().gbnns_multi_last_error().decode(errors="replace"))


class MultiIndex:
    """gbnns_multi: one replica of the index per entry of `devices` (host arrays in), a batch is cut into contiguous
    blocks, one per replica, each on its own host thread and HIP stream."""

    def __init__(self, db, graph_offsets, graph_nbrs, db_low=None, net=None, metric=METRIC_L2, devices=None):
        lib = load_library()
        self._lib = lib
        self._h = C.c_void_p()
        db = _host(db, np.float32)
        db_low = None if db_low is None else _host(db_low, np.float32)
        net = None if net is None else tuple(_host(x, np.float32) for x in net)
        off, nbr = _host(graph_offsets, np.uint64), _host(graph_nbrs, np.uint32)
        self.n, self.d = int(db.shape[0]), int(db.shape[1])
        self.d_low = int(db_low.shape[1]) if db_low is not None else 0
        self.d_hidden = int(net[0].shape[0]) if net is not None else 0
        desc = _IndexDesc(
            struct_size=C.sizeof(_IndexDesc), device=0, metric=metric, mem_kind=MEM_HOST, n=self.n, d=self.d,
            d_low=self.d_low, d_hidden=self.d_hidden, db=_ptr(db), db_low=_ptr(db_low), graph_offsets=_ptr(off),
            graph_nbrs=_ptr(nbr), net_l1=_ptr(net[0]) if net else None, net_l2=_ptr(net[1]) if net else None,
            net_l3=_ptr(net[2]) if net else None)
        devs = None if devices is None else np.ascontiguousarray(devices, np.int32)
        _check_multi(lib.gbnns_multi_create(C.byref(desc), _ptr(devs), 0 if devs is None else len(devs), C.byref(self._h)))
        self.size = lib.gbnns_multi_size(self._h)
        self.devices = [lib.gbnns_multi_device_of(self._h, i) for i in range(self.size)]

    def close(self):
        if self._h:
            self._lib.gbnns_multi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def shard_bounds(self, n_q, part):
        lo, hi = C.c_uint64(), C.c_uint64()
        self._lib.gbnns_shard_bounds(n_q, self.size, part, C.byref(lo), C.byref(hi))
        return lo.value, hi.value

    def set_aux_graph(self, offsets, nbrs):
        off = None if offsets is None else _host(offsets, np.uint64)
        nbr = None if nbrs is None else _host(nbrs, np.uint32)
        _check_multi(self._lib.gbnns_multi_set_aux_graph(self._h, _ptr(off), _ptr(nbr)))

    def search(self, queries, ef, mode=MODE_NET, k=1, queries_low=None, entry_ids=None, want=("hops", "dist_calc"),
               flags=0, hops_bound=0):
        """gbnns_multi_search_ex: host arrays holding the whole batch in, numpy results out."""
        q = _host(queries, np.float32)
        ql = None if queries_low is None else _host(queries_low, np.float32)
        ent = None if entry_ids is None else _host(entry_ids, np.uint32)
        nq = q.shape[0]
        kk = ef if mode != MODE_PLAIN else max(1, min(k, ef))
        res = {"ids": np.empty(nq, np.uint32)}
        a = _SearchArgs(struct_size=C.sizeof(_SearchArgs), mode=mode, ef=ef, k=kk, mem_kind=MEM_HOST, n_q=nq,
                        queries=_ptr(q), queries_low=_ptr(ql), entry_ids=_ptr(ent), out_ids=_ptr(res["ids"]),
                        flags=flags, hops_bound=hops_bound,
                        n_entries=int(ent.shape[1]) if ent is not None and ent.ndim == 2 else 0)
        if "hops" in want:
            res["hops"] = np.empty(nq, np.int32); a.out_hops = _ptr(res["hops"])
        if "dist_calc" in want:
            res["dist_calc"] = np.empty(nq, np.int32); a.out_dist_calc = _ptr(res["dist_calc"])
        if "cand" in want:
            res["cand"] = np.empty((nq, kk), np.uint32); a.out_cand = _ptr(res["cand"])
        if "cand_dist" in want:
            res["cand_dist"] = np.empty((nq, kk), np.float32); a.out_cand_dist = _ptr(res["cand_dist"])
        if "q_low" in want:
            res["q_low"] = np.empty((nq, self.d_low), np.float32); a.out_q_low = _ptr(res["q_low"])
        _check_multi(self._lib.gbnns_multi_search_ex(self._h, C.byref(a)))
        return res

    def search_device(self, query_blocks, ef, n_q, mode=MODE_NET, entry_blocks=None, flags=0):
        """gbnns_multi_search_device: query_blocks[r] = torch tensor on replica r's device; returns one [n_q] int32
        tensor per replica (every one holds all answers once synchronize() has returned)."""
        import torch
        outs = [torch.empty(n_q, dtype=torch.int32, device=b.device) for b in query_blocks]
        qp = (C.c_void_p * self.size)(*[b.data_ptr() for b in query_blocks])
        op = (C.c_void_p * self.size)(*[o.data_ptr() for o in outs])
        ep = None if entry_blocks is None else (C.c_void_p * self.size)(*[e.data_ptr() for e in entry_blocks])
        a = _SearchArgs(struct_size=C.sizeof(_SearchArgs), mode=mode, ef=ef, k=1, flags=flags)
        _check_multi(self._lib.gbnns_multi_search_device(self._h, C.byref(a), n_q, qp, ep, op))
        self._keep = (query_blocks, entry_blocks, outs)
        return outs

    def synchronize(self):
        _check_multi(self._lib.gbnns_multi_synchronize(self._h))

    def rccl_single_rank(self, on=True):
        """One replica: run the exchange leg all the same (one-rank communicator, ncclAllGather of the single block)."""
        _check_multi(self._lib.gbnns_multi_rccl_single_rank(self._h, int(on)))

    def rccl_version(self):
        """ncclGetVersion() of the librccl this handle has loaded, 0 while it has not loaded one."""
        return int(self._lib.gbnns_multi_rccl_version(self._h))


class Index:
    """One dataset resident in HBM (gbnns_index).  db / db_low / net may be numpy arrays (copied
    to the device) or torch CUDA tensors (borrowed; kept alive by this object)."""

    def __init__(self, db, graph_offsets, graph_nbrs, db_low=None, net=None, metric=METRIC_L2,
                 device=0):
        lib = load_library()
        self._lib = lib
        self._h = C.c_void_p()
        dev = _is_dev(db)
        db = _prep(db, np.float32, "float32")
        db_low = _prep(db_low, np.float32, "float32")
        if db_low is not None and _is_dev(db_low) != dev:
            raise TypeError("db and db_low must live in the same memory kind")
        if net is not None:
            net = tuple(_prep(x, np.float32, "float32") for x in net)
            if any(_is_dev(x) != dev for x in net):
                raise TypeError("net layers must live in the same memory kind as db")
        off = _host(graph_offsets, np.uint64)
        nbr = _host(graph_nbrs, np.uint32)
        self.n, self.d = int(db.shape[0]), int(db.shape[1])
        self.d_low = int(db_low.shape[1]) if db_low is not None else 0
        self.d_hidden = int(net[0].shape[0]) if net is not None else 0
        if off.shape[0] != self.n + 1:
            raise ValueError("graph_offsets must have n+1 entries")
        if net is not None:
            shapes = [tuple(x.shape) for x in net]
            want = [(self.d_hidden, self.d + 1), (self.d_hidden, self.d_hidden + 1),
                    (self.d_low, self.d_hidden + 1)]
            if shapes != want:
                raise ValueError(f"net layer shapes {shapes} != {want}")
        desc = _IndexDesc(
            struct_size=C.sizeof(_IndexDesc), device=device, metric=metric,
            mem_kind=MEM_DEVICE if dev else MEM_HOST, n=self.n, d=self.d, d_low=self.d_low,
            d_hidden=self.d_hidden, db=_ptr(db), db_low=_ptr(db_low), graph_offsets=_ptr(off),
            graph_nbrs=_ptr(nbr), net_l1=_ptr(net[0]) if net else None,
            net_l2=_ptr(net[1]) if net else None, net_l3=_ptr(net[2]) if net else None)
        _check(lib.gbnns_index_create(C.byref(desc), C.byref(self._h)))
        self._in_flight = collections.deque(maxlen=8)  # (inputs, outputs) of the deferred calls not yet joined
        self._last = None
        self._keep = (db, db_low, net) if dev else None
        self.metric = metric
        self.device = device

    def set_aux_graph(self, offsets, nbrs):
        """The reference's auxiliary_graph (host CSR); None, None removes it."""
        if offsets is None:
            _check(self._lib.gbnns_index_set_aux_graph(self._h, None, None))
            return
        off = _host(offsets, np.uint64)
        nbr = _host(nbrs, np.uint32)
        if off.shape[0] != self.n + 1:
            raise ValueError("auxiliary graph offsets must have n+1 entries")
        _check(self._lib.gbnns_index_set_aux_graph(self._h, _ptr(off), _ptr(nbr)))

    def close(self):
        if self._h:
            self._lib.gbnns_index_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- search ------------------------------------------------------------------------------
    def search(self, queries, ef, mode=MODE_NET, k=1, queries_low=None, entry_ids=None,
               want=("hops", "dist_calc"), hash_capacity=0, stream=None, out=None, flags=0,
               aux=False, llf=False, hops_bound=50, defer_depth=0):
        """Runs one batch.  numpy queries -> synchronous call, numpy results.  torch CUDA queries
        -> enqueued on `stream` (torch stream or None = current), torch results, no sync.  Pinned torch CPU queries ->
        HOST buffers in page-locked memory, pinned torch results: synchronous, or with FLAG_DEFER_JOIN a host batch in
        flight (copy in, kernels and ids out on a lane's stream; wait() / join() + stream.synchronize() to read them).
        `want` may also name "cand", "cand_dist", "q_low", "edges".  Returns a dict with "ids" + wanted.
        aux / llf / hops_bound: the reference's use_second_graph walk over set_aux_graph()'s graph."""
        if aux:
            flags |= FLAG_AUX_GRAPH | (FLAG_LLF if llf else 0)
        dev = _is_dev(queries) and queries.is_cuda
        pinned = _is_dev(queries) and not queries.is_cuda
        queries = _prep(queries, np.float32, "float32")
        queries_low = _prep(queries_low, np.float32, "float32")
        nq = int(queries.shape[0])
        if queries.shape[1] != self.d:
            raise ValueError("query dimension mismatch")
        kk = ef if mode != MODE_PLAIN else max(1, min(k, ef))
        res = {} if out is None else out
        if dev:
            import torch
            tdev = queries.device
            entry_ids = None if entry_ids is None else entry_ids.to(torch.int32).contiguous() \
                if entry_ids.dtype != torch.int32 else entry_ids.contiguous()

            def alloc(name, shape, dtype):
                if name not in res:
                    res[name] = torch.empty(shape, dtype=dtype, device=tdev)
                return res[name]
            i32, f32 = torch.int32, torch.float32
            if stream is None:
                stream = torch.cuda.current_stream(tdev)
            sptr = stream.cuda_stream
        elif pinned:
            import torch
            if entry_ids is not None and not (_is_dev(entry_ids) and entry_ids.dtype == torch.int32 and entry_ids.is_pinned()):
                raise TypeError("entry_ids must be a pinned int32 tensor beside pinned queries")

            def alloc(name, shape, dtype):
                if name not in res:
                    res[name] = torch.empty(shape, dtype=dtype, pin_memory=True)
                return res[name]
            i32, f32 = torch.int32, torch.float32
            if stream is None:
                stream = torch.cuda.current_stream(self.device)  # (the index's device, whatever the current one is)
            sptr = stream.cuda_stream
        else:
            entry_ids = None if entry_ids is None else _host(entry_ids, np.uint32)

            def alloc(name, shape, dtype):
                if name not in res:
                    res[name] = np.empty(shape, dtype=dtype)
                return res[name]
            i32, f32 = np.int32, np.float32
            sptr = None
        n_entries = 0
        if entry_ids is not None and len(entry_ids.shape) == 2:
            n_entries = int(entry_ids.shape[1])   # several entry points per query (row-major)
        ids = alloc("ids", (nq,), i32 if (dev or pinned) else np.uint32)
        a = _SearchArgs(struct_size=C.sizeof(_SearchArgs), mode=mode, ef=ef, k=kk,
                        mem_kind=MEM_DEVICE if dev else MEM_HOST, hash_capacity=hash_capacity,
                        n_q=nq, queries=_ptr(queries), queries_low=_ptr(queries_low),
                        entry_ids=_ptr(entry_ids), out_ids=_ptr(ids), stream=sptr, flags=flags,
                        hops_bound=hops_bound if aux else 0, n_entries=n_entries, defer_depth=defer_depth)
        if "hops" in want:
            a.out_hops = _ptr(alloc("hops", (nq,), i32))
        if "dist_calc" in want:
            a.out_dist_calc = _ptr(alloc("dist_calc", (nq,), i32))
        if "cand" in want:
            a.out_cand = _ptr(alloc("cand", (nq, kk), i32 if (dev or pinned) else np.uint32))
        if "cand_dist" in want:
            a.out_cand_dist = _ptr(alloc("cand_dist", (nq, kk), f32))
        if "q_low" in want:
            a.out_q_low = _ptr(alloc("q_low", (nq, self.d_low), f32))
        if "edges" in want:
            a.out_edges = _ptr(alloc("edges", (nq,), i32))
        _check(self._lib.gbnns_search_ex(self._h, C.byref(a)))
        # Inputs (and the result buffers handed out) stay referenced while a lane stream may still read / write them: torch's
        # caching allocators do not know the library's internal streams.  A plain call: until the next call; deferred
        # calls: the last four (= the most lanes a handle has), until wait() / join() / a plain call joins them.
        held = (queries, queries_low, entry_ids, res)
        if flags & FLAG_DEFER_JOIN:
            self._in_flight.append(held)
        else:
            self._in_flight.clear()
        self._last = held
        return res

    def wait(self, keep=0):
        """gbnns_index_wait: blocks until every FLAG_DEFER_JOIN batch but the `keep` most recent has finished."""
        _check(self._lib.gbnns_index_wait(self._h, keep))
        while len(self._in_flight) > keep:
            self._in_flight.popleft()

    def join(self):
        """gbnns_index_join: the stream of the last FLAG_DEFER_JOIN call waits for that call's pieces."""
        _check(self._lib.gbnns_index_join(self._h))
        self._in_flight.clear()  # (the caller's stream now waits for the lanes: stream-ordered reuse of the buffers is safe)

    def search_batch(self, queries, ef, entry_ids=None, want_cand=False):
        """The plain 9-argument C entry point (NET mode, host buffers)."""
        q = _host(queries, np.float32)
        nq = q.shape[0]
        ids = np.empty(nq, np.uint32)
        hops = np.empty(nq, np.int32)
        dc = np.empty(nq, np.int32)
        cand = np.empty((nq, ef), np.uint32) if want_cand else None
        ent = None if entry_ids is None else _host(entry_ids, np.uint32)
        _check(self._lib.gbnns_search_batch(self._h, q.ctypes.data, nq, ef, _ptr(ent),
                                            ids.ctypes.data, hops.ctypes.data, dc.ctypes.data,
                                            _ptr(cand)))
        r = dict(ids=ids, hops=hops, dist_calc=dc)
        if want_cand:
            r["cand"] = cand
        return r

    def project(self, x, stream=None):
        """GetLowQueryFromNet over the rows of x -> [rows x d_low]."""
        dev = _is_dev(x)
        x = _prep(x, np.float32, "float32")
        if dev:
            import torch
            out = torch.empty((x.shape[0], self.d_low), dtype=torch.float32, device=x.device)
            if stream is None:
                stream = torch.cuda.current_stream(x.device)
            sptr = stream.cuda_stream
        else:
            out = np.empty((x.shape[0], self.d_low), np.float32)
            sptr = None
        _check(self._lib.gbnns_project(self._h, _ptr(x), x.shape[0], _ptr(out),
                                       MEM_DEVICE if dev else MEM_HOST, sptr))
        return out

    def rerank(self, queries, cand, count=None):
        """getRealNearest over a batch (host buffers): cand [nq x stride] in pop order."""
        q = _host(queries, np.float32)
        cand = _host(cand, np.uint32)
        count = None if count is None else _host(count, np.int32)
        out = np.empty(q.shape[0], np.uint32)
        _check(self._lib.gbnns_rerank(self._h, q.ctypes.data, q.shape[0], cand.ctypes.data,
                                      cand.shape[1], _ptr(count), out.ctypes.data, MEM_HOST, None))
        return out

    # -- profiling -----------------------------------------------------------------------------
    def profile_enable(self, on=True):
        _check(self._lib.gbnns_profile_enable(self._h, int(on)))

    def knob(self, name, value):
        """A diagnostic knob of THIS handle (gbnns_index_knob; include/gbnns.h lists them).  Results never depend on it."""
        _check(self._lib.gbnns_index_knob(self._h, name.encode() if isinstance(name, str) else name, int(value)))

    def knob_get(self, name):
        v = C.c_int(0)
        _check(self._lib.gbnns_index_knob_get(self._h, name.encode() if isinstance(name, str) else name, C.byref(v)))
        return v.value

    def profile_read(self, reset=True):
        p = Profile()
        p.struct_size = C.sizeof(Profile)   # the callee writes no more than this
        _check(self._lib.gbnns_profile_read(self._h, C.byref(p), int(reset)))
        return p.as_dict()
