"""Host-side helpers of the C++ drop-in against golden vectors from the COMPILED REFERENCE (tests/golden/
utils_toy.npz, made by tests/golden/make_golden_utils.py): hnswlikeGD(need_const_degree = true), cutKNNbyK /
cutKNNbyThreshold / mergeGraph / fillGraphToConstantDegree, the KLgraph builders (support_classes.h:38-175),
createUniformData, and the fvecs / bvecs / mmap loaders (dim_red/data.py:69-78).  The functions are called through
tests/cpp/dropin_units.cpp, a driver compiled against the drop-in headers; none of them needs a GPU (the graph
builder runs its host path here: GBNNS_GD_HOST=1)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import datagen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden_utils as mgu  # noqa: E402  (inputs() only: the golden file itself is committed)

LIBDIR = os.path.join(ROOT, "gbnns_dim_red_amd", "lib")


@pytest.fixture(scope="module")
def unit(tmp_path_factory):
    import gbnns_dim_red_amd as g
    g.build_library()
    exe = str(tmp_path_factory.mktemp("units") / "dropin_units")
    # GBNNS_UNITS_SAN=1 (tests/test_sanitizers.py): the same driver with AddressSanitizer + UBSan
    opt = ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"] \
        if os.environ.get("GBNNS_UNITS_SAN") == "1" else ["-O2"]
    subprocess.check_call(["g++"] + opt + ["-std=c++11", "-ffp-contract=off", "-fno-fast-math", "-w", "-o", exe,
                           os.path.join(ROOT, "tests", "cpp", "dropin_units.cpp"), "-L" + LIBDIR, "-lgbnns_hip",
                           "-Wl,-rpath," + LIBDIR, "-lpthread"])

    def run(*args):
        p = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, GBNNS_GD_HOST="1"))
        assert p.returncode == 0, p.stdout + p.stderr
        return p.stdout
    return run


@pytest.fixture(scope="module")
def data(tmp_path_factory):
    d = tmp_path_factory.mktemp("unitdata")
    c, t, knn, knn_t, lists = mgu.inputs()
    _write_xvecs(d / "c.fvecs", c.base)
    _write_xvecs(d / "t.fvecs", t.base)
    _write_edges(d / "c_knn.ivecs", list(knn))
    _write_edges(d / "t_knn.ivecs", list(knn_t))
    _write_edges(d / "ragged.ivecs", lists)
    gold = np.load(os.path.join(ROOT, "tests", "golden", "utils_toy.npz"))
    return dict(dir=d, c=c, t=t, gold=gold)


def _write_xvecs(path, a):
    a = np.ascontiguousarray(a)
    rec = np.empty((a.shape[0], a.shape[1] + 1), np.uint32)
    rec[:, 0] = a.shape[1]
    rec[:, 1:] = a.view(np.uint32)
    rec.tofile(path)


def _write_edges(path, lists):
    with open(path, "wb") as f:
        for row in lists:
            np.array([len(row)], np.uint32).tofile(f)
            np.asarray(row, np.uint32).tofile(f)


def _read_edges(path, n):
    raw = np.fromfile(path, np.uint32)
    off, nbr, p = [0], [], 0
    for _ in range(n):
        k = int(raw[p])
        nbr.append(raw[p + 1:p + 1 + k])
        p += 1 + k
        off.append(off[-1] + k)
    assert p == raw.size
    return np.array(off, np.uint64), (np.concatenate(nbr) if nbr else np.zeros(0, np.uint32))


def _same(path, n, gold, key):
    off, nbr = _read_edges(path, n)
    assert np.array_equal(off, gold[key + "_off"]), key
    assert np.array_equal(nbr, gold[key + "_nbr"]), key


def test_const_degree_builder(unit, data):
    d = data["dir"]
    for name, case in (("c", data["c"]), ("t", data["t"])):
        for M, rev in ((6, 1), (10, 0), (14, 1)):
            out = d / f"cd_{name}_{M}_{rev}.ivecs"
            unit("constdeg", d / f"{name}_knn.ivecs", d / f"{name}.fvecs", case.n, case.d, M, rev, out)
            _same(out, case.n, data["gold"], f"constdeg_{name}_{M}_{rev}")


def test_cut_and_merge_helpers(unit, data):
    d, c, t = data["dir"], data["c"], data["t"]
    for name, case in (("c", c), ("t", t)):
        for k in (1, 7, 25, 64):
            out = d / f"cutk_{name}_{k}.ivecs"
            unit("cutk", d / f"{name}_knn.ivecs", d / f"{name}.fvecs", case.n, case.d, k, out)
            _same(out, case.n, data["gold"], f"cutk_{name}_{k}")
    for thr in (0.05, 0.4, 2.0):
        out = d / f"cutthr_{thr}.ivecs"
        unit("cutthr", d / "c_knn.ivecs", d / "c.fvecs", c.n, c.d, thr, out)
        _same(out, c.n, data["gold"], f"cutthr_{thr}")
    # merge: the golden GD graph (M = 6, reverse) + the ragged random lists
    g6 = d / "g6.ivecs"
    off, nbr = data["gold"]["constdeg_c_6_1_off"], data["gold"]["constdeg_c_6_1_nbr"]
    _write_edges(g6, [nbr[int(off[i]):int(off[i + 1])] for i in range(c.n)])
    unit("merge", g6, d / "ragged.ivecs", c.n, d / "merged.ivecs")
    _same(d / "merged.ivecs", c.n, data["gold"], "merge")
    for deg in (4, 16, 30):
        out = d / f"fill_{deg}.ivecs"
        unit("fill", d / "ragged.ivecs", d / "c_knn.ivecs", c.n, deg, out)
        _same(out, c.n, data["gold"], f"fill_{deg}")


def test_kl_graph_builders(unit, data):
    """Same seed, one thread -> the reference's graph, link for link (std::mt19937 and the distributions are the
    library's: the test pins the draw order and the container semantics of the restatement)."""
    d, c = data["dir"], data["c"]
    _write_xvecs(d / "c200.fvecs", c.base[:200])
    for which, l, sq, seed in ((1, 5, 24, 7), (1, 15, 20, 12345), (0, 4, 0, 3), (2, 3, 0, 99)):
        n = c.n if which != 0 else 200
        src = d / ("c.fvecs" if which != 0 else "c200.fvecs")
        out = d / f"kl_{which}_{seed}.ivecs"
        unit("kl", which, l, src, n, c.d, sq, seed, out)
        _same(out, n, data["gold"], f"kl_{which}_{l}_{sq}_{seed}")


def test_create_uniform_data(unit, data):
    d = data["dir"]
    for n, dim, seed in ((50, 3, 1), (40, 17, 2)):
        out = d / f"uni_{n}.fvecs"
        unit("uniform", n, dim, seed, out)
        got = np.fromfile(out, np.uint32).reshape(n, dim + 1)
        assert (got[:, 0] == dim).all()
        assert np.array_equal(got[:, 1:], data["gold"][f"uniform_{n}_{dim}_{seed}"])


def test_vector_loaders_fvecs_bvecs_mmap(unit, data):
    """loadVectorsAny: <prefix>.fvecs through the stream reader or a mapping (GBNNS_MMAP=1); <prefix>.bvecs (bigann
    byte vectors, [int32 dim][dim x uint8]) converted to float like dim_red/data.py's mmap_bvecs consumers do; a
    record with a wrong dimension is fatal ("file error", exit 1) as in readXvec."""
    d = data["dir"]
    rng = np.random.Generator(np.random.PCG64(77))
    n, dim = 37, 20
    x = rng.standard_normal((n, dim)).astype(np.float32)
    _write_xvecs(d / "v.fvecs", x)
    unit("bvecs", d / "v", n, dim, d / "v_out.fvecs")
    assert np.array_equal(np.fromfile(d / "v_out.fvecs", np.uint32).reshape(n, dim + 1)[:, 1:], x.view(np.uint32))
    os.environ["GBNNS_MMAP"] = "1"
    try:
        unit("bvecs", d / "v", n, dim, d / "v_out2.fvecs")
    finally:
        del os.environ["GBNNS_MMAP"]
    assert np.array_equal(np.fromfile(d / "v_out2.fvecs", np.uint32).reshape(n, dim + 1)[:, 1:], x.view(np.uint32))
    b = rng.integers(0, 256, size=(n, dim), dtype=np.uint8)
    rec = np.empty((n, 4 + dim), np.uint8)
    rec[:, :4] = np.frombuffer(np.int32(dim).tobytes(), np.uint8)
    rec[:, 4:] = b
    rec.tofile(d / "w.bvecs")
    unit("bvecs", d / "w", n, dim, d / "w_out.fvecs")
    got = np.fromfile(d / "w_out.fvecs", np.uint32).reshape(n, dim + 1)[:, 1:].view(np.float32)
    assert np.array_equal(got, b.astype(np.float32))
    # wrong dimension in the file -> the reference's fatal path ("file error", exit 1)
    with pytest.raises(AssertionError):
        unit("bvecs", d / "w", n, dim + 1, d / "w_bad.fvecs")


def test_make_step_shim_vs_reference(unit, tmp_path):
    """makeStep (search_function.h:15-40) as a host shim of the drop-in: one step on random heaps, visited marks and
    neighbour lists -- part-filled and full result heaps, neighbours already visited, exact ties on a lattice --
    against the compiled reference's makeStep: dist_calc, found, marks, both heaps in pop order (keys bit for bit)."""
    import oracle
    if not oracle.have_ref():
        pytest.skip("compiled reference (oracle/_ref) not present")
    ref = oracle.Ref()
    rng = np.random.Generator(np.random.PCG64(77))
    for case in range(40):
        n, d = int(rng.integers(30, 200)), int(rng.choice([4, 8, 13, 32]))
        db = (rng.integers(-4, 5, size=(n, d)) / 2.0).astype(np.float32) if case % 2 else rng.standard_normal((n, d)).astype(np.float32)
        q = db[int(rng.integers(0, n))] + (0 if case % 4 == 1 else rng.standard_normal(d).astype(np.float32) * 0.1)
        q = q.astype(np.float32)
        ef = int(rng.integers(1, 12))
        ids = rng.permutation(n)
        n_top = int(rng.integers(1, ef + 1))
        top_id = ids[:n_top].astype(np.uint32)
        top_key = np.array([ref.l2(q, db[i]) for i in top_id], np.float32)
        n_cand = int(rng.integers(0, 6))
        cand_id = ids[n_top:n_top + n_cand].astype(np.uint32)
        cand_key = -np.array([ref.l2(q, db[i]) for i in cand_id], np.float32)
        visited = np.concatenate([top_id, cand_id, ids[40:40 + int(rng.integers(0, 10))].astype(np.uint32)])
        nb = rng.choice(n, size=int(rng.integers(0, 25)), replace=False).astype(np.uint32)
        want = ref.make_step(db, q, nb, ef, (top_key, top_id), (cand_key, cand_id), visited)
        blob = tmp_path / "case.bin"
        with open(blob, "wb") as f:
            np.array([n, d, ef, len(nb), n_top, n_cand, len(visited)], np.uint32).tofile(f)
            db.tofile(f)
            q.tofile(f)
            nb.tofile(f)
            for k, i in list(zip(top_key, top_id)) + list(zip(cand_key, cand_id)):
                np.array([k], np.float32).tofile(f)
                np.array([i], np.uint32).tofile(f)
            visited.tofile(f)
        unit("makestep", blob, tmp_path / "out.bin")
        raw = np.fromfile(tmp_path / "out.bin", np.uint32)
        dc, found, nt, nc, marked = (int(x) for x in raw[:5])
        pairs = raw[5:].reshape(-1, 2)
        assert (dc, bool(found), marked) == (want["dist_calc"], want["found"], want["marked"]), case
        assert nt == len(want["top"][0]) and nc == len(want["cand"][0]), case
        assert np.array_equal(pairs[:nt, 0], want["top"][0].view(np.uint32)) and np.array_equal(pairs[:nt, 1], want["top"][1]), case
        assert np.array_equal(pairs[nt:, 0], want["cand"][0].view(np.uint32)) and np.array_equal(pairs[nt:, 1], want["cand"][1]), case


def test_visited_list_pool_stand_in(unit):
    """The source-compatibility VisitedList / VisitedListPool (the device keeps its visited sets in LDS): hand-out,
    on-demand growth, epoch reset across the uint16 wrap, release, destruction (new[] / delete[] paired, unlike the
    reference's visited_list_pool.h:30 -- which the sanitizer run of this test would flag)."""
    assert unit("vlpool", 97).strip() == "vlpool ok stale 0"


def test_reference_drivers_compile_against_dropin_headers(tmp_path):
    """The boundary pin: the reference's OWN drivers (final_test.cpp, naive_test.cpp, prepare_graph.cpp), copied to a
    scratch directory so that their `#include "search_function.h"` resolves to the drop-in headers, must compile
    unchanged.  Skipped where /root/reference is absent (the GPU box)."""
    refdir = "/root/reference/search"
    if not os.path.exists(os.path.join(refdir, "final_test.cpp")):
        pytest.skip("reference sources not present")
    import shutil
    for name in ("final_test.cpp", "naive_test.cpp", "prepare_graph.cpp"):
        shutil.copy(os.path.join(refdir, name), tmp_path / name)
        p = subprocess.run(["g++", "-std=c++11", "-fopenmp", "-fsyntax-only", "-w", "-I",
                            os.path.join(ROOT, "gbnns_dim_red_amd", "search"), str(tmp_path / name)],
                           capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, name + "\n" + p.stderr[-3000:]
