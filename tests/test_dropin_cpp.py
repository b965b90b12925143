"""The C++ drop-in (gbnns_dim_red_amd/search/final_test) end to end on the GPU: files in the
reference's formats in, result lines out, compared with the lines the reference's own harness
(performRealNetTests / performRealTests, compiled reference) printed for the same inputs."""
import os
import subprocess

import numpy as np
import pytest

import datagen
import golden_util as gu
import oracle as orc_mod

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "gbnns_dim_red_amd", "search", "final_test")


def write_xvecs(path, a):
    a = np.ascontiguousarray(a)
    n, d = a.shape
    rec = np.empty((n, d + 1), np.uint32)
    rec[:, 0] = d
    rec[:, 1:] = a.view(np.uint32)
    rec.tofile(path)


def write_edges(path, off, nbr):
    with open(path, "wb") as f:
        for i in range(len(off) - 1):
            row = nbr[int(off[i]):int(off[i + 1])]
            np.array([len(row)], np.uint32).tofile(f)
            row.astype(np.uint32).tofile(f)


@pytest.mark.parametrize("name,devices", [("sift_toy", None), ("tail_toy", None), ("ties_toy", None),
                                          # GBNNS_DEVICES: query blocks over several replicas (gbnns_multi_*); a one-GPU
                                          # box lists its device twice -- same result lines
                                          ("sift_toy", "0,0"), ("ties_toy", "0,0,0")])
def test_final_test_result_lines(tmp_path, orc, name, devices):
    assert os.path.exists(BIN), "build() must produce the drop-in driver"
    gd = gu.load(name)
    c = gd.case
    ds = "toy"
    data = tmp_path / "data"
    models = tmp_path / "models"
    data.mkdir()
    models.mkdir()
    db_low = orc.project(c.net, c.base, threads=4)
    write_xvecs(data / f"{ds}_base.fvecs", c.base)
    write_xvecs(data / f"{ds}_query.fvecs", c.queries)
    write_xvecs(data / f"{ds}_groundtruth.ivecs", gd["truth"])
    write_xvecs(data / f"{ds}_base_angular_optimal.fvecs", db_low)
    off, nbr = gd.graph
    write_edges(models / "hnsw_toygraph.ivecs", off, nbr)
    write_edges(models / "hnsw_toygraph_angular_optimal.ivecs", off, nbr)
    for i, layer in enumerate(c.net, 1):
        write_xvecs(models / f"{ds}_net_as_matrix_angular_optimal_{i}.fvecs", layer)
    efs = ",".join(str(e) for e in gd.efs)
    params = tmp_path / "params.txt"
    params.write_text("\n".join([
        f"{ds} n {c.n}", f"{ds} n_q {c.nq}", f"{ds} n_tr {gd['truth'].shape[1]}", f"{ds} d {c.d}",
        # (the tail_toy case has no d_hidden row, as gist / deep in the reference's file: the width comes from second_part)
        f"{ds} d_low {c.dlow}", (f"{ds} second_part _{c.dlow}_l_2_1m_5_40_w_{c.dh}_e_40" if name == "tail_toy" else f"{ds} d_hidden {c.dh}"),
        f"{ds} efs {efs}", f"{ds} efs_hnsw {efs}",
        f"{ds} hnsw_name toygraph", "other n 5", "# comment line with three tokens"]) + "\n")
    env = dict(os.environ, GBNNS_NUM_EXPER="2")
    if devices:
        env["GBNNS_DEVICES"] = devices
    p = subprocess.run([BIN, ds, str(data), str(models), str(tmp_path), str(params)], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    lines = open(tmp_path / f"final_results_{ds}.txt").read().splitlines()
    got = sorted(ln.split(" work_time ")[0] for ln in lines)
    assert got == sorted(gd.meta["result_lines"])
    # the same lines were echoed to stdout, and every one carries a positive work_time
    for ln in lines:
        assert ln in p.stdout
        assert float(ln.split(" work_time ")[1]) > 0


def test_final_test_rejects_bad_files(tmp_path):
    # wrong dimension in an fvecs file -> "file error" + exit(1), as the reference does
    (tmp_path / "params.txt").write_text("toy n 2\ntoy n_q 1\ntoy n_tr 2\ntoy d 4\ntoy d_low 2\n"
                                         "toy d_hidden 4\ntoy efs 1\ntoy efs_hnsw 1\ntoy hnsw_name g\n")
    write_xvecs(tmp_path / "toy_base.fvecs", np.zeros((2, 3), np.float32))
    p = subprocess.run([BIN, "toy", str(tmp_path), str(tmp_path), str(tmp_path),
                        str(tmp_path / "params.txt")], capture_output=True, text=True, timeout=60)
    assert p.returncode == 1 and "file error" in p.stdout
    p = subprocess.run([BIN], capture_output=True, text=True, timeout=60)
    assert p.returncode == 1 and "Need to specify parameters" in p.stdout


def test_naive_test_result_lines(tmp_path, orc):
    """The second driver (naive_test.cpp:98-105): plain, knn, knn + auxiliary KL graph with llf, and the same in
    the low-dim space with re-rank -- result lines equal to the reference harness's (tests/golden/aux_toy.npz)."""
    import json
    nbin = os.path.join(ROOT, "gbnns_dim_red_amd", "search", "naive_test")
    assert os.path.exists(nbin), "build() must produce the naive_test driver"
    z = np.load(gu.GOLDEN_DIR + "/aux_toy.npz")
    meta = json.loads(bytes(z["meta"]).decode())
    gd = gu.load("sift_toy")
    c = gd.case
    ds = "toy"
    data = tmp_path / "data"
    models = tmp_path / "models"
    data.mkdir()
    models.mkdir()
    db_low = orc.project(c.net, c.base, threads=4)
    q_low = orc.project(c.net, c.queries)
    write_xvecs(data / f"{ds}_base.fvecs", c.base)
    write_xvecs(data / f"{ds}_query.fvecs", c.queries)
    write_xvecs(data / f"{ds}_groundtruth.ivecs", gd["truth"])
    write_xvecs(data / f"{ds}_base_naive.fvecs", db_low)
    write_xvecs(data / f"{ds}_querynaive.fvecs", q_low)
    off, nbr = gd.graph
    aoff, anbr = z["sift_toy_aux_off"], z["sift_toy_aux_nbr"]
    for fn in ("hnsw_toygraph.ivecs", f"{ds}knn.ivecs", f"{ds}knn_low.ivecs"):
        write_edges(models / fn, off, nbr)
    for fn in (f"{ds}_kl_sqrt_style.ivecs", f"{ds}_kl_llow_sqrt_style.ivecs"):
        write_edges(models / fn, aoff, anbr)
    efs = ",".join(str(e) for e in meta["cases"]["sift_toy"]["efs"])
    params = tmp_path / "params.txt"
    params.write_text("\n".join([
        f"{ds} n {c.n}", f"{ds} n_q {c.nq}", f"{ds} n_tr {gd['truth'].shape[1]}", f"{ds} d {c.d}",
        f"{ds} d_low {c.dlow}", f"{ds} kl_size 5", f"{ds} efs {efs}", f"{ds} efs_hnsw {efs}",
        f"{ds} hnsw_name toygraph"]) + "\n")
    env = dict(os.environ, GBNNS_NUM_EXPER="2", GBNNS_SEED=str(meta["naive_seed"]))
    p = subprocess.run([nbin, ds, str(data), str(models), str(tmp_path), str(params)], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    lines = open(tmp_path / f"naive_results_{ds}.txt").read().splitlines()
    got = [ln.split(" work_time ")[0] for ln in lines]
    assert got == meta["naive_result_lines"]
    # the JSON sidecar: one object per result line, same order, same recall / counters
    side = [json.loads(ln) for ln in open(str(tmp_path / f"naive_results_{ds}.txt") + ".json").read().splitlines()]
    assert len(side) == len(lines)
    for ln, js in zip(lines, side):
        tok = ln.split()
        assert js["graph_type"] == tok[1] and js["n_q"] == c.nq and js["repeats"] == 2
        assert abs(js["recall_at_1"] - float(tok[3])) < 1e-5 and int(js["mean_hops"]) == int(tok[5])
        assert js["queries_per_s"] > 0 and js["devices"] == [0]
    # missing KL files (naive_test.cpp:77-88): both long-link graphs are built -- L = kl_size, sqrt(n) candidates,
    # the generator seeded like the entry points' -- written, and used; with the compiled reference at hand the files
    # must equal its KLgraph::BuildByNumberCustom for that seed (one thread), link for link
    os.remove(models / f"{ds}_kl_sqrt_style.ivecs")
    os.remove(models / f"{ds}_kl_llow_sqrt_style.ivecs")
    p = subprocess.run([nbin, ds, str(data), str(models), str(tmp_path), str(params)], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    sq = int(c.n ** 0.5)  # pow(n, 0.5) converted to size_t
    for fn, vecs in ((f"{ds}_kl_sqrt_style.ivecs", c.base), (f"{ds}_kl_llow_sqrt_style.ivecs", db_low)):
        raw = np.fromfile(models / fn, np.uint32).reshape(c.n, 6)
        assert (raw[:, 0] == 5).all() and (raw[:, 1:] < c.n).all()
        assert all(len(set(r[1:])) == 5 and i not in r[1:] for i, r in enumerate(raw))
        if orc_mod.have_ref():
            ref = orc_mod.Ref()
            _, want = ref.kl_build(1, 5, vecs, sq, meta["naive_seed"])
            assert np.array_equal(raw[:, 1:].reshape(-1), want), fn
    assert len(open(tmp_path / f"naive_results_{ds}.txt").read().splitlines()) == len(lines)


def test_prepare_graph_builds_missing_knn_on_device(tmp_path, orc):
    """prepare_graph without a kNN file: exact K-NN lists are built on the device first (gbnns_exact_knn), written
    in the reference's edge format, then pruned -- both files equal what the CPU restatements give."""
    pbin = os.path.join(ROOT, "gbnns_dim_red_amd", "search", "prepare_graph")
    gd = gu.load("tail_toy")
    c = gd.case
    db_low = orc.project(c.net, c.base)
    write_xvecs(tmp_path / "toy_base_lat.fvecs", db_low)
    (tmp_path / "params.txt").write_text(f"toy n {c.n}\ntoy d_low {c.dlow}\n")
    K, M = 24, 12
    env = dict(os.environ, GBNNS_GD_M=str(M), GBNNS_KNN_K=str(K))
    p = subprocess.run([pbin, "toy", "lat", str(tmp_path), str(tmp_path), str(tmp_path / "params.txt")],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "building exact 24-NN lists on the device" in p.stdout
    want_knn, _ = orc.exact_knn(db_low, db_low, K, 0, self_offset=0, threads=8)
    raw = np.fromfile(tmp_path / "toy_knn_1k_lat.ivecs", np.uint32).reshape(c.n, K + 1)
    assert (raw[:, 0] == K).all() and np.array_equal(raw[:, 1:], want_knn)
    koff = np.arange(c.n + 1, dtype=np.uint64) * np.uint64(K)
    off, nbr = orc.hnswlike_gd(koff, want_knn.reshape(-1), db_low, M, reverse=True, threads=4)
    got = np.fromfile(tmp_path / "toy_gd_knn_lat.ivecs", np.uint32)
    want = np.concatenate([np.concatenate([[int(off[i + 1] - off[i])], nbr[int(off[i]):int(off[i + 1])]])
                           for i in range(c.n)]).astype(np.uint32)
    assert np.array_equal(got, want)


@pytest.fixture(scope="module")
def units_exe(tmp_path_factory):
    """tests/cpp/dropin_units.cpp built against the drop-in headers and the product library (as test_dropin_units.py)."""
    libdir = os.path.join(ROOT, "gbnns_dim_red_amd", "lib")
    exe = str(tmp_path_factory.mktemp("units_gpu") / "dropin_units")
    subprocess.check_call(["g++", "-O2", "-std=c++11", "-ffp-contract=off", "-fno-fast-math", "-w", "-o", exe,
                           os.path.join(ROOT, "tests", "cpp", "dropin_units.cpp"), "-L" + libdir, "-lgbnns_hip",
                           "-Wl,-rpath," + libdir, "-lpthread"])
    return exe


def _score(ans, truth, base):
    # search_function.h:391-400: hit on GT[0], or on GT[1] when GT[0] and GT[1] are exact duplicates
    t0, t1 = truth[:, 0], truth[:, 1]
    dup = (((base[t0] - base[t1]) ** 2).sum(1) == 0) & (t0 != t1)
    return int(((ans == t0) | (dup & (ans == t1))).sum())


def _line_fields(ln):
    tok = ln.split()
    assert tok[0] == "graph_type" and tok[2] == "acc" and tok[4] == "hops" and tok[6] == "dist_calc" and tok[8] == "work_time"
    return tok[1], float(tok[3]), int(tok[5]), int(tok[7]), float(tok[9])


def test_perform_net_test_without_rerank(tmp_path, orc, units_exe):
    """performNetTest's `recheck_size <= 0` branch (search_function.h:363-372): project the query, walk the low-dim
    graph with (ef, k), answer = top of the trimmed heap, NO re-rank, nothing added to dist_calc.  Expectation: the
    oracle's plain walk over the projected queries in the low-dim space; and, as a cross-check, recheck_size = ef
    (the two-stage branch) through the same driver against the oracle's two-stage answers."""
    gd = gu.load("sift_toy")
    c = gd.case
    off, nbr = gd.graph
    db_low = orc.project(c.net, c.base, threads=4)
    q_low = orc.project(c.net, c.queries)
    truth = gd["truth"]
    write_xvecs(tmp_path / "base.fvecs", c.base)
    write_xvecs(tmp_path / "query.fvecs", c.queries)
    write_xvecs(tmp_path / "truth.ivecs", truth)
    write_xvecs(tmp_path / "base_low.fvecs", db_low)
    write_edges(tmp_path / "graph.ivecs", off, nbr)
    for i, layer in enumerate(c.net, 1):
        write_xvecs(tmp_path / f"net_{i}.fvecs", layer)
    for ef, recheck in ((8, -1), (64, -1), (64, 0), (20, 20)):
        out = tmp_path / f"res_{ef}_{recheck}.txt"
        p = subprocess.run([units_exe, "nettest", str(tmp_path), str(c.n), str(c.nq), str(truth.shape[1]), str(c.d), str(c.dlow),
                            str(c.dh), str(ef), str(recheck), str(out)], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stdout + p.stderr
        name, acc, hops, dc, wt = _line_fields(open(out).read().splitlines()[-1])
        if recheck > 0:
            e = orc.search_batch(orc_mod.MODE_NET, c.queries, c.base, off, nbr, recheck, db_low=db_low, net=c.net, threads=4)
        else:
            e = orc.search_batch(orc_mod.MODE_PLAIN, q_low, db_low, off, nbr, ef, k=1, threads=4)
        assert name == "hnsw_unit" and wt > 0
        assert abs(acc - _score(e["ids"].astype(np.int64), truth.astype(np.int64), c.base) / c.nq) < 2e-6, (ef, recheck)
        assert hops == int(e["hops"].astype(np.int64).sum()) // c.nq, (ef, recheck)
        assert dc == int(e["dist_calc"].astype(np.int64).sum()) // c.nq, (ef, recheck)


def test_perform_synthetic_tests_smoke(tmp_path, orc, units_exe):
    """performSyntheticTests (search_function.h:214-287; no caller in the reference): d = 5 beam-search sweep over
    ef 7 ... 30 on the kNN graph cut to its average degree, every query entering at node 0 -- six result lines, each
    equal to the oracle's plain walk on the same cut graph."""
    c = datagen.Case("syn", 6100, 3000, 200, 5, 4, 8)
    K = 12
    knn, _ = orc.exact_knn(c.base, c.base, K, 0, self_offset=0, threads=8)
    truth, _ = orc.exact_knn(c.base, c.queries, 2, 0, threads=8)
    koff = np.arange(c.n + 1, dtype=np.uint64) * np.uint64(K)
    write_xvecs(tmp_path / "base.fvecs", c.base)
    write_xvecs(tmp_path / "query.fvecs", c.queries)
    write_xvecs(tmp_path / "truth.ivecs", truth)
    write_edges(tmp_path / "knn.ivecs", koff, knn.reshape(-1))
    # the graph the sweep walks: cutKNNbyK at the average degree (pinned on the reference by test_dropin_units.py)
    subprocess.check_call([units_exe, "cutk", str(tmp_path / "knn.ivecs"), str(tmp_path / "base.fvecs"), str(c.n), str(c.d),
                           str(K), str(tmp_path / "cut.ivecs")])
    raw = np.fromfile(tmp_path / "cut.ivecs", np.uint32).reshape(c.n, K + 1)
    assert (raw[:, 0] == K).all()
    cut = raw[:, 1:].reshape(-1).copy()
    out = tmp_path / "syn.txt"
    p = subprocess.run([units_exe, "synthetic", str(tmp_path), str(c.n), str(c.nq), "2", str(c.d), str(out)],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    lines = open(out).read().splitlines()
    assert len(lines) == 6
    for ln, ef in zip(lines, (7, 10, 15, 22, 25, 30)):
        name, acc, hops, dc, wt = _line_fields(ln)
        e = orc.search_batch(orc_mod.MODE_PLAIN, c.queries, c.base, koff, cut, ef, k=1, threads=4)
        assert name == "knn_synth" and wt > 0
        assert abs(acc - _score(e["ids"].astype(np.int64), truth.astype(np.int64), c.base) / c.nq) < 2e-6, ef
        assert hops == int(e["hops"].astype(np.int64).sum()) // c.nq and dc == int(e["dist_calc"].astype(np.int64).sum()) // c.nq, ef
