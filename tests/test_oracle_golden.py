"""CPU tests: the oracle restatement (oracle/gbnns_oracle.cpp) against the golden vectors captured
from the compiled reference, plus live oracle-vs-reference checks where oracle/_ref is present."""
import numpy as np
import pytest

import datagen
import golden_util as gu
import oracle as orc_mod


def test_kats(orc):
    z = np.load(gu.GOLDEN_DIR + "/kats.npz")
    pairs = datagen.kat_pairs()
    assert list(z["dims"]) == datagen.KAT_DIMS
    l2 = np.array([gu.bits(orc.l2(a, b)) for a, b in pairs]).reshape(-1)
    nd = np.array([gu.bits(orc.negdot(a, b)) for a, b in pairs]).reshape(-1)
    assert np.array_equal(l2, z["l2_bits"])
    assert np.array_equal(nd, z["negdot_bits"])


def test_l2_ignores_tail_and_negdot_masks(orc):
    a = np.arange(1, 8, dtype=np.float32)
    b = np.zeros(7, np.float32)
    assert orc.l2(a, b) == np.float32(1 + 4 + 9 + 16)            # dims 4..6 dropped
    assert orc.negdot(a, a) == -np.float32(sum(i * i for i in range(1, 8)))
    assert orc.l2(a[:3], b[:3]) == 0.0


@pytest.mark.parametrize("name", gu.CASE_NAMES)
def test_inputs_regenerate_bit_exact(name):
    g = gu.load(name)
    assert g.case.input_hash() == g.meta["input_hash"]


@pytest.mark.parametrize("name", gu.CASE_NAMES)
def test_project_bits(orc, name):
    g = gu.load(name)
    q_low = orc.project(g.case.net, g.case.queries)
    assert np.array_equal(gu.bits(q_low), g["q_low_bits"])
    db_low = orc.project(g.case.net, g.case.base, threads=4)
    assert datagen.sha(db_low) == g.meta["db_low_sha"]


@pytest.mark.parametrize("name", gu.CASE_NAMES)
def test_walk_and_search(orc, name):
    g = gu.load(name)
    c = g.case
    off, nbr = g.graph
    db_low = orc.project(c.net, c.base, threads=4)
    q_low = orc.project(c.net, c.queries)
    for ef in g.efs:
        w = orc.walk(q_low, db_low, off, nbr, ef, metric=g.metric, threads=2)
        assert np.array_equal(w["ids"], g[f"walk_ids_{ef}"])
        assert np.array_equal(gu.bits(w["dists"]), g[f"walk_dist_bits_{ef}"])
        assert np.array_equal(w["count"], g[f"walk_count_{ef}"])
        assert np.array_equal(w["hops"], g[f"walk_hops_{ef}"])
        assert np.array_equal(w["dist_calc"], g[f"walk_dc_{ef}"])
        we = orc.walk(q_low, db_low, off, nbr, ef, entries=g["entries"], metric=g.metric)
        assert np.array_equal(we["ids"], g[f"walk_e_ids_{ef}"])
        assert np.array_equal(we["hops"], g[f"walk_e_hops_{ef}"])
        assert np.array_equal(we["dist_calc"], g[f"walk_e_dc_{ef}"])
        r = orc.rerank(c.queries, w["ids"], w["count"], c.base, metric=g.metric)
        assert np.array_equal(r, g[f"net_ans_{ef}"])
        s = orc.search_batch(orc_mod.MODE_NET, c.queries, c.base, off, nbr, ef, db_low=db_low,
                             net=c.net, metric=g.metric, threads=2)
        assert np.array_equal(s["ids"], g[f"net_ans_{ef}"])
        assert np.array_equal(s["hops"], g[f"walk_hops_{ef}"])
        assert np.array_equal(s["dist_calc"], g[f"walk_dc_{ef}"] + ef)
        s = orc.search_batch(orc_mod.MODE_LOWQ, c.queries, c.base, off, nbr, ef, db_low=db_low,
                             q_low=q_low, metric=g.metric)
        assert np.array_equal(s["ids"], g[f"net_ans_{ef}"])
        p = orc.search_batch(orc_mod.MODE_PLAIN, c.queries, c.base, off, nbr, ef, k=1,
                             metric=g.metric)
        assert np.array_equal(p["ids"], g[f"plain_ans_{ef}"])
        assert np.array_equal(p["hops"], g[f"plain_hops_{ef}"])
        assert np.array_equal(p["dist_calc"], g[f"plain_dc_{ef}"])


def test_tie_heavy_lattice_walks(orc):
    g = gu.load("ties_toy")
    c = g.case
    graphs = dict(gd=g.graph, rnd=(g["rgraph_off"], g["rgraph_nbr"]))
    for ef in g.efs:
        for tag, (off, nbr) in graphs.items():
            w = orc.walk(c.queries, c.base, off, nbr, ef, entries=g["entries"])
            assert np.array_equal(w["ids"], g[f"lat_{tag}_ids_{ef}"])
            assert np.array_equal(gu.bits(w["dists"]), g[f"lat_{tag}_dist_bits_{ef}"])
            assert np.array_equal(w["hops"], g[f"lat_{tag}_hops_{ef}"])
            assert np.array_equal(w["dist_calc"], g[f"lat_{tag}_dc_{ef}"])
    # the fixture really is tie-heavy: many equal distances inside the result lists
    d = g["lat_rnd_dist_bits_64"]
    assert (np.diff(np.sort(d, axis=1), axis=1) == 0).mean() > 0.5


def _aux_fixture():
    import json
    z = np.load(gu.GOLDEN_DIR + "/aux_toy.npz")
    return z, json.loads(bytes(z["meta"]).decode())


def test_auxiliary_graph_walks(orc):
    """use_second_graph walks (search_function.h:73-89) against the compiled reference's outputs."""
    z, meta = _aux_fixture()
    for name, info in meta["cases"].items():
        g = gu.load(name)
        c = g.case
        off, nbr = g.graph
        aux = (z[f"{name}_aux_off"], z[f"{name}_aux_nbr"])
        if name == "ties_toy":
            q, db = c.queries, c.base
        else:
            db = orc.project(c.net, c.base, threads=4)
            q = orc.project(c.net, c.queries)
        for ef in info["efs"]:
            for llf, hb in meta["variants"]:
                tag = f"{name}_{ef}_{llf}_{hb}"
                w = orc.walk(q, db, off, nbr, ef, entries=g["entries"], metric=c.metric, aux=aux,
                             llf=bool(llf), hops_bound=hb, threads=2)
                assert np.array_equal(w["ids"], z[f"walk_ids_{tag}"]), tag
                assert np.array_equal(gu.bits(w["dists"]), z[f"walk_dist_bits_{tag}"]), tag
                assert np.array_equal(w["hops"], z[f"walk_hops_{tag}"]), tag
                assert np.array_equal(w["dist_calc"], z[f"walk_dc_{tag}"]), tag
                if name == "sift_toy":
                    s = orc.search_batch(orc_mod.MODE_NET, c.queries, c.base, off, nbr, ef, db_low=db,
                                         net=c.net, entries=g["entries"], metric=c.metric, aux=aux,
                                         llf=bool(llf), hops_bound=hb, threads=2)
                    assert np.array_equal(s["ids"], z[f"net_ans_{tag}"]), tag
                p = orc.search_batch(orc_mod.MODE_PLAIN, c.queries, c.base, off, nbr, ef, k=1,
                                     entries=g["entries"], metric=c.metric, aux=aux, llf=bool(llf),
                                     hops_bound=hb, threads=2)
                assert np.array_equal(p["ids"], z[f"plain_ans_{tag}"]), tag
                assert np.array_equal(p["hops"], z[f"plain_hops_{tag}"]), tag
                assert np.array_equal(p["dist_calc"], z[f"plain_dc_{tag}"]), tag


def _knn_cases():
    import json
    z = np.load(gu.GOLDEN_DIR + "/knn_toy.npz")
    for name, metric, space in json.loads(bytes(z["meta"]).decode())["cases"]:
        yield name, metric, space, z[f"truth_{name}_{metric}_{space}"]


def test_exact_knn_restatement_vs_get_truth(orc):
    """k = 1 of the brute-force restatement is the reference's getTruth (support_func.h:270-290); k > 1 extends
    it by the same (distance, id) order."""
    for name, metric, space, truth in _knn_cases():
        c = gu.load(name).case
        base, q = (orc.project(c.net, c.base, threads=4), orc.project(c.net, c.queries)) if space == "low" \
            else (c.base, c.queries)
        ids1, _ = orc.exact_knn(base, q, 1, metric)
        assert np.array_equal(ids1[:, 0], truth), (name, metric, space)
        ids, dist = orc.exact_knn(base, q, 7, metric)
        assert np.array_equal(ids[:, 0], truth)
        # ascending (distance, id) pairs
        assert (np.diff(dist, axis=1) >= 0).all()
        tie = np.diff(dist, axis=1) == 0
        assert (np.diff(ids.astype(np.int64), axis=1)[tie] > 0).all()
    # a set against itself: the row itself is left out
    c = gu.load("tail_toy").case
    ids, _ = orc.exact_knn(c.base, c.base[100:164], 5, 0, self_offset=100)
    assert not (ids == (np.arange(64) + 100)[:, None]).any()
    ids0, _ = orc.exact_knn(c.base, c.base[100:164], 1, 0)
    assert np.array_equal(ids0[:, 0], np.arange(64) + 100)  # distinct points: nearest is itself


def test_graph_builder_restatement(orc):
    g = gu.load("tail_toy")
    c = g.case
    db_low = orc.project(c.net, c.base)
    koff, knbr = datagen.dense_to_csr(g["knn"])
    for threads in (1, 4):
        off, nbr = orc.hnswlike_gd(koff, knbr, db_low, g.meta["gd_M"], reverse=True,
                                   threads=threads)
        assert np.array_equal(off, g["graph_off"])
        assert np.array_equal(nbr, g["graph_nbr"])


# ---- live cross-checks against the compiled reference (only where oracle/_ref was built) ------

@pytest.mark.parametrize("seed", [1, 2, 3])
def test_oracle_vs_ref_random(orc, ref, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    n, nq = 3000, 200
    d, dlow, dh = [(64, 16, 32), (100, 24, 48), (37, 10, 21)][seed - 1]
    c = datagen.Case("x", 1000 + seed, n, nq, d, dlow, dh)
    assert np.array_equal(gu.bits(orc.project(c.net, c.queries)),
                          gu.bits(ref.project(c.net, c.queries)))
    db_low = orc.project(c.net, c.base, threads=4)
    off, nbr = datagen.random_graph(rng, n, 2, 40)
    q_low = orc.project(c.net, c.queries)
    for metric in (0, 1):
        for ef in (1, 5, 50, 300):
            a = orc.walk(q_low, db_low, off, nbr, ef, metric=metric, threads=2)
            b = ref.walk(q_low, db_low, off, nbr, ef, metric=metric, threads=2)
            for k in ("ids", "count", "hops", "dist_calc"):
                assert np.array_equal(a[k], b[k]), (metric, ef, k)
            assert np.array_equal(gu.bits(a["dists"]), gu.bits(b["dists"]))
            sa = orc.search_batch(orc_mod.MODE_NET, c.queries, c.base, off, nbr, ef,
                                  db_low=db_low, net=c.net, metric=metric)
            sb = ref.search_batch(orc_mod.MODE_NET, c.queries, c.base, off, nbr, ef,
                                  db_low=db_low, net=c.net, metric=metric)
            for k in ("ids", "hops", "dist_calc"):
                assert np.array_equal(sa[k], sb[k])
    # auxiliary graph (use_second_graph), both llf settings, several hop bounds
    aux = datagen.random_graph(rng, n, 0, 5)
    for metric in (0, 1):
        for ef, llf, hb in ((1, True, 50), (10, True, 2), (10, False, 50), (120, True, 50)):
            a = orc.walk(q_low, db_low, off, nbr, ef, metric=metric, aux=aux, llf=llf, hops_bound=hb)
            b = ref.walk(q_low, db_low, off, nbr, ef, metric=metric, aux=aux, llf=llf, hops_bound=hb)
            for k in ("ids", "count", "hops", "dist_calc"):
                assert np.array_equal(a[k], b[k]), (metric, ef, llf, hb, k)
    # multi entry points and k < ef
    ent = rng.integers(0, n, size=(nq, 3)).astype(np.uint32)
    a = orc.walk(q_low, db_low, off, nbr, 20, k=7, entries=ent)
    b = ref.walk(q_low, db_low, off, nbr, 20, k=7, entries=ent)
    for k in ("ids", "count", "hops", "dist_calc"):
        assert np.array_equal(a[k], b[k])


def test_builder_vs_ref_random(orc, ref):
    c = datagen.Case("b", 77, 1500, 8, 32, 12, 16)
    db_low = orc.project(c.net, c.base, threads=4)
    knn = datagen.knn_bruteforce(db_low, 20)
    koff, knbr = datagen.dense_to_csr(knn)
    for M, rev in ((8, True), (14, False), (5, True)):
        a = orc.hnswlike_gd(koff, knbr, db_low, M, reverse=rev, threads=3)
        b = ref.hnswlike_gd(koff, knbr, db_low, M, reverse=rev, threads=1)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
