#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the COMPILED REFERENCE (oracle/_ref/libgbnns_ref.so).

Run in the build container only (needs /root/reference to have built oracle/_ref):

    python tests/golden/make_golden.py

Inputs are regenerated from seeds by tests/datagen.py (bit-portable integer recipe) and are NOT
stored; the fixtures hold the reference's OUTPUTS (ids, distance bit patterns, hops, dist_calc,
projected-query bit patterns, result lines), the graphs those outputs were produced on, and
sha256 digests of the regenerated inputs / large intermediates.  The fixtures are data only: no
reference source text is stored.
"""
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))            # tests/
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))  # repo root

import datagen  # noqa: E402
import oracle  # noqa: E402


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def ground_truth(base, queries, k):
    b = base.astype(np.float64)
    q = queries.astype(np.float64)
    dm = (q * q).sum(1)[:, None] + (b * b).sum(1)[None, :] - 2.0 * q @ b.T
    return np.argsort(dm, axis=1, kind="stable")[:, :k].astype(np.uint32)


def make_case(ref, spec):
    spec = dict(spec)
    efs = spec.pop("efs")
    c = datagen.Case(**spec)
    out = {}
    meta = dict(spec, efs=efs, input_hash=c.input_hash())

    db_low = ref.project(c.net, c.base)
    q_low = ref.project(c.net, c.queries)
    assert np.isfinite(db_low).all() and np.isfinite(q_low).all()
    out["q_low_bits"] = bits(q_low)
    meta["db_low_sha"] = datagen.sha(db_low)

    K = 24 if c.n <= 2048 else 32
    M = 12
    knn = datagen.knn_bruteforce(db_low, K)
    koff, knbr = datagen.dense_to_csr(knn)
    # The builder needs a positive metric (it drops candidates with dist <= 1e-10,
    # support_func.h:535), so graphs are always built with L2 -- as prepare_graph.cpp:70 does.
    goff, gnbr = ref.hnswlike_gd(koff, knbr, db_low, M, metric=0, reverse=True, threads=1)
    # multi-threaded reference build must give the same graph (the parallel part is per node)
    goff2, gnbr2 = ref.hnswlike_gd(koff, knbr, db_low, M, metric=0, reverse=True, threads=4)
    assert np.array_equal(goff, goff2) and np.array_equal(gnbr, gnbr2)
    out["graph_off"], out["graph_nbr"] = goff, gnbr
    meta["gd_M"], meta["gd_K"] = M, K
    if c.n <= 1024:
        out["knn"] = knn  # lets the builder restatement be checked against the reference's output

    rng = np.random.Generator(np.random.PCG64(c.seed + 7))
    entries = rng.integers(0, c.n, size=c.nq, dtype=np.int64).astype(np.uint32)
    out["entries"] = entries

    for ef in efs:
        w = ref.walk(q_low, db_low, goff, gnbr, ef, metric=c.metric)
        out[f"walk_ids_{ef}"] = w["ids"]
        out[f"walk_dist_bits_{ef}"] = bits(w["dists"])
        out[f"walk_count_{ef}"] = w["count"]
        out[f"walk_hops_{ef}"] = w["hops"]
        out[f"walk_dc_{ef}"] = w["dist_calc"]
        s = ref.search_batch(oracle.MODE_NET, c.queries, c.base, goff, gnbr, ef, db_low=db_low,
                             net=c.net, metric=c.metric)
        out[f"net_ans_{ef}"] = s["ids"]
        assert np.array_equal(s["hops"], w["hops"])
        assert np.array_equal(s["dist_calc"], w["dist_calc"] + ef)
        s1 = ref.search_batch(oracle.MODE_LOWQ, c.queries, c.base, goff, gnbr, ef, db_low=db_low,
                              q_low=q_low, metric=c.metric)
        assert np.array_equal(s1["ids"], s["ids"])
        p = ref.search_batch(oracle.MODE_PLAIN, c.queries, c.base, goff, gnbr, ef, k=1,
                             metric=c.metric)
        out[f"plain_ans_{ef}"] = p["ids"]
        out[f"plain_hops_{ef}"] = p["hops"]
        out[f"plain_dc_{ef}"] = p["dist_calc"]
        we = ref.walk(q_low, db_low, goff, gnbr, ef, entries=entries, metric=c.metric)
        out[f"walk_e_ids_{ef}"] = we["ids"]
        out[f"walk_e_hops_{ef}"] = we["hops"]
        out[f"walk_e_dc_{ef}"] = we["dist_calc"]

    if c.kind == "lattice":
        # tie-heavy walks directly on the integer lattice (distances are small exact integers)
        roff, rnbr = datagen.random_graph(rng, c.n, 4, 20)
        out["rgraph_off"], out["rgraph_nbr"] = roff, rnbr
        for ef in efs:
            for tag, (o, nb) in (("gd", (goff, gnbr)), ("rnd", (roff, rnbr))):
                w = ref.walk(c.queries, c.base, o, nb, ef, entries=entries, metric=c.metric)
                out[f"lat_{tag}_ids_{ef}"] = w["ids"]
                out[f"lat_{tag}_dist_bits_{ef}"] = bits(w["dists"])
                out[f"lat_{tag}_hops_{ef}"] = w["hops"]
                out[f"lat_{tag}_dc_{ef}"] = w["dist_calc"]

    # The reference harness end to end: result lines of performRealNetTests / performRealTests.
    truth = ground_truth(c.base, c.queries, 4)
    out["truth"] = truth
    if c.metric == 0:
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "res.txt")
            ref.perform_real_net_tests(c.base, c.queries, db_low, c.net, goff, gnbr, truth, efs,
                                       path, graph_name="hnsw_new_ar", number_exper=2, threads=1)
            ref.perform_real_tests(c.base, c.queries, c.base, c.queries, goff, gnbr, truth, efs,
                                   path, graph_name="hnsw", number_exper=2, threads=1)
            lines = open(path).read().splitlines()
        # work_time differs run to run: keep everything before it
        meta["result_lines"] = [ln.split(" work_time ")[0] for ln in lines]

    out["meta"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
    return out


def make_kats(ref):
    pairs = datagen.kat_pairs()
    return dict(dims=np.array(datagen.KAT_DIMS, np.int32),
                l2_bits=np.array([bits(ref.l2(a, b)) for a, b in pairs], np.uint32).reshape(-1),
                negdot_bits=np.array([bits(ref.negdot(a, b)) for a, b in pairs],
                                     np.uint32).reshape(-1))


def main():
    oracle.build()
    ref = oracle.Ref()
    np.savez_compressed(os.path.join(HERE, "kats.npz"), **make_kats(ref))
    for spec in datagen.GOLDEN_CASES:
        print("golden:", spec["name"], flush=True)
        rec = make_case(ref, spec)
        np.savez_compressed(os.path.join(HERE, spec["name"] + ".npz"), **rec)
    ref.close()


if __name__ == "__main__":
    main()
