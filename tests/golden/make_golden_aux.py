#!/usr/bin/env python3
"""Generate tests/golden/aux_toy.npz from the COMPILED REFERENCE: walks with an auxiliary graph
(use_second_graph = true, search_function.h:73-89 -- what naive_test.cpp:103-105 runs with the KL graph).

    python tests/golden/make_golden_aux.py        (build container only: needs oracle/_ref)

Inputs are the regenerated `sift_toy` (clustered, L2) and `ties_toy` (integer lattice: tie-heavy) cases of
tests/datagen.py with the main graph stored in their own fixtures; the auxiliary graph (random long
links, 1..6 per node) is stored here together with the reference's outputs.  Data only.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import datagen  # noqa: E402
import golden_util as gu  # noqa: E402
import oracle  # noqa: E402

VARIANTS = [(1, 50), (0, 50), (1, 3), (0, 0)]  # (llf, hops_bound)


def main():
    oracle.build()
    ref = oracle.Ref()
    out = {}
    meta = {"variants": VARIANTS, "cases": {}}
    for name, efs in (("sift_toy", [1, 8, 64]), ("ties_toy", [2, 8, 64])):
        g = gu.load(name)
        c = g.case
        goff, gnbr = g.graph
        rng = np.random.Generator(np.random.PCG64(c.seed + 77))
        aoff, anbr = datagen.random_graph(rng, c.n, 1, 6)
        out[f"{name}_aux_off"], out[f"{name}_aux_nbr"] = aoff, anbr
        entries = g["entries"]
        meta["cases"][name] = {"efs": efs}
        if name == "ties_toy":   # walk directly on the lattice (exact small-integer distances)
            q, db = c.queries, c.base
        else:
            db = ref.project(c.net, c.base)
            q = ref.project(c.net, c.queries)
        for ef in efs:
            for llf, hb in VARIANTS:
                tag = f"{name}_{ef}_{llf}_{hb}"
                w = ref.walk(q, db, goff, gnbr, ef, entries=entries, metric=c.metric, aux=(aoff, anbr),
                             llf=bool(llf), hops_bound=hb)
                out[f"walk_ids_{tag}"] = w["ids"]
                out[f"walk_dist_bits_{tag}"] = gu.bits(w["dists"])
                out[f"walk_hops_{tag}"] = w["hops"]
                out[f"walk_dc_{tag}"] = w["dist_calc"]
                if name == "sift_toy":
                    s = ref.search_batch(oracle.MODE_NET, c.queries, c.base, goff, gnbr, ef, db_low=db,
                                         net=c.net, entries=entries, metric=c.metric, aux=(aoff, anbr),
                                         llf=bool(llf), hops_bound=hb)
                    out[f"net_ans_{tag}"] = s["ids"]
                    assert np.array_equal(s["hops"], w["hops"])
                p = ref.search_batch(oracle.MODE_PLAIN, c.queries, c.base, goff, gnbr, ef, k=1,
                                     entries=entries, metric=c.metric, aux=(aoff, anbr), llf=bool(llf),
                                     hops_bound=hb)
                out[f"plain_ans_{tag}"] = p["ids"]
                out[f"plain_hops_{tag}"] = p["hops"]
                out[f"plain_dc_{tag}"] = p["dist_calc"]
        if name == "sift_toy":
            # the reference's harness end to end, as naive_test.cpp:98-105 calls it (entry points from
            # mt19937(7) because the graph names do not start with "hnsw")
            import tempfile
            truth = g["truth"]
            with tempfile.TemporaryDirectory() as td:
                path = os.path.join(td, "res.txt")
                for gname, a, l in (("hnsw", None, False), ("knn", None, False), ("knn_lk", (aoff, anbr), True)):
                    ref.perform_real_tests(c.base, c.queries, c.base, c.queries, goff, gnbr, truth, efs, path,
                                           graph_name=gname, number_exper=2, threads=1, aux=a, llf=l, seed=7)
                ref.perform_real_tests(c.base, c.queries, db, q, goff, gnbr, truth, efs, path,
                                       graph_name="knn_lk_low", number_exper=2, threads=1, aux=(aoff, anbr),
                                       llf=True, seed=7)
                lines = open(path).read().splitlines()
            meta["naive_result_lines"] = [ln.split(" work_time ")[0] for ln in lines]
            meta["naive_seed"] = 7
        # hops_bound = 0 must equal the walk without an auxiliary graph
        w0 = ref.walk(q, db, goff, gnbr, efs[-1], entries=entries, metric=c.metric)
        assert np.array_equal(w0["ids"], out[f"walk_ids_{name}_{efs[-1]}_0_0"])
        # the auxiliary graph really changes the walk
        assert not np.array_equal(w0["hops"], out[f"walk_hops_{name}_{efs[-1]}_1_50"])
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "aux_toy.npz"), **out)
    ref.close()


if __name__ == "__main__":
    main()
