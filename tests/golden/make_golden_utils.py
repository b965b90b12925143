#!/usr/bin/env python3
"""Generate tests/golden/utils_toy.npz from the COMPILED REFERENCE: outputs of the host-side helpers that surround
the search path -- hnswlikeGD with need_const_degree = true (getConstantDegreeForGD, support_func.h:466-485),
cutKNNbyK / cutKNNbyThreshold / mergeGraph / fillGraphToConstantDegree (:292-340, :383-399, :489-518), the three
KLgraph builders (support_classes.h:38-175, one thread, seeded generator) and createUniformData (:252-267).
Data only (adjacency lists, vectors); inputs are regenerated from tests/datagen.py.
    python tests/golden/make_golden_utils.py   (build container only)"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import datagen  # noqa: E402
import oracle  # noqa: E402


def inputs():
    """Shared by this script and tests/test_dropin_units.py."""
    c = datagen.Case("u", 9100, 600, 4, 24, 8, 16)            # clustered vectors, d % 8 == 0
    t = datagen.Case("u", 9101, 400, 4, 10, 4, 8, kind="lattice")  # many exactly equal distances
    rng = np.random.Generator(np.random.PCG64(9102))
    knn = datagen.knn_bruteforce(c.base, 40)
    knn_t = datagen.knn_bruteforce(t.base, 30)
    # ragged second graph for merge / fill
    lists = [rng.integers(0, c.n, size=int(rng.integers(0, 25))).astype(np.uint32) for _ in range(c.n)]
    return c, t, knn, knn_t, lists


def main():
    oracle.build()
    ref = oracle.Ref()
    c, t, knn, knn_t, lists = inputs()
    out = {}
    koff, knbr = datagen.dense_to_csr(knn)
    toff, tnbr = datagen.dense_to_csr(knn_t)
    for name, (ko, kn, ds) in {"c": (koff, knbr, c.base), "t": (toff, tnbr, t.base)}.items():
        for M, rev in ((6, 1), (10, 0), (14, 1)):
            o, nb = ref.hnswlike_gd_const(ko, kn, ds, M, reverse=bool(rev))
            out[f"constdeg_{name}_{M}_{rev}_off"], out[f"constdeg_{name}_{M}_{rev}_nbr"] = o, nb
        for k in (1, 7, 25, 64):
            o, nb = ref.cut_knn_by_k(ko, kn, ds, k)
            out[f"cutk_{name}_{k}_off"], out[f"cutk_{name}_{k}_nbr"] = o, nb
    for thr in (0.05, 0.4, 2.0):
        o, nb = ref.cut_knn_by_threshold(koff, knbr, c.base, thr)
        out[f"cutthr_{thr}_off"], out[f"cutthr_{thr}_nbr"] = o, nb
    loff, lnbr = datagen.lists_to_csr(lists)
    g6 = (out["constdeg_c_6_1_off"], out["constdeg_c_6_1_nbr"])
    o, nb = ref.merge_graph(g6[0], g6[1], loff, lnbr)
    out["merge_off"], out["merge_nbr"] = o, nb
    for deg in (4, 16, 30):
        o, nb = ref.fill_const_degree(loff, lnbr, koff, knbr, deg)
        out[f"fill_{deg}_off"], out[f"fill_{deg}_nbr"] = o, nb
    for which, l, sq, seed in ((1, 5, 24, 7), (1, 15, 20, 12345), (0, 4, 0, 3), (2, 3, 0, 99)):
        ds = c.base if which != 0 else c.base[:200]
        o, nb = ref.kl_build(which, l, ds, sq, seed)
        out[f"kl_{which}_{l}_{sq}_{seed}_off"], out[f"kl_{which}_{l}_{sq}_{seed}_nbr"] = o, nb
    for n, d, seed in ((50, 3, 1), (40, 17, 2)):
        out[f"uniform_{n}_{d}_{seed}"] = ref.create_uniform_data(n, d, seed).view(np.uint32)
    np.savez_compressed(os.path.join(HERE, "utils_toy.npz"), **out)
    print("wrote utils_toy.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
