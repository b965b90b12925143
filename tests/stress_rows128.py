#!/usr/bin/env python3
"""Stress (GPU box; lives under tests/ because it uses the oracle as the checker, but it is not collected by
pytest): many seeded random configurations with 128-byte walked rows against the oracle -- two-stage NET mode
(fused re-rank) and PLAIN walks, both metrics, auxiliary graphs.
usage: python tests/stress_rows128.py [seed] [cases] [only_case] [any]
With a fourth argument "any" the walked rows take other lengths too (64 .. 400 bytes: the 12- / 16-step and generic
distance instantiations)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gbnns_dim_red_amd as g, oracle, datagen, golden_util as gu
g.load_library()
orc = oracle.Oracle()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1  # replay the generator, run just this case, print details
any_rows = len(sys.argv) > 4 and sys.argv[4] == "any"
rng = np.random.Generator(np.random.PCG64(seed))
bad = 0
for case in range(cases):
    n = int(rng.integers(200, 8000)); nq = int(rng.integers(4, 100))
    kind = "lattice" if rng.integers(0, 3) == 0 else "clustered"
    metric = int(rng.integers(0, 2))
    net_mode = bool(rng.integers(0, 2))
    d = int(rng.choice([32, 64, 96, 128])) if net_mode else 32
    d_low = 32
    if any_rows:
        walked = int(rng.choice([16, 48, 64, 100, 32]))
        if net_mode:
            d, d_low = max(d, walked), walked
        else:
            d = walked
    c = datagen.Case("s", 9000 + seed * 1000 + case, n, nq, d, d_low if net_mode else 4, int(rng.choice([16, 40])), kind=kind)
    deg_hi = int(rng.choice([5, 17, 32, 33, 64, 70]))
    off, nbr = datagen.random_graph(rng, n, 0, min(deg_hi, n - 1))
    ent = rng.integers(0, n, size=nq).astype(np.uint32)
    ef = int(rng.choice([1, 3, 16, 40, 64, 65, 90, 128, 130, 256, 257, 400, 513, 700, 1024, 1100]))
    cap = int(rng.choice([0, 0, 0, 128, 512]))
    # a third of the cases walk with an auxiliary graph (use_second_graph), random llf / hops_bound
    use_aux = rng.integers(0, 3) == 0
    aux = datagen.random_graph(rng, n, 0, min(int(rng.choice([3, 40, 70])), n - 1)) if use_aux else None
    llf = bool(rng.integers(0, 2))
    hb = int(rng.choice([0, 2, 50, 100000]))
    okw = dict(aux=aux, llf=llf, hops_bound=hb) if use_aux else {}
    gkw = dict(aux=True, llf=llf, hops_bound=hb) if use_aux else {}
    if rng.integers(0, 8) == 0:   # an eighth of the cases take the HBM-bitmap first pass
        gkw["flags"] = g.FLAG_BITMAP_PASS
    # a sixth of the plain walks start from several entry points (general kernel)
    m_ent = int(rng.choice([1, 1, 1, 1, 1, 2, 3]))
    if m_ent > 1 and not net_mode:
        ent = rng.integers(0, n, size=(nq, m_ent)).astype(np.uint32)
    tag = (case, n, nq, kind, metric, net_mode, d, deg_hi, ef, cap, use_aux, llf, hb, m_ent)
    if only >= 0 and case != only:
        continue
    try:
        if net_mode:
            db_low = orc.project(c.net, c.base)
            if not np.isfinite(db_low).all():
                continue
            ix = g.Index(c.base, off, nbr, db_low=db_low, net=c.net, metric=metric)
            if use_aux:
                ix.set_aux_graph(*aux)
            s = orc.search_batch(oracle.MODE_NET, c.queries, c.base, off, nbr, ef, db_low=db_low, net=c.net, entries=ent, metric=metric, **okw)
            if not np.isfinite(orc.project(c.net, c.queries)).all():
                ix.close(); continue
            r = ix.search(c.queries, ef, entry_ids=ent, want=("hops", "dist_calc"), hash_capacity=cap, **gkw)
            ok = np.array_equal(r["ids"], s["ids"]) and np.array_equal(r["hops"], s["hops"]) and np.array_equal(r["dist_calc"] + ef, s["dist_calc"])
        else:
            ix = g.Index(c.base, off, nbr, metric=metric)
            if use_aux:
                ix.set_aux_graph(*aux)
            w = orc.walk(c.queries, c.base, off, nbr, ef, entries=ent, metric=metric, **okw)
            r = ix.search(c.queries, ef, mode=g.MODE_PLAIN, k=ef, entry_ids=ent, want=("hops", "dist_calc", "cand", "cand_dist"), hash_capacity=cap, **gkw)
            ok = (np.array_equal(r["cand"], w["ids"]) and np.array_equal(gu.bits(r["cand_dist"]), gu.bits(w["dists"]))
                  and np.array_equal(r["hops"], w["hops"]) and np.array_equal(r["dist_calc"], w["dist_calc"]))
            if only >= 0:
                bq = np.nonzero((r["cand"] != w["ids"]).any(1) | (r["hops"] != w["hops"]) | (r["dist_calc"] != w["dist_calc"]))[0]
                print("bad queries", bq[:10], "of", nq)
                for q in bq[:2]:
                    print(" q", q, "hops", r["hops"][q], w["hops"][q], "dc", r["dist_calc"][q], w["dist_calc"][q])
                    dif = np.nonzero(r["cand"][q] != w["ids"][q])[0]
                    print("  first diff positions", dif[:8], "gpu", r["cand"][q][dif[:8]], "ref", w["ids"][q][dif[:8]])
                    print("  gpu dist", r["cand_dist"][q][dif[:4]], "ref dist", w["dists"][q][dif[:4]])
                dd = np.nonzero(gu.bits(r["cand_dist"]) != gu.bits(w["dists"]))
                print("dist-bit diffs", len(dd[0]), [(hex(a), hex(b)) for a, b in zip(gu.bits(r["cand_dist"])[dd][:4], gu.bits(w["dists"])[dd][:4])])
        ix.close()
    except Exception as e:  # noqa: BLE001
        ok = False
        print("EXC", tag, e)
    if not ok:
        bad += 1
        print("MISMATCH", tag, flush=True)
print("seed", seed, "cases", cases, "bad", bad)
