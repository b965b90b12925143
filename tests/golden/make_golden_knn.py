#!/usr/bin/env python3
"""Generate tests/golden/knn_toy.npz from the COMPILED REFERENCE: getTruth (support_func.h:270-290) outputs
on regenerated inputs -- original-space and low-dim sets, a d % 4 != 0 case, the tie-heavy lattice with both
metrics.  Data only (ids).      python tests/golden/make_golden_knn.py   (build container only)"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import golden_util as gu  # noqa: E402
import oracle  # noqa: E402

CASES = [("sift_toy", 0, "orig"), ("sift_toy", 0, "low"), ("tail_toy", 0, "orig"), ("tail_toy", 0, "low"),
         ("ties_toy", 0, "orig"), ("ties_toy", 1, "orig"), ("deep_toy", 0, "orig"), ("glove_toy", 1, "low")]


def main():
    oracle.build()
    ref = oracle.Ref()
    out = {}
    for name, metric, space in CASES:
        c = gu.load(name).case
        if space == "low":
            base, q = ref.project(c.net, c.base), ref.project(c.net, c.queries)
        else:
            base, q = c.base, c.queries
        out[f"truth_{name}_{metric}_{space}"] = ref.get_truth(base, q, metric)
    out["meta"] = np.frombuffer(json.dumps({"cases": CASES}).encode(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "knn_toy.npz"), **out)
    ref.close()


if __name__ == "__main__":
    main()
