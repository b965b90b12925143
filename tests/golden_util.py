"""Loading of the committed golden fixtures (tests/golden/*.npz)."""
import functools
import json
import os

import numpy as np

import datagen

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASE_NAMES = [c["name"] for c in datagen.GOLDEN_CASES]


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


class Golden:
    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.meta = json.loads(bytes(self.z["meta"]).decode())
        spec = {k: self.meta[k] for k in ("name", "seed", "n", "nq", "d", "dlow", "dh")}
        spec["kind"] = self.meta.get("kind", "clustered")
        spec["metric"] = self.meta.get("metric", 0)
        self.case = datagen.Case(**spec)
        self.efs = self.meta["efs"]
        self.metric = spec["metric"]
        self.graph = (self.z["graph_off"], self.z["graph_nbr"])

    def __getitem__(self, k):
        return self.z[k]

    def __contains__(self, k):
        return k in self.z.files


@functools.lru_cache(maxsize=None)
def load(name):
    return Golden(name)
