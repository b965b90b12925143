"""CPU test of the prepare_graph drop-in (host-only code path): knn lists + low-dim vectors in the
reference's file formats -> GD graph file, compared with the graph the compiled reference's
hnswlikeGD produced for the same input (golden fixture tail_toy, M = 12)."""
import os
import subprocess

import numpy as np

import datagen
import golden_util as gu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "gbnns_dim_red_amd", "search", "prepare_graph")


def _write_xvecs(path, a):
    a = np.ascontiguousarray(a)
    rec = np.empty((a.shape[0], a.shape[1] + 1), np.uint32)
    rec[:, 0] = a.shape[1]
    rec[:, 1:] = a.view(np.uint32)
    rec.tofile(path)


def _read_edges(path, n):
    raw = np.fromfile(path, np.uint32)
    off = [0]
    nbr = []
    p = 0
    for _ in range(n):
        k = int(raw[p])
        nbr.append(raw[p + 1:p + 1 + k])
        p += 1 + k
        off.append(off[-1] + k)
    assert p == raw.size
    return np.array(off, np.uint64), np.concatenate(nbr)


def test_prepare_graph_matches_reference_builder(tmp_path, orc):
    import gbnns_dim_red_amd as g
    g.build_library()
    assert os.path.exists(BIN)
    gd = gu.load("tail_toy")
    c = gd.case
    db_low = orc.project(c.net, c.base)
    assert datagen.sha(db_low) == gd.meta["db_low_sha"]
    _write_xvecs(tmp_path / "toy_base_lat.fvecs", db_low)
    knn = gd["knn"]
    with open(tmp_path / "toy_knn_1k_lat.ivecs", "wb") as f:  # edge-list format: [size][ids]
        for row in knn:
            np.array([len(row)], np.uint32).tofile(f)
            row.astype(np.uint32).tofile(f)
    (tmp_path / "params.txt").write_text(f"toy n {c.n}\ntoy d_low {c.dlow}\n")
    env = dict(os.environ, GBNNS_GD_M=str(gd.meta["gd_M"]))
    p = subprocess.run([BIN, "toy", "lat", str(tmp_path), str(tmp_path), str(tmp_path / "params.txt")],
                       env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    off, nbr = _read_edges(tmp_path / "toy_gd_knn_lat.ivecs", c.n)
    assert np.array_equal(off, gd["graph_off"])
    assert np.array_equal(nbr, gd["graph_nbr"])
    assert "GD_knn" in p.stdout and "knn_low" in p.stdout
