"""Multi-process (world_size 2 and 3, gloo, CPU) test of the query-sharding host logic: block
bounds, the padded all-gather of answer ids, and shard equivalence of a whole search -- the search
function here is the CPU oracle (a stand-in for one rank's GPU search; the gather logic is the
thing under test)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import datagen


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_q, tmpdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from gbnns_dim_red_amd import sharding
    orc = oracle.Oracle()
    c = datagen.Case("s", 31, 2000, n_q, 32, 8, 16)
    rng = np.random.Generator(np.random.PCG64(32))
    off, nbr = datagen.random_graph(rng, c.n, 4, 20)
    db_low = orc.project(c.net, c.base)

    def search(block):
        r = orc.search_batch(oracle.MODE_NET, block.numpy(), c.base, off, nbr, 16, db_low=db_low,
                             net=c.net)
        return torch.from_numpy(r["ids"].astype(np.int32))

    q = torch.from_numpy(c.queries)
    full = sharding.sharded_search(search, q)
    lo, hi = sharding.shard_bounds(n_q, world, rank)
    np.save(os.path.join(tmpdir, f"r{rank}.npy"), full.numpy())
    np.save(os.path.join(tmpdir, f"b{rank}.npy"), np.array([lo, hi]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_q", [(2, 101), (3, 64), (2, 3)])
def test_sharded_search_equals_single(tmp_path, orc, world, n_q):
    import oracle
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_q, str(tmp_path)), nprocs=world, join=True)
    c = datagen.Case("s", 31, 2000, n_q, 32, 8, 16)
    rng = np.random.Generator(np.random.PCG64(32))
    off, nbr = datagen.random_graph(rng, c.n, 4, 20)
    db_low = orc.project(c.net, c.base)
    single = orc.search_batch(oracle.MODE_NET, c.queries, c.base, off, nbr, 16, db_low=db_low,
                              net=c.net)["ids"].astype(np.int32)
    covered = []
    for r in range(world):
        got = np.load(tmp_path / f"r{r}.npy")
        assert np.array_equal(got, single), f"rank {r}"
        covered.append(tuple(np.load(tmp_path / f"b{r}.npy")))
    # blocks tile [0, n_q) without gaps or overlap, sizes differ by at most one
    assert covered[0][0] == 0 and covered[-1][1] == n_q
    assert all(covered[i][1] == covered[i + 1][0] for i in range(world - 1))
    sizes = [b - a for a, b in covered]
    assert max(sizes) - min(sizes) <= 1


def test_shard_bounds_properties():
    from gbnns_dim_red_amd import sharding
    for n_q in (0, 1, 7, 10000, 1000003):
        for world in (1, 2, 4, 8):
            b = [sharding.shard_bounds(n_q, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n_q
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
