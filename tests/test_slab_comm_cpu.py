"""world_size-2/3 gloo tests (CPU) of the slab communication layer and partition used by the multi-GPU path."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, results):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ma_amd.slab_mg import SlabComm
    from ma_amd.slab import Partition
    c = SlabComm(rank, world)
    ok = True
    # halo exchange along the chain: first column -> left neighbour, last column -> right neighbour
    n, m = 23, 5
    part = Partition(n, world, rank)
    field = torch.arange(n * m, dtype=torch.float64).reshape(n, m)          # global field, column j = row j here
    mine = field[part.j0:part.j1]
    from_left = torch.full((m,), -1.0, dtype=torch.float64)
    from_right = torch.full((m,), -1.0, dtype=torch.float64)
    c.exchange(to_left=mine[0].clone(), to_right=mine[-1].clone(), from_left=from_left, from_right=from_right)
    if c.left is not None:
        ok &= bool(torch.equal(from_left, field[part.j0 - 1]))
    else:
        ok &= bool((from_left == -1).all())
    if c.right is not None:
        ok &= bool(torch.equal(from_right, field[part.j1]))
    else:
        ok &= bool((from_right == -1).all())
    # one-directional exchange (x halo: first column -> left neighbour's right halo only)
    fr = torch.full((m,), -2.0, dtype=torch.float64)
    c.exchange(to_left=mine[0].clone() if c.left is not None else None, from_right=fr if c.right is not None else None)
    if c.right is not None:
        ok &= bool(torch.equal(fr, field[part.j1]))
    # all-gather in rank order and fixed-order sum -> identical on every rank
    vec = torch.tensor([float(rank + 1), 10.0 ** rank, 1e-3 * rank], dtype=torch.float64)
    g = torch.zeros((world, 3), dtype=torch.float64)
    c.all_gather(g, vec)
    for r in range(world):
        ok &= bool(torch.equal(g[r], torch.tensor([float(r + 1), 10.0 ** r, 1e-3 * r], dtype=torch.float64)))
    s = torch.tensor([float(rank)], dtype=torch.float64)
    c.all_reduce_sum(s)
    ok &= float(s.item()) == sum(range(world))
    results[rank] = ok
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_slab_comm_gloo(world):
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_worker, args=(world, 29510 + world, results), nprocs=world, join=True)
    assert all(results[r] for r in range(world)) and len(results) == world


def _batch_worker(rank, world, port, results):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import numpy as np
    from ma_amd import batch

    class Stub:   # stands in for the GPU shard solver: "solves" problem p as a function of its images only
        def __init__(self, R, T, *a, **kw):
            self.R, self.T = np.asarray(R), np.asarray(T)

        def run(self, maxIter, numSteps, tol=1e-3):
            u = np.stack([np.concatenate([r.ravel() + 2 * t.ravel(), r.ravel() - t.ravel()]) for r, t in zip(self.R, self.T)]) \
                if len(self.R) else np.zeros((0, 2 * self.R.shape[1] * self.R.shape[2]))
            return u, np.full((len(self.R), numSteps), maxIter, dtype=np.int64)

        def close(self):
            pass

    batch.BatchRegistration = Stub
    P, m, n = 7, 3, 2
    rng = np.random.default_rng(0)
    R, T = rng.standard_normal((P, m, n)), rng.standard_normal((P, m, n))
    u, its = batch.tv_registration_batch(R, T, 1.0, 1.0, 5, 2)
    want = np.stack([np.concatenate([r.ravel() + 2 * t.ravel(), r.ravel() - t.ravel()]) for r, t in zip(R, T)])
    ok = np.array_equal(u, want) and its.shape == (P, 2)
    u_loc, _, (p0, p1) = batch.tv_registration_batch(R, T, 1.0, 1.0, 5, 2, gather=False)
    ok &= (p0, p1) == batch.shard_range(P, world, rank) and np.array_equal(u_loc, want[p0:p1])
    # multi-image batch: the driver is looked up by name in ma_amd.drivers; a stub stands in for the GPU driver
    from ma_amd import drivers
    drivers.var_tv_registration_no_ref_ml = lambda imgs, opt, device=0, **kw: (len(imgs), float(np.sum(imgs[0])) * opt["mu"])
    data = [[rng.standard_normal((3 + p, 2)) for _ in range(2 + p % 3)] for p in range(5)]
    res = batch.multi_image_registration_batch(data, dict(mu=2.0), "var_tv_registration_no_ref_ml")
    ok &= res == [(len(d), float(np.sum(d[0])) * 2.0) for d in data]
    loc, (q0, q1) = batch.multi_image_registration_batch(data, dict(mu=2.0), "var_tv_registration_no_ref_ml", gather=False)
    ok &= (q0, q1) == batch.shard_range(len(data), world, rank) and loc == res[q0:q1]
    try:
        batch.multi_image_registration_batch(data, {}, "no_such_driver")
        ok = False
    except ValueError:
        pass
    results[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_batch_sharding_gloo(world):
    from ma_amd.batch import shard_range
    for P in (1, 2, 7, 256):
        for w in (1, 2, 3, 8):
            rngs = [shard_range(P, w, r) for r in range(w)]
            assert rngs[0][0] == 0 and rngs[-1][1] == P
            assert all(rngs[i][1] == rngs[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in rngs) - min(b - a for a, b in rngs) <= 1
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_batch_worker, args=(world, 29520 + world, results), nprocs=world, join=True)
    assert all(results[r] for r in range(world))
