"""Worker of tests/test_gpu_batch.py: run under torch.distributed.run with 2+ ranks (gloo) on ONE GPU. Every rank
solves its shard of a batch of TV-L2 registrations; rank 0 also solves the whole batch on one handle and compares
(bitwise: problems are independent)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    from ma_amd import synth
    from ma_amd.batch import tv_registration_batch, BatchRegistration, shard_range
    m, n, P = 40, 36, int(os.environ.get("BATCH_P", "7"))
    fp32 = os.environ.get("BATCH_FP32", "0") == "1"
    fr = synth.dynamic_test_image(m, n, 2).numpy()
    R = np.stack([fr[0]] * P)
    T = np.stack([np.roll(fr[1], s, axis=0) for s in range(P)])
    u, its = tv_registration_batch(R, T, 50.0, 4.0, 25, 3, "SSD", fp32=fp32)
    assert u.shape == (P, 2 * m * n) and its.shape == (P, 3)
    p0, p1 = shard_range(P, world, rank)
    u_loc, it_loc, rng = tv_registration_batch(R, T, 50.0, 4.0, 25, 3, "SSD", fp32=fp32, gather=False)
    assert rng == (p0, p1) and np.array_equal(u_loc, u[p0:p1])
    if rank == 0:
        one = BatchRegistration(R, T, 50.0, 4.0, "SSD", fp32=fp32)
        u1, it1 = one.run(25, 3)
        one.close()
        assert np.array_equal(u1, u), float(np.abs(u1 - u).max())
        assert np.array_equal(it1, its)
        assert np.abs(u).max() > 1e-3
        print("batch shards == single handle", P, world, "fp32" if fp32 else "fp64")
    if os.environ.get("BATCH_MULTI", "0") == "1":
        multi_image(rank, world)
    dist.barrier()
    dist.destroy_process_group()


def _same(a, b):
    if isinstance(a, (list, tuple)):
        return isinstance(b, (list, tuple)) and len(a) == len(b) and all(_same(x, y) for x, y in zip(a, b))
    if a is None or b is None:
        return a is None and b is None
    return np.array_equal(np.asarray(a), np.asarray(b))


def multi_image(rank, world):
    """batch of multi-image registrations (sequences of different sizes and lengths) sharded over the ranks: every rank gets
    the results of all data sets, equal to the drivers called one by one"""
    from ma_amd import synth, drivers as D
    from ma_amd.batch import multi_image_registration_batch, shard_range
    shapes = [(48, 40, 3), (70, 66, 5), (40, 36, 4), (52, 44, 18)]
    data = [[f for f in synth.dynamic_test_image(m, n, k).numpy()] for m, n, k in shapes]
    for driver, opt in (("mf_nn_tv_registration_no_ref_ml", dict(theta=1.0, maxIter=30, tol=1e-3, outerIter=[2, 2], mu=0.2, nu_factor=[0.9, 0.9], bc="neumann")),
                        ("var_curvature_registration_no_ref_ml", dict(theta=1.0, maxIter=30, tol=1e-3, outerIter=[2, 2], mu=5.0, bc="dirichlet"))):
        res = multi_image_registration_batch(data, opt, driver)
        assert len(res) == len(data)
        loc, rng = multi_image_registration_batch(data, opt, driver, gather=False)
        assert rng == shard_range(len(data), world, rank) and _same(loc, res[rng[0]:rng[1]])
        if rank == 0:
            for p, imgs in enumerate(data):
                assert _same(res[p], getattr(D, driver)(imgs, opt)), (driver, p)
    if rank == 0:
        print("multi-image batch == drivers one by one", len(data), world)


if __name__ == "__main__":
    main()
