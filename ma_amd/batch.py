"""Throughput mode: a batch of independent pairwise registrations sharded over the ranks of one node (BASELINE config 5:
256 TV-L2 SSD_registration pairs; SURVEY.md section 8e "batch mode"). Problems are independent, so there is no
data-path collective: rank r owns the contiguous range shard_range(P, world, r), solves it with one engine handle
(batch = local count, one problem per blockIdx.y, per-problem stop flags and histories), and only the results are
gathered -- if the caller asks for them.

Each problem is the script PDO/test_TV_L2_registration.m:56-122 (or test_TV_L1_registration.m) as
ma_amd.drivers.tv_registration runs it for one pair."""
import numpy as np

from . import engine as E
from .drivers import normalize, _vecF, _stackF


def shard_range(nprob, world, rank):
    """contiguous range [p0, p1) of the problems owned by `rank`; the first nprob % world ranks own one more"""
    base, rem = divmod(nprob, world)
    p0 = rank * base + min(rank, rem)
    return p0, p0 + base + (1 if rank < rem else 0)


class BatchRegistration:
    """The local shard of a batch of pairwise registrations on one GPU.

    R, T: arrays (P_local, m, n) of reference / template images (already normalised or not: `normalise` applies
    PDO/test_TV_L1_registration.m:57-58 per image)."""

    def __init__(self, R, T, lam, tau, data_term="SSD", h=(1.0, 1.0), theta=1.0, device=0, fp32=False,
                 normalise=True):
        R = np.asarray(R, dtype=np.float64)
        T = np.asarray(T, dtype=np.float64)
        if R.shape != T.shape or R.ndim != 3:
            raise ValueError("R and T must be (problems, m, n) arrays of the same shape")
        self.P, self.m, self.n = R.shape
        if normalise:
            R = np.stack([normalize(r) for r in R]) if self.P else R
            T = np.stack([normalize(t) for t in T]) if self.P else T
        self.theta, self.tau = theta, tau
        L_squared = 4 * (1 / h[0] ** 2 + 1 / h[1] ** 2)
        self.sigma = (1 - 1e-4) / (L_squared * tau)                              # PDO/test_TV_L1_registration.m:74-76
        model = E.MODEL_TV_L1_REG if data_term == "SAD" else E.MODEL_TV_L2_REG
        self.eng = None
        if self.P:
            self.eng = E.Engine(model, self.m, self.n, h=h, lam=lam, device=device, batch=self.P,
                                flags=E.FLAG_FP32 if fp32 else 0)
            self.eng.set_images(_stackF(T), _stackF(R))

    def run(self, maxIter, numSteps, tol=1e-3):
        """numSteps outer re-linearisations (warm start on the device), maxIter CP iterations each.
        Returns (u, iterations): u (P_local, 2mn) displacement vectors, iterations (P_local, numSteps)."""
        its = np.zeros((self.P, numSteps), dtype=np.int64)
        if not self.P:
            return np.zeros((0, 2 * self.m * self.n)), its
        for s in range(numSteps):
            self.eng.linearize()
            self.eng.solve(self.theta, self.tau, self.sigma, maxIter, tol)
            its[:, s] = [self.eng.result(p).iterations for p in range(self.P)]
        return self.eng.primal().reshape(self.P, -1), its

    def close(self):
        if self.eng is not None:
            self.eng.close()
            self.eng = None


def tv_registration_batch(R, T, lam, tau, maxIter, numSteps, data_term="SSD", h=(1.0, 1.0), theta=1.0, device=0,
                          fp32=False, gather=True, tol=1e-3):
    """All ranks call this with the SAME full arrays R, T (P, m, n); every rank solves its shard. With gather=True the
    displacement fields of all problems are returned on every rank (all_gather of the results, outside any timed
    region); with gather=False only the local shard (rows p0:p1) is returned together with (p0, p1)."""
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    P = len(R)
    p0, p1 = shard_range(P, world, rank)
    solver = BatchRegistration(np.asarray(R)[p0:p1], np.asarray(T)[p0:p1], lam, tau, data_term, h, theta, device, fp32)
    try:
        u_loc, it_loc = solver.run(maxIter, numSteps, tol)
    finally:
        solver.close()
    if world == 1 or not gather:
        return (u_loc, it_loc) if world == 1 else (u_loc, it_loc, (p0, p1))
    parts = [None] * world
    dist.all_gather_object(parts, (p0, u_loc, it_loc))
    parts.sort(key=lambda t: t[0])
    return np.concatenate([p[1] for p in parts]), np.concatenate([p[2] for p in parts])


MULTI_IMAGE_DRIVERS = ("mf_nn_tv_registration_no_ref_ml", "mf_nn_curvature_registration_no_ref_ml",
                       "var_tv_registration_no_ref_ml", "var_curvature_registration_no_ref_ml",
                       "mf_nn_tv_registration_no_ref", "var_tv_registration_no_ref")


def multi_image_registration_batch(datasets, optPara, driver="mf_nn_tv_registration_no_ref_ml", device=0, gather=True, **kw):
    """A batch of independent multi-image registrations (one image sequence per entry of `datasets`, each a list of k
    m x n images; sizes may differ between entries) sharded over the ranks like the pairwise batch above: rank r runs the
    named driver of ma_amd.drivers (= the reference's NNE/<driver>.m, one call per data set) on the contiguous range
    shard_range(P, world, r), one after the other on its GPU -- a multi-image problem of useful size fills the GPU by
    itself (one engine handle per problem: cp_create takes batch = 1 for these models), so problems of one rank gain
    nothing from running side by side. No data-path collective. Returns the list of the drivers' result tuples, for all
    data sets on every rank (gather=True) or for the local shard together with (p0, p1)."""
    from . import drivers as D
    import torch.distributed as dist
    if driver not in MULTI_IMAGE_DRIVERS:
        raise ValueError(f"unknown multi-image driver {driver!r}")
    fn = getattr(D, driver)
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    p0, p1 = shard_range(len(datasets), world, rank)
    local = [fn(datasets[p], optPara, device=device, **kw) for p in range(p0, p1)]
    if world == 1:
        return local
    if not gather:
        return local, (p0, p1)
    parts = [None] * world
    dist.all_gather_object(parts, (p0, local))
    parts.sort(key=lambda t: t[0])
    return [res for _, loc in parts for res in loc]
