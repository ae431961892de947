"""Row-slab decomposition of one mf_nn problem over the ranks of a torch.distributed job.

The device layout is [field][j][i] with i (image rows) contiguous; the image is split along j into
`world` contiguous slabs.  world == 1 is the plain single-GPU engine.
"""
import os

import numpy as np

from . import engine as E


class Partition:
    """columns [j0, j1) of an image with n columns owned by `rank` of `world`"""

    def __init__(self, n, world=1, rank=0):
        base, rem = divmod(n, world)
        self.n, self.world, self.rank = n, world, rank
        self.j0 = rank * base + min(rank, rem)
        self.nl = base + (1 if rank < rem else 0)
        self.j1 = self.j0 + self.nl
        if self.nl < 2:
            raise ValueError("slab thinner than 2 columns")

    @classmethod
    def from_bounds(cls, n, world, rank, j0, j1):
        p = cls.__new__(cls)
        p.n, p.world, p.rank, p.j0, p.j1, p.nl = n, world, rank, j0, j1, j1 - j0
        if p.nl < 2:
            raise ValueError("slab thinner than 2 columns")
        return p


def hierarchical_partitions(ns, world, rank):
    """Partitions of a pyramid of column counts ns (coarsest first, each level ceil(n/2) of the next finer one) whose
    slab boundaries double from level to level: the parent column (j >> 1) of every owned column is owned by the same
    rank one level below, so prolongation (kron(x, ones(2)), NNE/mf_nn_tv...:113-119) needs no communication."""
    parts = [Partition(ns[0], world, rank)]
    bounds = [Partition(ns[0], world, r).j0 for r in range(world)] + [ns[0]]
    for n in ns[1:]:
        bounds = [min(2 * b, n) for b in bounds[:-1]] + [n]
        parts.append(Partition.from_bounds(n, world, rank, bounds[rank], bounds[rank + 1]))
    return parts


class SlabSolver:
    def __init__(self, model, m, n, k, bc, h, omega, mu, device, part, flags=0):
        self.part = part
        self.m, self.n, self.k = m, n, k
        # CP_FORCE_SLAB_PATH=1 (tests): run the multi-GPU driver on a single rank too, so that the RCCL code path
        # (device-tensor all-gather, stream hand-over) is exercised on a one-GPU box
        if part.world == 1 and os.environ.get("CP_FORCE_SLAB_PATH", "0") != "1":
            self.eng = E.Engine(model, m, n, k=k, bc=bc, h=h, omega=omega, mu=mu, device=device, flags=flags)
        else:
            from .slab_mg import MultiGpuEngine
            self.eng = MultiGpuEngine(model, m, n, k, bc, h, omega, mu, device, part, flags=flags)

    def set_images(self, imgs_full):
        self.eng.set_images(imgs_full)

    def set_primal(self, x0_local):
        self.eng.set_primal(x0_local)

    def linearize(self):
        self.eng.linearize()

    def singular_values(self, which):
        return self.eng.singular_values(which)

    def set_nu(self, nu):
        self.eng.set_nu(nu)

    def norm_estimate(self):
        return self.eng.norm_estimate()

    def iterate(self, theta, tau, sigma, iters):
        self.eng.iterate(theta, tau, sigma, iters)

    def iterate_begin(self, theta, tau, sigma, max_iters):
        self.eng.iterate_begin(theta, tau, sigma, max_iters)

    def iterate_steps(self, iters):
        self.eng.iterate_steps(iters)

    def iterate_end(self):
        self.eng.iterate_end()

    def sync(self):
        self.eng.sync()

    def kernel_times(self):
        return {name: self.eng.kernel_time_ms(name) for name in ("dual", "eig", "primal", "fin", "iterate")}

    def _handle(self):
        return getattr(self.eng, "eng", self.eng)        # MultiGpuEngine wraps the rank's Engine

    def dominant_kernel(self):
        return self._handle().dominant_kernel()

    def handover_fallbacks(self):
        return self._handle().handover_fallbacks()

    def close(self):
        self.eng.close()
