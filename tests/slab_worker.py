"""Worker of tests/test_gpu_slab.py: run under torch.distributed.run with 2+ ranks (gloo) on ONE GPU.
Every rank owns a column slab of the same NN-TV problem; rank 0 also solves the whole problem on a single handle
and compares."""
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
    from ma_amd import engine as E, slab, synth
    from ma_amd.slab_mg import MultiGpuEngine
    m, n, k = 48, 44, int(os.environ.get("SLAB_K", "16"))
    reg = os.environ.get("SLAB_REG", "tv")
    model = E.MODEL_MF_NN_TV if reg == "tv" else E.MODEL_MF_NN_CURV
    bc = "neumann" if reg == "tv" else os.environ.get("SLAB_BC", "dirichlet")
    mu = 0.2 if reg == "tv" else 5.0
    iters = 14
    imgs = synth.dynamic_test_image(m, n, k).numpy()
    imgs_host = np.concatenate([imgs[i].reshape(-1, order="F") for i in range(k)])
    u0 = synth.smooth_displacement(m, n, k, amp=1.5).numpy()            # [k][2][n][m]
    omega = [0.0, float(m), 0.0, float(n)]
    part = slab.Partition(n, world, rank)
    mg = MultiGpuEngine(model, m, n, k, bc, (1.0, 1.0), omega, mu, 0, part)
    mg.set_images(imgs_host)
    x0_loc = np.concatenate([u0[:, :, part.j0:part.j1, :].reshape(-1), np.zeros(k * m * part.nl)])
    mg.set_primal(x0_loc)
    mg.linearize()
    sv = mg.singular_values(0)
    nu = 0.9 * float(np.sum(sv))
    mg.set_nu(nu)
    est, npow = mg.norm_estimate()
    tau = sigma = float(np.sqrt(0.99 / est ** 2))
    mg.iterate(1.0, tau, sigma, iters)
    x_loc = mg.primal()
    y_loc = mg.dual()
    mg.close()
    # gather the slabs on rank 0
    xs = [None] * world
    ys = [None] * world
    dist.gather_object((part.j0, part.j1, x_loc, y_loc), xs if rank == 0 else None, dst=0)
    ok = True
    if rank == 0:
        ref = E.Engine(model, m, n, k=k, bc=bc, h=(1.0, 1.0), omega=omega, mu=mu)
        ref.set_images(imgs_host)
        ref.set_primal(np.concatenate([u0.reshape(-1), np.zeros(k * m * n)]))
        ref.linearize()
        sv1 = ref.singular_values(0)
        est1, npow1 = ref.norm_estimate()
        ref.set_nu(nu)
        ref.iterate(1.0, tau, sigma, iters)
        ref.sync()
        X = ref.primal().reshape(3 * k, n, m)
        ny = 6 if reg == "tv" else 4
        Y = ref.dual().reshape(ny * k, n, m)
        ref.close()
        ex = ey = 0.0
        for (j0, j1, xl, yl) in xs:
            ex = max(ex, np.abs(xl.reshape(3 * k, j1 - j0, m) - X[:, j0:j1, :]).max())
            ey = max(ey, np.abs(yl.reshape(ny * k, j1 - j0, m) - Y[:, j0:j1, :]).max())
        esv = np.abs(sv - sv1).max() / sv1.max()
        print(f"slab world={world}: max|x-x1|={ex:.3e} max|y-y1|={ey:.3e} sv_rel={esv:.3e} norm {est:.15g} vs {est1:.15g} "
              f"power its {npow} vs {npow1}", flush=True)
        ok = ex < 1e-11 and ey < 1e-11 and esv < 1e-11 and abs(est - est1) < 1e-11 * est1 and npow == npow1
    flag = torch.tensor([1 if ok else 0])
    dist.broadcast(flag, 0)
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()
