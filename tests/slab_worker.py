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


def _finish(ok):
    flag = torch.tensor([1 if ok else 0])
    dist.broadcast(flag, 0)
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


def main_solve(rank, world):
    """MultiGpuEngine.solve (history row 0, stop rule on the device, polled every 32 iterations) against cp_solve on a
    single handle: identical iteration counts, histories and iterates; fp64 and fp32 storage"""
    from ma_amd import engine as E, slab, synth
    from ma_amd.slab_mg import MultiGpuEngine
    reg = os.environ.get("SLAB_REG", "tv")
    fp32 = os.environ.get("SLAB_FP32", "0") == "1"
    flags = E.FLAG_FP32 if fp32 else 0
    m, n, k = 70, 52, int(os.environ.get("SLAB_K", "5"))
    model = E.MODEL_MF_NN_TV if reg == "tv" else E.MODEL_MF_NN_CURV
    bc = "neumann" if reg == "tv" else os.environ.get("SLAB_BC", "dirichlet")
    mu = 0.2 if reg == "tv" else 5.0
    max_iter, tol = int(os.environ.get("SLAB_MAXITER", "150")), float(os.environ.get("SLAB_TOL", "2e-2"))
    imgs = synth.dynamic_test_image(m, n, k).numpy()
    imgs_host = np.concatenate([imgs[i].reshape(-1, order="F") for i in range(k)])
    u0 = synth.smooth_displacement(m, n, k, amp=1.0).numpy()
    omega = [0.0, float(m), 0.0, float(n)]
    part = slab.Partition(n, world, rank)
    mg = MultiGpuEngine(model, m, n, k, bc, (1.0, 1.0), omega, mu, 0, part, flags=flags)
    mg.set_images(imgs_host)
    mg.set_primal(np.concatenate([u0[:, :, part.j0:part.j1, :].reshape(-1), np.zeros(k * m * part.nl)]))
    mg.linearize()
    nu = 0.9 * float(np.sum(mg.singular_values(0)))
    mg.set_nu(nu)
    est, npow = mg.norm_estimate()
    tau = sigma = float(np.sqrt(0.99 / est ** 2))
    res = mg.solve(1.0, tau, sigma, max_iter, tol, chunk=int(os.environ.get("SLAB_CHUNK", "32")))
    ph, dh = mg.history()
    x_full = mg.primal_full()
    svL = mg.singular_values(1)
    mg.close()
    ok = True
    if rank == 0:
        ref = E.Engine(model, m, n, k=k, bc=bc, h=(1.0, 1.0), omega=omega, mu=mu, flags=flags)
        ref.set_images(imgs_host)
        ref.set_primal(np.concatenate([u0.reshape(-1), np.zeros(k * m * n)]))
        ref.linearize()
        ref.set_nu(nu)
        r1 = ref.solve(1.0, tau, sigma, max_iter, tol)
        ph1, dh1 = ref.history()
        x1 = ref.primal()
        sv1 = ref.singular_values(1)
        ref.close()
        rt = 1e-5 if fp32 else 1e-9
        same_rows = ph.shape == ph1.shape and res.iterations == r1.iterations and res.stop_reason == r1.stop_reason
        eh = np.abs(ph[:, :5] - ph1[:, :5]).max() / np.abs(ph1[:, :5]).max() if same_rows else np.inf
        ed = np.abs(dh[:, :5] - dh1[:, :5]).max() / np.abs(dh1[:, :5]).max() if same_rows else np.inf
        ex = np.linalg.norm(x_full - x1) / np.linalg.norm(x1)
        print(f"slab solve world={world} reg={reg} fp32={fp32}: iterations {res.iterations} vs {r1.iterations} (reason "
              f"{res.stop_reason} vs {r1.stop_reason}), histories {eh:.2e} {ed:.2e}, x {ex:.2e}, svL "
              f"{np.abs(svL - sv1).max() / sv1.max():.2e}", flush=True)
        ok = same_rows and eh < rt and ed < rt and ex < rt and np.abs(svL - sv1).max() < rt * sv1.max()
        if os.environ.get("SLAB_EXPECT_STOP", "0") == "1":
            ok = ok and r1.stop_reason == E.STOP_TOLERANCE and r1.iterations < max_iter
    _finish(ok)


def main_driver(rank, world):
    """the multi-level driver with the fine levels on row slabs (coarse levels replicated, slab boundaries doubling from
    level to level) against the same driver on one handle per level"""
    from ma_amd import drivers as D, synth
    reg = os.environ.get("SLAB_REG", "tv")
    m, n, k = int(os.environ.get("SLAB_M", "128")), int(os.environ.get("SLAB_N", "136")), int(os.environ.get("SLAB_K", "3"))
    imgs = [synth.dynamic_test_image(m, n, k).numpy()[i] for i in range(k)]
    bc, mu = ("neumann", 0.2) if reg == "tv" else ("dirichlet", 20.0)
    opt = dict(theta=1.0, maxIter=40, tol=1e-3, outerIter=[2, 2], mu=mu, nu_factor=[0.9, 0.9], bc=bc)
    keep_s = {}
    us, Ls, SVs = D._mf_nn_ml(imgs, opt, reg, keep=keep_s, slabs=(world, rank), slab_min_pixels=60 * 60)
    ok = True
    if rank == 0:
        keep_1 = {}
        u1, L1, SV1 = D._mf_nn_ml(imgs, opt, reg, keep=keep_1)
        worst = 0.0
        ok = len(us) == len(u1) == 3 and set(keep_s) == set(keep_1)
        for key in sorted(keep_1):
            a, b = keep_s[key], keep_1[key]
            ok = ok and a["iterations"] == b["iterations"] and a["n_power"] == b["n_power"]
            ok = ok and abs(a["nu"] - b["nu"]) <= 1e-11 * abs(b["nu"]) and abs(a["norm_est"] - b["norm_est"]) <= 1e-11 * b["norm_est"]
            lev, o = key
            worst = max(worst, np.linalg.norm(us[lev][o] - u1[lev][o]) / np.linalg.norm(u1[lev][o]),
                        np.linalg.norm(Ls[lev][o] - L1[lev][o]) / np.linalg.norm(L1[lev][o]),
                        np.abs(a["primal_history"][:, :5] - b["primal_history"][:, :5]).max() / np.abs(b["primal_history"][:, :5]).max())
        for lev in range(3):
            worst = max(worst, np.abs(SVs[lev] - SV1[lev]).max() / np.abs(SV1[lev]).max())
        print(f"slab driver world={world} reg={reg}: levels {len(us)}, worst relative difference {worst:.2e}", flush=True)
        ok = ok and worst < 1e-9
    _finish(ok)


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    mode = os.environ.get("SLAB_MODE", "iterate")
    if mode == "solve":
        return main_solve(rank, world)
    if mode == "driver":
        return main_driver(rank, world)
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
