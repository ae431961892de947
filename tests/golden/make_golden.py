"""Generates the committed fixtures under tests/golden/ (run in the build container only).

    python tests/golden/make_golden.py

Inputs : the data files the reference ships (Data/*.png, gray channel as MATLAB/Octave imread returns it for
         gray+alpha PNGs) and seeded synthetic arrays.
Outputs: what the CPU restatement (oracle/) computes from them.  The reference itself cannot be run here (MATLAB
         source, no MATLAB/Octave in the container), so these vectors pin the ORACLE against regressions and give the
         GPU tests inputs/outputs that travel to the GPU box; they are not outputs of the reference.
"""
import glob
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import drivers as od, operators as oo, prox as px   # noqa: E402
from oracle.chambolle_pock import chambolle_pock as cp_ref      # noqa: E402
from ma_amd import synth                                        # noqa: E402

REF_DATA = "/root/reference/Data"


def vecF(a):
    return np.asarray(a, dtype=np.float64).reshape(-1, order="F")


def data_png():
    from PIL import Image
    out = {}
    for f in sorted(glob.glob(os.path.join(REF_DATA, "*.png"))):
        name = os.path.splitext(os.path.basename(f))[0]
        a = np.array(Image.open(f))
        if a.shape[2] == 2:          # gray + alpha -> gray channel
            out[name] = a[:, :, 0].copy()
    np.savez_compressed(os.path.join(HERE, "data_png.npz"), **out)
    return out


def operators_fixture():
    rng = np.random.default_rng(20240607)
    out = {}
    m, n, k = 9, 7, 3
    out["shape"] = np.array([m, n, k])
    h = (0.5, 2.0)
    out["h"] = np.array(h)
    x_u = rng.standard_normal(2 * k * m * n)
    y_d = rng.standard_normal(4 * k * m * n)
    y_l = rng.standard_normal(2 * k * m * n)
    L = rng.standard_normal(k * m * n)
    out.update(x_u=x_u, y_d=y_d, y_l=y_l, L=L)
    for bc in ("neumann", "linear"):
        D = oo.finite_difference_operator(m, n, h, k, bc)
        out[f"D_{bc}_x"] = D @ x_u
        out[f"Dt_{bc}_y"] = D.T @ y_d
    for bc in ("neumann", "dirichlet"):
        Lp = oo.discrete_laplacian(m, n, h, k, bc)
        out[f"Lap_{bc}_x"] = Lp @ x_u
    out["B_L"] = oo.mean_free_operator(m, n, k) @ L
    # prox / value / conjugate evaluations (the six the solver uses, plus the remaining branches)
    mn = m * n
    v4 = rng.standard_normal(4 * mn) * 1.5
    I = rng.standard_normal(mn)
    Lv = rng.standard_normal(mn) * 2
    mu, sig = 0.7, 0.3
    out.update(v4=v4, I=I, Lv=Lv, mu=mu, sig=sig)
    out["SAD_val"] = np.array(px.SAD(Lv, I, mu, sig, False, 2)[:2], dtype=float)
    out["SAD_prox"] = px.SAD(Lv, I, mu, sig, False, 3)[2]
    out["SAD_cval"] = np.array(px.SAD(Lv, I, mu, sig, True, 2)[:2], dtype=float)
    out["SAD_cprox"] = px.SAD(Lv, I, mu, sig, True, 3)[2]
    out["SSD_val"] = np.array(px.SSD(Lv, I, mu, sig, False, 2)[:2], dtype=float)
    out["SSD_prox"] = px.SSD(Lv, I, mu, sig, False, 3)[2]
    out["SSD_cval"] = np.array(px.SSD(Lv, I, mu, sig, True, 2)[:2], dtype=float)
    out["SSD_cprox"] = px.SSD(Lv, I, mu, sig, True, 3)[2]
    out["n21_val"] = np.array(px.norm21(v4, mu, sig, False, 2)[:2], dtype=float)
    out["n21_prox"] = px.norm21(v4, mu, sig, False, 3)[2]
    out["n21_cval"] = np.array(px.norm21(v4, mu, sig, True, 2)[:2], dtype=float)
    out["n21_cprox"] = px.norm21(v4, mu, sig, True, 3)[2]
    out["nn2_val"] = np.array(px.pointwise_2x2_nn(v4, mu, sig, False, 2)[:2], dtype=float)
    out["nn2_prox"] = px.pointwise_2x2_nn(v4, mu, sig, False, 3)[2]
    out["nn2_cval"] = np.array(px.pointwise_2x2_nn(v4, mu, sig, True, 2)[:2], dtype=float)
    out["nn2_cprox"] = px.pointwise_2x2_nn(v4, mu, sig, True, 3)[2]
    # svd_2x2 incl. degenerate rows: zero, rank one, sigma1 = sigma2 (scaled rotation), diagonal
    M = np.vstack([rng.standard_normal((6, 4)), np.zeros((1, 4)), [[1.0, 2.0, 2.0, 4.0]], [[0.0, 1.0, -1.0, 0.0]],
                   [[3.0, 0.0, 0.0, -2.0]], [[2.0, 0.0, 0.0, 2.0]]])
    U, S, V = px.svd_2x2(M)
    out.update(svd_M=M, svd_U=U, svd_S=S, svd_V=V)
    # l1 ball: inside, outside, ties, negative entries
    l1_in = [np.array([0.2, -0.3, 0.1]), np.array([3.0, -1.0, 0.5, 0.5]), np.array([1.0, 1.0, 1.0, 1.0]),
             rng.standard_normal(10) * 2, np.array([5.0])]
    for i, v in enumerate(l1_in):
        out[f"l1_in{i}"] = v
        out[f"l1_out{i}"] = px.l1ball_projection(v)
    Y = rng.standard_normal(mn * k)
    nu, tau = 2.5, 0.4
    out.update(nnc_Y=Y, nnc_nu=nu, nnc_tau=tau)
    out["nnc_val"] = np.array(px.nuclear_norm_constraint(Y, k, tau, nu, False, 2)[:2], dtype=float)
    out["nnc_prox"] = px.nuclear_norm_constraint(Y, k, tau, nu, False, 3)[2]
    out["nnc_cval"] = np.array(px.nuclear_norm_constraint(Y, k, tau, nu, True, 2)[:2], dtype=float)
    out["nnc_cprox"] = px.nuclear_norm_constraint(Y, k, tau, nu, True, 3)[2]
    out["mzi_val"] = np.array(px.mean_zero_indicator(x_u, (m, n, k), False, 2)[:2], dtype=float)
    out["mzi_prox"] = px.mean_zero_indicator(x_u, (m, n, k), False, 3)[2]
    r = px.mean_zero_indicator(x_u, (m, n, k), True, 3)
    out["mzi_cval"] = np.array(r[:2], dtype=float)
    out["mzi_cprox"] = r[2]
    out["zf_cval"] = np.array(px.zero_function(L, True, 2)[:2], dtype=float)
    out["tvr_val"] = np.array(px.TV_registration(v4, sig, False, 2)[:2], dtype=float)
    r = px.TV_registration(v4, sig, True, 3)
    out["tvr_cval"] = np.array(r[:2], dtype=float)
    out["tvr_cprox"] = r[2]
    v2 = rng.standard_normal(2 * mn)
    out["v2"] = v2
    r = px.TV_denoise(v2, sig, True, 3)
    out["tvd_cval"] = np.array(r[:2], dtype=float)
    out["tvd_cprox"] = r[2]
    np.savez_compressed(os.path.join(HERE, "operators.npz"), **out)


def warp_fixture(png):
    rng = np.random.default_rng(77)
    T = od.normalize(png["rect_2"].astype(np.float64))
    m, n = T.shape
    u = 3.0 * rng.standard_normal((m * n, 2))
    u[:40] += 60.0           # far out of the domain
    u[40:80] = np.round(u[40:80])   # exactly on nodes / cell boundaries
    Tu, g = oo.evaluate_displacement(T, (1.0, 1.0), u)
    # s_grid resampling facility (C2 upsampling recipe), small target
    Tup, _ = oo.evaluate_displacement(T, (1.0, 1.0), np.zeros((96 * 80, 2)), [0.0, float(m), 0.0, float(n)], (96, 80))
    # multi-level style call: h_grid as image spacing, omega fixed
    Tc = oo.downsample(T)
    mc, nc = Tc.shape
    hg = (m / mc, n / nc)
    uc = 2.0 * rng.standard_normal((mc * nc, 2))
    Tcu, gc = oo.evaluate_displacement(Tc, hg, uc, [0.0, float(m), 0.0, float(n)])
    np.savez_compressed(os.path.join(HERE, "warp.npz"), T=T, u=u, Tu=Tu, g=g, Tup=Tup, Tc=Tc, hg=np.array(hg), uc=uc,
                        Tcu=Tcu, gc=gc)


def pyramid_fixture():
    rng = np.random.default_rng(5)
    out = {}
    for (m, n) in ((9, 6), (8, 8), (7, 11)):
        a = rng.standard_normal((m, n))
        d = oo.downsample(a)
        xl = rng.standard_normal(d.size * 2)
        out[f"img_{m}x{n}"] = a
        out[f"down_{m}x{n}"] = d
        out[f"xlow_{m}x{n}"] = xl
        out[f"prol_{m}x{n}"] = oo.prolong_fields(xl, d.shape[0], d.shape[1], m, n, 2)
    np.savez_compressed(os.path.join(HERE, "pyramid.npz"), **out)


def pairwise_e2e(png):
    out = {}
    # C1: TV-L1 denoising on rect1.png with a committed salt & pepper mask (own definition, replaces imnoise)
    img = od.normalize(png["rect1"].astype(np.float64))
    rng = np.random.default_rng(1234)
    r = rng.random(img.shape)
    p = 0.25
    noisy = img.copy()
    noisy[r < p / 2] = 0.0
    noisy[(r >= p / 2) & (r < p)] = 1.0
    x, y, ph, dh, sigma = od.tv_l1_denoising(noisy, lam=1.5, tau=0.02, maxIter=1000, tol=1e-5)
    out.update(c1_noisy=noisy, c1_x=x, c1_y=y, c1_ph=ph, c1_dh=dh)
    # C2-small: rect_1 -> rect_2 (TV-L1, lambda 0.7, tau 5, 30 its) and rect_in_rect (TV-L2): states of the first
    # outer steps (each step is checked from the stored start state), plus the final energies of the whole script
    for tag, a, b, term, lam, tau, iters, steps in (("tvl1_rect", "rect_1", "rect_2", "SAD", 0.7, 5.0, 30, 40),
                                                    ("tvl2_rir", "rect_in_rect_1", "rect_in_rect_2", "SSD", 50.0, 3.0, 25, 40),
                                                    ("tvl1_slide", "sliding_rect_1", "sliding_rect_2", "SAD", 4.0, 1.5, 30, 40)):
        R = od.normalize(png[a].astype(np.float64))
        T = od.normalize(png[b].astype(np.float64))
        m, n = R.shape
        K = oo.pairwise_gradient(m, n, (1.0, 1.0), 2)
        sigma = (1 - 1e-4) / (8 * tau)
        Gfun = px.SAD_registration if term == "SAD" else px.SSD_registration
        F = lambda v, c, nargout: px.TV_registration(v, sigma, c, nargout)
        u0, v0 = np.zeros(2 * m * n), np.zeros(4 * m * n)
        keep_steps = (0, 2, 20)
        energies, unorm, traj = [], [], []
        for s in range(steps):
            G = lambda u, c, nargout, u0_=u0.copy(): Gfun(u, u0_, T, R, (1.0, 1.0), lam, tau, c, nargout)
            xs, ys, ph, dh = cp_ref(F, G, K, u0, v0, 1.0, tau, sigma, iters)
            if s in keep_steps:
                out[f"{tag}_s{s}_u0"] = u0.copy()
                out[f"{tag}_s{s}_v0"] = v0.copy()
                out[f"{tag}_s{s}_u"] = xs
                out[f"{tag}_s{s}_v"] = ys
                out[f"{tag}_s{s}_ph"] = ph
                out[f"{tag}_s{s}_dh"] = dh
            energies.append([ph[-1, 0], dh[-1, 0], ph.shape[0]])
            traj.append(xs.copy())
            u0, v0 = xs, ys
        out[f"{tag}_energies"] = np.array(energies)
        out[f"{tag}_par"] = np.array([lam, tau, iters, steps])
        out[f"{tag}_u_final"] = traj[-1]
        # Sensitivity of the free-running script: the same script with the template perturbed by ONE ulp in every pixel
        # (up, then down). Every re-linearisation re-selects interpolation cells (floor, IRE/bilinear_interpolation.m:42)
        # and the SAD prox divides by |grad T|^2 (PDO/SAD_registration.m:74-75), so after a few outer steps two correct
        # implementations are 1e-4 .. 5e-3 apart. Column 0: relative distance of u after step s, column 1: relative
        # difference of the final primal energy of step s (maximum over the two perturbations).
        sens = np.zeros((steps, 2))
        for Tp in (np.nextafter(T, 2.0), np.nextafter(T, -1.0)):
            u0, v0 = np.zeros(2 * m * n), np.zeros(4 * m * n)
            for s in range(steps):
                G = lambda u, c, nargout, u0_=u0.copy(): Gfun(u, u0_, Tp, R, (1.0, 1.0), lam, tau, c, nargout)
                xs, ys, ph, dh = cp_ref(F, G, K, u0, v0, 1.0, tau, sigma, iters)
                sens[s, 0] = max(sens[s, 0], np.linalg.norm(xs - traj[s]) / np.linalg.norm(traj[s]))
                sens[s, 1] = max(sens[s, 1], abs(ph[-1, 0] - energies[s][0]) / abs(energies[s][0]))
                u0, v0 = xs, ys
        out[f"{tag}_sens"] = sens
    np.savez_compressed(os.path.join(HERE, "pairwise_e2e.npz"), **out)


def mfnn_e2e():
    out = {}
    m, n, k = 64, 64, 4
    imgs = synth.dynamic_test_image(m, n, k).numpy()
    out["imgs"] = imgs
    for reg, bc, mu in (("tv", "neumann", 0.2), ("curvature", "dirichlet", 20.0)):
        opt = dict(theta=1.0, maxIter=60, tol=1e-3, outerIter=[3, 2], mu=mu, nu_factor=[0.9, 0.9], bc=bc)
        keep = {}
        u, L, SV = od.mf_nn_registration_no_ref_ml([imgs[i] for i in range(k)], opt, reg, keep=keep)
        levels = len(u)
        out[f"{reg}_levels"] = np.array([levels])
        for lev in range(levels):
            out[f"{reg}_SV{lev}"] = SV[lev]
            for o in range(len(u[lev])):
                if u[lev][o] is None:
                    continue
                out[f"{reg}_u_{lev}_{o}"] = u[lev][o]
                out[f"{reg}_L_{lev}_{o}"] = L[lev][o]
                kk = keep[(lev, o)]
                out[f"{reg}_sched_{lev}_{o}"] = np.array([kk["nu"], kk["tau"], kk["norm_est"], kk["n_power"],
                                                          kk["primal_history"].shape[0]])
                out[f"{reg}_ph_{lev}_{o}"] = kk["primal_history"]
                out[f"{reg}_dh_{lev}_{o}"] = kk["dual_history"]
    np.savez_compressed(os.path.join(HERE, "mfnn_e2e.npz"), **out)


def var_e2e():
    """variance drivers (NNE/var_{tv,curvature}_registration_no_ref_ml.m) on the frames of mfnn_e2e.npz"""
    out = {}
    m, n, k = 64, 64, 4
    imgs = synth.dynamic_test_image(m, n, k).numpy()
    for reg, bc, mu in (("tv", "neumann", 0.1), ("curvature", "neumann", 5.0)):
        opt = dict(theta=1.0, maxIter=60, tol=1e-3, outerIter=[3, 2], mu=mu, bc=bc)
        keep = {}
        u = od.var_registration_no_ref_ml([imgs[i] for i in range(k)], opt, reg, keep=keep)
        out[f"{reg}_levels"] = np.array([len(u)])
        for lev in range(len(u)):
            for o in range(len(u[lev])):
                if u[lev][o] is None:
                    continue
                kk = keep[(lev, o)]
                out[f"{reg}_u_{lev}_{o}"] = u[lev][o]
                out[f"{reg}_sched_{lev}_{o}"] = np.array([kk["tau"], kk["norm_est"], kk["n_power"],
                                                          kk["primal_history"].shape[0]])
                out[f"{reg}_ph_{lev}_{o}"] = kk["primal_history"]
                out[f"{reg}_dh_{lev}_{o}"] = kk["dual_history"]
    np.savez_compressed(os.path.join(HERE, "var_e2e.npz"), **out)


def single_level_e2e():
    """single-level drivers NNE/mf_nn_tv_registration_no_ref.m (F1 = SAD weight 1, F2 = norm21 weight mu, :206-216) and
    NNE/var_tv_registration_no_ref.m (swapped SSD prox arguments, :173) on 56 x 48, K = 3 (the _ml drivers would still
    use one level here, so the difference to them is the functional alone)"""
    out = {}
    m, n, k = 56, 48, 3
    imgs = synth.dynamic_test_image(m, n, k).numpy()
    out["imgs"] = imgs
    opt = dict(theta=1.0, maxIter=50, tol=1e-3, outerIter=3, mu=0.15, nu_factor=0.9, bc="linear")
    keep = {}
    u, L, SV = od.mf_nn_registration_no_ref_ml([imgs[i] for i in range(k)], opt, "tv", keep=keep, single_level=True)
    out["mfnn_SV"] = SV[0]
    for o in range(3):
        kk = keep[(0, o)]
        out[f"mfnn_u_{o}"] = u[0][o]
        out[f"mfnn_L_{o}"] = L[0][o]
        out[f"mfnn_sched_{o}"] = np.array([kk["nu"], kk["tau"], kk["norm_est"], kk["n_power"], kk["primal_history"].shape[0]])
        out[f"mfnn_ph_{o}"] = kk["primal_history"]
        out[f"mfnn_dh_{o}"] = kk["dual_history"]
    # the same inputs through the _ml functional (pointwise 2x2 nuclear norm): must differ, or the fixture pins nothing
    u_ml, _, _ = od.mf_nn_registration_no_ref_ml([imgs[i] for i in range(k)], dict(opt, outerIter=[3, 3], nu_factor=[0.9, 0.9]), "tv")
    out["mfnn_ml_u_2"] = u_ml[0][2]
    optv = dict(theta=1.0, maxIter=50, tol=1e-3, outerIter=3, mu=0.1, bc="neumann")
    keep = {}
    uv = od.var_registration_no_ref_ml([imgs[i] for i in range(k)], optv, "tv", keep=keep, single_level=True)
    for o in range(3):
        kk = keep[(0, o)]
        out[f"var_u_{o}"] = uv[0][o]
        out[f"var_sched_{o}"] = np.array([kk["tau"], kk["norm_est"], kk["n_power"], kk["primal_history"].shape[0]])
        out[f"var_ph_{o}"] = kk["primal_history"]
        out[f"var_dh_{o}"] = kk["dual_history"]
    np.savez_compressed(os.path.join(HERE, "single_level_e2e.npz"), **out)


if __name__ == "__main__":
    only = set(sys.argv[1:])        # e.g. `make_golden.py mfnn single` regenerates two fixtures only
    png = data_png()
    steps = dict(operators=operators_fixture, warp=lambda: warp_fixture(png), pyramid=pyramid_fixture,
                 pairwise=lambda: pairwise_e2e(png), mfnn=mfnn_e2e, var=var_e2e, single=single_level_e2e)
    for name, fn in steps.items():
        if not only or name in only:
            fn()
    for f in sorted(glob.glob(os.path.join(HERE, "*.npz"))):
        print(os.path.basename(f), os.path.getsize(f) // 1024, "KiB")
