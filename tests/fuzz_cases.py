"""Random problem shapes (odd sizes, few columns, every K, both boundary conditions, grid spacings, forced 8-wave
instantiations) of the four multi-image models, engine against numpy oracle. Used by tests/test_gpu_fuzz.py and
scratch/fuzz_parity.py (a helper module: pytest does not collect it)."""
import os
import numpy as np

rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def run_case(t, rng):
    """one random problem (model family alternates with t) on the engine and on the numpy oracle: dict with the relative
    deviations of x and y, the worst history deviation beyond 1e-9 relative (absolute), and ok"""
    from ma_amd import engine as E, synth
    from oracle import drivers as od, prox as px
    from oracle.chambolle_pock import chambolle_pock as cp_ref
    fam = ("mf", "var")[t % 2]
    reg = ("tv", "curvature")[int(rng.integers(2))]
    bc = ("neumann", "linear")[int(rng.integers(2))] if reg == "tv" else ("neumann", "dirichlet")[int(rng.integers(2))]
    m = int(rng.choice([5, 17, 33, 63, 64, 65, 100, 129, 140]))
    n = int(rng.choice([3, 4, 5, 7, 8, 9, 15, 21, 33, 40, 66]))
    k = int(rng.integers(2, 17 if fam == "mf" else 24))
    h = (float(rng.choice([1.0, 0.5, 2.0])), float(rng.choice([1.0, 0.25, 1.5])))
    waves = str(rng.choice(["", "4", "8"]))
    for var in ("CP_TUNE_FUSED_WAVES", "CP_TUNE_VAR_WAVES"):
        os.environ.pop(var, None)
        if waves:
            os.environ[var] = waves
    mu = (0.2 if reg == "tv" else 20.0) if fam == "mf" else (0.1 if reg == "tv" else 5.0)
    imgs = synth.dynamic_test_image(m, n, k).numpy()
    if not np.isfinite(imgs).all():          # the generator degenerates on a few pixels: any images will do
        imgs = np.random.default_rng(100 + t).random((k, m, n))
    u0 = synth.smooth_displacement(m, n, k, amp=1.0).numpy().reshape(k, 2, n * m)
    u0_ref = np.transpose(u0, (2, 1, 0))
    omega = [0.0, m * h[0], 0.0, n * h[1]]
    kmn = k * m * n
    iters = 12
    tau = sigma = 0.2 if reg == "tv" else 0.05 * min(h) ** 2
    yr0 = np.random.default_rng(t)
    if fam == "mf":
        A, b, T_u, gs = od.mf_nn_problem([imgs[i] for i in range(k)], u0_ref, h, omega, reg, bc)
        ncomp = 6 if reg == "tv" else 4
        x0 = np.concatenate([u0.reshape(-1), 0.5 * T_u.reshape(-1, order="F")])
        y0 = 0.01 * yr0.standard_normal(ncomp * kmn)
        nu = 2.0 * k
        Fh = lambda y, c, nargout: od.mf_nn_F(y, b, k, h, mu, nu, sigma, c, nargout, reg)
        Gh = lambda x, c, nargout: od.mf_nn_G(x, (m, n, k), c, nargout)
        model = E.MODEL_MF_NN_TV if reg == "tv" else E.MODEL_MF_NN_CURV
    else:
        A, b, T_u, gs = od.var_problem([imgs[i] for i in range(k)], u0_ref, h, omega, reg, bc)
        ncomp = 5 if reg == "tv" else 3
        x0 = u0.reshape(-1)
        y0 = 0.02 * yr0.standard_normal(ncomp * kmn) + 0.01
        Fh = lambda y, c, nargout: od.var_F(y, -b, k, h, mu, sigma, c, nargout, reg)
        Gh = lambda x, c, nargout: px.mean_zero_indicator(x, (m, n, k), c, nargout)
        model = E.MODEL_VAR_TV if reg == "tv" else E.MODEL_VAR_CURV
    xr, yr, phr, dhr = cp_ref(Fh, Gh, A, x0, y0, 1.0, tau, sigma, iters, 0.0)
    eng = E.Engine(model, m, n, k=k, bc=bc, h=h, omega=omega, mu=mu)
    eng.set_images(np.concatenate([imgs[i].reshape(-1, order="F") for i in range(k)]))
    eng.set_primal(x0)
    eng.linearize()
    eng.set_dual(y0)
    if fam == "mf":
        eng.set_nu(nu)
    eng.solve(1.0, tau, sigma, iters, 0.0)
    x, y = eng.primal(), eng.dual()
    ph, dh = eng.history()
    eng.close()
    ex, ey = rel(x, xr), rel(y, yr)
    dev = np.maximum(np.abs(ph[:, :5] - phr[:, :5]) - 1e-9 * np.abs(phr[:, :5]), np.abs(dh[:, :5] - dhr[:, :5]) - 1e-9 * np.abs(dhr[:, :5]))
    eh = float(dev.max())                     # as the tests: |a - b| <= atol + 1e-9 |b|, atol = 1e-9
    ok = ex < 1e-10 and ey < 1e-10 and eh < 1e-9
    return (dict(case=t, fam=fam, reg=reg, bc=bc, m=m, n=n, k=k, h=h, waves=waves, x=ex, y=ey, hist=eh, hist_col=int(np.argmax(dev.max(axis=0))), ok=bool(ok)))

