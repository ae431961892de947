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
    m = int(rng.choice([5, 17, 33, 63, 64, 65, 100, 129, 130, 140]))
    n = int(rng.choice([3, 4, 5, 7, 8, 9, 15, 21, 33, 40, 66]))
    k = int(rng.integers(2, 17 if fam == "mf" else 24))
    h = (float(rng.choice([1.0, 0.5, 2.0])), float(rng.choice([1.0, 0.25, 1.5])))
    waves = str(rng.choice(["", "4", "8"]))
    for var in ("CP_TUNE_FUSED_WAVES", "CP_TUNE_VAR_WAVES"):
        os.environ.pop(var, None)
        if waves:
            os.environ[var] = waves
    os.environ.pop("CP_TUNE_Y2_R2", None)
    if rng.integers(2):                      # curvature models: the two-rows-per-lane y2 pass forced at even sizes below 512 rows
        os.environ["CP_TUNE_Y2_R2"] = "1"
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
    os.environ.pop("CP_TUNE_Y2_R2", None)
    ex, ey = rel(x, xr), rel(y, yr)
    dev = np.maximum(np.abs(ph[:, :5] - phr[:, :5]) - 1e-9 * np.abs(phr[:, :5]), np.abs(dh[:, :5] - dhr[:, :5]) - 1e-9 * np.abs(dhr[:, :5]))
    eh = float(dev.max())                     # as the tests: |a - b| <= atol + 1e-9 |b|, atol = 1e-9
    ok = ex < 1e-10 and ey < 1e-10 and eh < 1e-9
    return (dict(case=t, fam=fam, reg=reg, bc=bc, m=m, n=n, k=k, h=h, waves=waves, x=ex, y=ey, hist=eh, hist_col=int(np.argmax(dev.max(axis=0))), ok=bool(ok)))



def run_pair_case(t, rng):
    """one random pairwise TV-L1 / TV-L2 registration problem: one outer step (linearise at a random smooth u0, 20 CP
    iterations from a random dual) on the engine and on the numpy oracle; random strip width of the marching kernel"""
    from ma_amd import engine as E, synth
    from ma_amd.drivers import normalize
    from oracle import prox as px, operators as oo
    from oracle.chambolle_pock import chambolle_pock as cp_ref
    term = ("SAD", "SSD")[int(rng.integers(2))]
    m = int(rng.choice([9, 17, 40, 62, 63, 64, 65, 126, 127, 150]))
    n = int(rng.choice([5, 6, 7, 8, 9, 16, 17, 33, 50, 90]))
    strip = str(rng.choice(["", "4", "6", "8", "16"]))
    two = str(rng.choice(["", "1"]))          # k_pw_march2 (two rows per lane) forced at even sizes below its default threshold
    os.environ.pop("CP_TUNE_PW_STRIP", None)
    os.environ.pop("CP_TUNE_PW_TWO", None)
    if strip:
        os.environ["CP_TUNE_PW_STRIP"] = strip
    if two:
        os.environ["CP_TUNE_PW_TWO"] = two
    lam, tau = (4.0, 1.5) if term == "SAD" else (50.0, 4.0)
    fr = synth.dynamic_test_image(m, n, 2).numpy()
    if not np.isfinite(fr).all():
        fr = np.random.default_rng(200 + t).random((2, m, n))
    R, T = normalize(fr[0]), normalize(fr[1])
    h = (1.0, 1.0)
    sigma = (1 - 1e-4) / (8 * tau)
    K = oo.pairwise_gradient(m, n, h, 2)
    Gfun = px.SAD_registration if term == "SAD" else px.SSD_registration
    r2 = np.random.default_rng(300 + t)
    u0 = 0.5 * synth.smooth_displacement(m, n, 1, amp=1.0).numpy().reshape(-1)
    v0 = 0.3 * r2.standard_normal(4 * m * n)
    F = lambda v, c, nargout: px.TV_registration(v, sigma, c, nargout)
    G = lambda u, c, nargout: Gfun(u, u0, T, R, h, lam, tau, c, nargout)
    iters = 20
    xr, yr, phr, dhr = cp_ref(F, G, K, u0, v0, 1.0, tau, sigma, iters)
    eng = E.Engine(E.MODEL_TV_L1_REG if term == "SAD" else E.MODEL_TV_L2_REG, m, n, lam=lam)
    eng.set_images(T.reshape(-1, order="F"), R.reshape(-1, order="F"))
    eng.set_primal(u0)
    eng.set_dual(v0)
    eng.linearize()
    eng.solve(1.0, tau, sigma, iters, 0.0)
    x, y = eng.primal(), eng.dual()
    ph, dh = eng.history()
    eng.close()
    os.environ.pop("CP_TUNE_PW_STRIP", None)
    os.environ.pop("CP_TUNE_PW_TWO", None)
    nr = min(ph.shape[0], phr.shape[0])
    ex, ey = rel(x, xr), rel(y, yr)
    dp = float(np.max(np.abs(ph[:nr] - phr[:nr]) - 1e-9 * np.abs(phr[:nr])))       # atol as tests/test_gpu_pairwise.py
    dd = float(np.max(np.abs(dh[:nr] - dhr[:nr]) - 1e-9 * np.abs(dhr[:nr])))
    ok = ph.shape == phr.shape and ex < 1e-10 and ey < 1e-10 and dp < 1e-10 and dd < 1e-7
    return dict(case=t, term=term, m=m, n=n, strip=strip, x=ex, y=ey, hist_p=dp, hist_d=dd, ok=bool(ok))
