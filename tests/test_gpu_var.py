"""GPU parity of the variance models (NNE/var_tv_registration_no_ref_ml.m, NNE/var_curvature_registration_no_ref_ml.m):
HIP engine through the C ABI vs the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _rel(a, b):
    return np.linalg.norm(np.ravel(a) - np.ravel(b)) / max(np.linalg.norm(np.ravel(b)), 1e-300)


@pytest.mark.parametrize("reg,bc,m,n,k,mu", [("tv", "neumann", 48, 40, 3, 0.1), ("tv", "linear", 33, 50, 4, 0.1),
                                             ("curvature", "dirichlet", 40, 36, 3, 5.0),
                                             ("curvature", "neumann", 70, 55, 5, 5.0), ("tv", "neumann", 64, 64, 16, 0.1),
                                             # more than 16 images: the variance models have no K x K spectral work (limit 64)
                                             ("tv", "neumann", 40, 33, 21, 0.1), ("curvature", "dirichlet", 36, 40, 18, 5.0)])
def test_var_solve_matches_oracle(reg, bc, m, n, k, mu):
    from ma_amd import engine as E, synth
    from oracle import drivers as od, prox as px
    from oracle.chambolle_pock import chambolle_pock as cp_ref
    imgs = synth.dynamic_test_image(m, n, k).numpy()
    u0 = synth.smooth_displacement(m, n, k, amp=1.5).numpy().reshape(k, 2, n * m)
    u0_ref = np.transpose(u0, (2, 1, 0))
    omega = [0.0, float(m), 0.0, float(n)]
    h = (1.0, 1.0)
    A, b, T_u, gs = od.var_problem([imgs[i] for i in range(k)], u0_ref, h, omega, reg, bc)
    tau = sigma = 0.25 if reg == "tv" else 0.1
    kmn = k * m * n
    ncomp = 5 if reg == "tv" else 3
    rng = np.random.default_rng(7)
    x0 = u0.reshape(-1)
    y0 = 0.01 * rng.standard_normal(ncomp * kmn)
    Fh = lambda y, c, nargout: od.var_F(y, -b, k, h, mu, sigma, c, nargout, reg)
    Gh = lambda x, c, nargout: px.mean_zero_indicator(x, (m, n, k), c, nargout)
    iters = 20
    xr, yr, phr, dhr = cp_ref(Fh, Gh, A, x0, y0, 1.0, tau, sigma, iters, 1e-3)

    eng = E.Engine(E.MODEL_VAR_TV if reg == "tv" else E.MODEL_VAR_CURV, m, n, k=k, bc=bc, h=h, omega=omega, mu=mu)
    eng.set_images(np.concatenate([imgs[i].reshape(-1, order="F") for i in range(k)]))
    eng.set_primal(x0)
    eng.linearize()
    _, _, bg = eng.get_linearization()
    assert _rel(bg, b) < 1e-13                     # stored centred: B*b
    eng.set_dual(y0)
    est, npow = eng.norm_estimate(od.normest_start(2 * kmn))
    from oracle import operators as oo
    e_ref = oo.matrix_norm(A, od.normest_start(2 * kmn))
    assert npow == len(e_ref) and abs(est - e_ref[-1]) <= 1e-11 * e_ref[-1]
    eng.solve(1.0, tau, sigma, iters, 1e-3)
    x, y = eng.primal(), eng.dual()
    ph, dh = eng.history()
    eng.close()
    assert ph.shape == phr.shape and dh.shape == dhr.shape and ph.shape[1] == 6
    np.testing.assert_allclose(ph, phr, rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(dh, dhr, rtol=1e-9, atol=1e-10)
    assert _rel(x, xr) < TOL and _rel(y, yr) < TOL


@pytest.mark.parametrize("reg,bc,mu", [("tv", "neumann", 0.1), ("curvature", "neumann", 5.0)])
def test_var_ml_driver(reg, bc, mu):
    """multi-level driver end to end: x2 prolongation, outer re-linearisation, power iteration, stop rule"""
    from ma_amd import drivers as D, synth
    from oracle import drivers as od
    m, n, k = 70, 66, 4
    imgs = synth.dynamic_test_image(m, n, k).numpy()
    opt = dict(theta=1.0, maxIter=40, tol=1e-3, outerIter=[3, 2], mu=mu, bc=bc)
    keep_r, keep_g = {}, {}
    ur = od.var_registration_no_ref_ml([imgs[i] for i in range(k)], opt, reg, keep=keep_r)
    ug = D._var_ml([imgs[i] for i in range(k)], opt, reg, normest_x0=od.normest_start, keep=keep_g)
    assert set(keep_r) == set(keep_g)
    for key in sorted(keep_r):
        a, b = keep_g[key], keep_r[key]
        assert abs(a["norm_est"] - b["norm_est"]) <= 1e-11 * b["norm_est"] and a["n_power"] == b["n_power"]
        assert a["primal_history"].shape == b["primal_history"].shape
        np.testing.assert_allclose(a["primal_history"], b["primal_history"], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(a["dual_history"], b["dual_history"], rtol=1e-8, atol=1e-9)
    for lev in range(len(ur)):
        for o in range(len(ur[lev])):
            if ur[lev][o] is None:
                assert ug[lev][o] is None
                continue
            assert _rel(ug[lev][o], ur[lev][o]) < 1e-9, (lev, o)


def test_var_iterate_in_parts():
    from ma_amd import engine as E, synth
    m, n, k = 40, 36, 4
    imgs = synth.dynamic_test_image(m, n, k).numpy()
    u0 = synth.smooth_displacement(m, n, k, amp=1.5).numpy().reshape(-1)
    out = []
    for parts in (None, (3, 6)):
        eng = E.Engine(E.MODEL_VAR_TV, m, n, k=k, bc="neumann", omega=[0.0, float(m), 0.0, float(n)], mu=0.1)
        eng.set_images(np.concatenate([imgs[i].reshape(-1, order="F") for i in range(k)]))
        eng.set_primal(u0)
        eng.linearize()
        if parts is None:
            eng.iterate(1.0, 0.2, 0.2, 9)
        else:
            eng.iterate_begin(1.0, 0.2, 0.2, 9)
            eng.iterate_steps(parts[0])
            eng.iterate_steps(parts[1])
            eng.iterate_end()
        eng.sync()
        out.append((eng.primal(), eng.dual()))
        with pytest.raises(E.CpEngineError):
            eng.singular_values(1)
        eng.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


def test_single_level_drivers():
    """NNE/var_tv_registration_no_ref.m (swapped SSD prox arguments, :173) and NNE/mf_nn_tv_registration_no_ref.m"""
    from ma_amd import drivers as D, synth
    from oracle import drivers as od
    m, n, k = 72, 80, 3           # the _ml drivers would use 2 levels here
    imgs = [synth.dynamic_test_image(m, n, k).numpy()[i] for i in range(k)]
    opt = dict(theta=1.0, maxIter=40, tol=1e-3, outerIter=2, mu=0.1, bc="neumann")
    keep_r, keep_g = {}, {}
    ur = od.var_registration_no_ref_ml(imgs, opt, "tv", keep=keep_r, single_level=True)[0]
    ug = D.var_tv_registration_no_ref(imgs, opt, keep=keep_g)
    assert set(keep_g) == {(0, 0), (0, 1)} == set(keep_r)
    for key in keep_r:
        np.testing.assert_allclose(keep_g[key]["dual_history"], keep_r[key]["dual_history"], rtol=1e-8, atol=1e-9)
    for o in range(2):
        assert _rel(ug[o], ur[o]) < 1e-9
    # the swap matters: the _ml arithmetic gives a different iterate
    u_ml = od.var_registration_no_ref_ml(imgs, dict(opt, outerIter=[2, 2]), "tv", single_level=False)
    assert len(u_ml) == 2
    # NNE/mf_nn_tv_registration_no_ref.m: F1 = SAD(y1, b, 1, sigma), F2 = norm21(y2_j, mu, sigma) (:206-216)
    opt2 = dict(theta=1.0, maxIter=30, tol=1e-3, outerIter=2, mu=0.2, nu_factor=0.9, bc="neumann")
    keep_r, keep_g = {}, {}
    ur2, Lr2, SVr2 = od.mf_nn_registration_no_ref_ml(imgs, opt2, "tv", single_level=True, keep=keep_r)
    ug2, Lg2, SVg2 = D.mf_nn_tv_registration_no_ref(imgs, opt2, normest_x0=od.normest_start, keep=keep_g)
    for o in range(2):
        assert _rel(ug2[o], ur2[0][o]) < 1e-9 and _rel(Lg2[o], Lr2[0][o]) < 1e-9
        a, b = keep_g[(0, o)], keep_r[(0, o)]
        assert a["primal_history"].shape == b["primal_history"].shape
        np.testing.assert_allclose(a["primal_history"][:, :5], b["primal_history"][:, :5], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(a["dual_history"], b["dual_history"], rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(SVg2, SVr2[0], rtol=1e-9, atol=1e-10)
    # the _ml functional (pointwise 2x2 nuclear norm) on the same single level gives a different iterate
    u_nn, _, _ = od.mf_nn_registration_no_ref_ml(imgs, dict(opt2, outerIter=[2, 2], nu_factor=[0.9, 0.9]), "tv")
    assert len(u_nn) == 2


def test_var_tv_lazy_state_over_a_long_run(monkeypatch):
    """k_var_lazy_tv keeps y1 as alpha - beta (per-pixel beta, cp_var_fused.cuh): 300 iterations from a dual start that is not
    mean-free over the images, against the oracle and against the split passes (no lazy state) -- the offset must not
    drift, and the ssd_swap form of the single-level driver takes the same path"""
    from ma_amd import engine as E, synth
    from oracle import drivers as od, prox as px
    from oracle.chambolle_pock import chambolle_pock as cp_ref
    m, n, k, mu = 40, 36, 4, 0.1
    imgs = synth.dynamic_test_image(m, n, k).numpy()
    u0 = synth.smooth_displacement(m, n, k, amp=1.5).numpy().reshape(k, 2, n * m)
    u0_ref = np.transpose(u0, (2, 1, 0))
    omega, h = [0.0, float(m), 0.0, float(n)], (1.0, 1.0)
    A, b, T_u, gs = od.var_problem([imgs[i] for i in range(k)], u0_ref, h, omega, "tv", "neumann")
    tau = sigma = 0.25
    rng = np.random.default_rng(11)
    x0 = u0.reshape(-1)
    y0 = 0.05 * rng.standard_normal(5 * k * m * n) + 0.02
    iters = 300
    Fh = lambda y, c, nargout: od.var_F(y, -b, k, h, mu, sigma, c, nargout, "tv")
    Gh = lambda x, c, nargout: px.mean_zero_indicator(x, (m, n, k), c, nargout)
    xr, yr, phr, dhr = cp_ref(Fh, Gh, A, x0, y0, 1.0, tau, sigma, iters, 0.0)

    def run(flags=0):
        eng = E.Engine(E.MODEL_VAR_TV, m, n, k=k, bc="neumann", h=h, omega=omega, mu=mu, flags=flags)
        eng.set_images(np.concatenate([imgs[i].reshape(-1, order="F") for i in range(k)]))
        eng.set_primal(x0)
        eng.linearize()
        eng.set_dual(y0)
        eng.solve(1.0, tau, sigma, iters, 0.0)
        out = (eng.primal(), eng.dual()) + tuple(eng.history())
        eng.close()
        return out

    xl, yl, phl, dhl = run()
    assert _rel(xl, xr) < 1e-9 and _rel(yl, yr) < 1e-9
    np.testing.assert_allclose(phl, phr, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(dhl, dhr, rtol=1e-8, atol=1e-10)
    monkeypatch.setenv("CP_TUNE_VAR_SPLIT", "1")
    xs, ys, phs, dhs = run()
    assert _rel(xl, xs) < 1e-11 and _rel(yl, ys) < 1e-11
    np.testing.assert_allclose(phl, phs, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(dhl, dhs, rtol=1e-10, atol=1e-12)
    xs2, ys2, _, _ = run(E.FLAG_VAR_SSD_SWAP)
    monkeypatch.delenv("CP_TUNE_VAR_SPLIT")
    xl2, yl2, _, _ = run(E.FLAG_VAR_SSD_SWAP)
    assert _rel(xl2, xs2) < 1e-11 and _rel(yl2, ys2) < 1e-11


@pytest.mark.parametrize("m,n,k,fp32", [(70, 37, 5, False), (300, 21, 16, False), (129, 8, 3, False), (64, 3, 21, False),
                                        (70, 37, 5, True)])
def test_var_tv_wave_hand_over_equals_the_recomputation(monkeypatch, m, n, k, fp32):
    """k_var_lazy_tv with 8 waves per workgroup takes the neighbour column's own-pixel values from the next wave through LDS
    (hs_give / hs_take) instead of loading them again: same iterates and histories as the 4-wave instantiation (partial sums
    are grouped by workgroup: 1e-13), with partial segments, column groups ending inside a workgroup, fewer columns than
    waves; CP_TUNE_VAR_WAVES=8 forces the instantiation at these sizes and in fp32 storage"""
    from ma_amd import engine as E, synth
    imgs = synth.dynamic_test_image(m, n, k).numpy()
    u0 = synth.smooth_displacement(m, n, k, amp=1.0).numpy().reshape(-1)
    omega, h = [0.0, float(m), 0.0, float(n)], (1.0, 1.0)
    y0 = 0.05 * np.random.default_rng(5).standard_normal(5 * k * m * n) + 0.01

    def run(waves):
        monkeypatch.setenv("CP_TUNE_VAR_WAVES", waves)
        eng = E.Engine(E.MODEL_VAR_TV, m, n, k=k, bc="neumann", h=h, omega=omega, mu=0.1, flags=E.FLAG_FP32 if fp32 else 0)
        eng.set_images(np.concatenate([imgs[i].reshape(-1, order="F") for i in range(k)]))
        eng.set_primal(u0)
        eng.linearize()
        eng.set_dual(y0)
        res = eng.solve(1.0, 0.25, 0.25, 80, 1e-3)
        out = (eng.primal(), eng.dual()) + tuple(eng.history()) + (res.iterations,)
        eng.close()
        return out

    a, b = run("8"), run("4")
    assert a[4] == b[4]
    tol = 1e-6 if fp32 else 1e-13
    for u, v in zip(a[:2], b[:2]):
        assert np.linalg.norm(u - v) <= tol * np.linalg.norm(v)
    for u, v in zip(a[2:4], b[2:4]):
        np.testing.assert_allclose(u, v, rtol=1e3 * tol, atol=10 * tol)


@pytest.mark.parametrize("bc", ["dirichlet", "neumann"])
def test_var_curvature_two_pass_path_over_a_long_run(monkeypatch, bc):
    """k_var_lazy_curv + k_curv_y2 + k_var_curv_fin (cp_var_fused.cuh) against the oracle and the split passes: 250 iterations,
    dual start not mean-free over the images; dirichlet rows exercise the boundary term of the mean known in advance, the
    stop rule fires at the same iteration"""
    from ma_amd import engine as E, synth
    from oracle import drivers as od, prox as px
    from oracle.chambolle_pock import chambolle_pock as cp_ref
    m, n, k, mu = 37, 44, 5, 5.0
    imgs = synth.dynamic_test_image(m, n, k).numpy()
    u0 = synth.smooth_displacement(m, n, k, amp=1.5).numpy().reshape(k, 2, n * m)
    u0_ref = np.transpose(u0, (2, 1, 0))
    omega, h = [0.0, float(m), 0.0, float(n)], (1.0, 1.0)
    A, b, T_u, gs = od.var_problem([imgs[i] for i in range(k)], u0_ref, h, omega, "curvature", bc)
    tau = sigma = 0.1
    rng = np.random.default_rng(5)
    x0 = u0.reshape(-1)
    y0 = 0.05 * rng.standard_normal(3 * k * m * n) + 0.02
    iters, tol = 250, 2e-2
    Fh = lambda y, c, nargout: od.var_F(y, -b, k, h, mu, sigma, c, nargout, "curvature")
    Gh = lambda x, c, nargout: px.mean_zero_indicator(x, (m, n, k), c, nargout)
    xr, yr, phr, dhr = cp_ref(Fh, Gh, A, x0, y0, 1.0, tau, sigma, iters, tol)

    def run():
        eng = E.Engine(E.MODEL_VAR_CURV, m, n, k=k, bc=bc, h=h, omega=omega, mu=mu)
        eng.set_images(np.concatenate([imgs[i].reshape(-1, order="F") for i in range(k)]))
        eng.set_primal(x0)
        eng.linearize()
        eng.set_dual(y0)
        eng.solve(1.0, tau, sigma, iters, tol)
        out = (eng.primal(), eng.dual()) + tuple(eng.history())
        eng.close()
        return out

    xl, yl, phl, dhl = run()
    assert phl.shape == phr.shape                                  # same stopping iteration
    assert _rel(xl, xr) < 1e-9 and _rel(yl, yr) < 1e-9
    np.testing.assert_allclose(phl, phr, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(dhl, dhr, rtol=1e-8, atol=1e-10)
    monkeypatch.setenv("CP_TUNE_VAR_SPLIT", "1")
    xs, ys, phs, dhs = run()
    assert phs.shape == phl.shape and _rel(xl, xs) < 1e-11 and _rel(yl, ys) < 1e-11
    np.testing.assert_allclose(phl, phs, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(dhl, dhs, rtol=1e-10, atol=1e-12)
