"""GPU parity of the mf_nn (NN-TV / NN-curvature) hot path: HIP engine through the C ABI vs the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-10   # north_star: fp64 within 1e-10 relative on the displacement field and final energy


def _setup(m, n, k, reg, bc, amp=1.5):
    from ma_amd import synth
    from oracle import drivers as od
    imgs = synth.dynamic_test_image(m, n, k).numpy()
    u0 = synth.smooth_displacement(m, n, k, amp=amp).numpy().reshape(k, 2, n * m)
    u0_ref = np.transpose(u0, (2, 1, 0))
    omega = [0.0, float(m), 0.0, float(n)]
    h = (1.0, 1.0)
    A, b, T_u, gs = od.mf_nn_problem([imgs[i] for i in range(k)], u0_ref, h, omega, reg, bc)
    return imgs, u0, omega, h, A, b, T_u, gs


def _rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.mark.parametrize("reg,bc,m,n,k", [("tv", "neumann", 48, 40, 3), ("tv", "linear", 33, 50, 4),
                                          ("curvature", "dirichlet", 40, 36, 3), ("curvature", "neumann", 70, 55, 5),
                                          ("tv", "neumann", 64, 64, 16)])
def test_solve_matches_oracle(reg, bc, m, n, k):
    from ma_amd import engine as E
    from oracle import drivers as od
    from oracle.chambolle_pock import chambolle_pock as cp_ref
    imgs, u0, omega, h, A, b, T_u, gs = _setup(m, n, k, reg, bc)
    mu = 0.2 if reg == "tv" else 20.0
    nu, tau, sigma = 2.0 * k, 0.3 if reg == "tv" else 0.12, 0.3 if reg == "tv" else 0.12
    kmn = k * m * n
    ncomp = 6 if reg == "tv" else 4
    rng = np.random.default_rng(5)
    x0 = np.concatenate([u0.reshape(-1), 0.5 * T_u.reshape(-1, order="F")])
    y0 = 0.01 * rng.standard_normal(ncomp * kmn)
    Fh = lambda y, c, nargout: od.mf_nn_F(y, b, k, h, mu, nu, sigma, c, nargout, reg)
    Gh = lambda x, c, nargout: od.mf_nn_G(x, (m, n, k), c, nargout)
    iters = 20
    xr, yr, phr, dhr = cp_ref(Fh, Gh, A, x0, y0, 1.0, tau, sigma, iters, 1e-3)

    model = E.MODEL_MF_NN_TV if reg == "tv" else E.MODEL_MF_NN_CURV
    eng = E.Engine(model, m, n, k=k, bc=bc, h=h, omega=omega, mu=mu)
    eng.set_images(np.concatenate([imgs[i].reshape(-1, order="F") for i in range(k)]))
    eng.set_primal(x0)
    eng.linearize()
    Tg, gg, bg = eng.get_linearization()
    assert _rel(bg, b) < 1e-14
    assert _rel(Tg, T_u.reshape(-1, order="F")) < 1e-14
    eng.set_dual(y0)
    eng.set_nu(nu)
    res = eng.solve(1.0, tau, sigma, iters, 1e-3)
    x, y = eng.primal(), eng.dual()
    ph, dh = eng.history()
    eng.close()
    assert ph.shape == phr.shape and dh.shape == dhr.shape
    print("x", _rel(x, xr), "y", _rel(y, yr))
    for c in range(7):
        print("col", c, np.max(np.abs(ph[:, c] - phr[:, c])), np.max(np.abs(phr[:, c])),
              np.max(np.abs(dh[:, c] - dhr[:, c])), np.max(np.abs(dhr[:, c])))
    assert _rel(x[: 2 * kmn], xr[: 2 * kmn]) < TOL      # displacement field
    assert _rel(x, xr) < TOL
    assert _rel(y, yr) < TOL
    assert abs(ph[-1, 0] - phr[-1, 0]) <= TOL * abs(phr[-1, 0])          # final energy
    np.testing.assert_allclose(ph[:, :5], phr[:, :5], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(dh[:, :5], dhr[:, :5], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(ph[:, 5:], phr[:, 5:], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(dh[:, 5:], dhr[:, 5:], rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("reg,k", [("tv", 16), ("tv", 5), ("curvature", 4)])
def test_iterate_in_parts_is_bitwise_iterate(reg, k):
    """cp_iterate_begin / cp_iterate_steps (twice) / cp_iterate_end == cp_iterate; the open iteration locks the state"""
    from ma_amd import engine as E
    m, n = 40, 36
    imgs, u0, omega, h, A, b, T_u, gs = _setup(m, n, k, reg, "neumann")
    model = E.MODEL_MF_NN_TV if reg == "tv" else E.MODEL_MF_NN_CURV
    mu = 0.2 if reg == "tv" else 20.0
    x0 = np.concatenate([u0.reshape(-1), 0.5 * T_u.reshape(-1, order="F")])
    out = []
    for parts in (None, (4, 7)):
        eng = E.Engine(model, m, n, k=k, bc="neumann", h=h, omega=omega, mu=mu)
        eng.set_images(np.concatenate([imgs[i].reshape(-1, order="F") for i in range(k)]))
        eng.set_primal(x0)
        eng.linearize()
        eng.set_nu(2.0 * k)
        if parts is None:
            eng.iterate(1.0, 0.2, 0.2, 11)
        else:
            eng.iterate_begin(1.0, 0.2, 0.2, 11)
            eng.iterate_steps(parts[0])
            with pytest.raises(E.CpEngineError, match="badState"):
                eng.primal()
            eng.iterate_steps(parts[1])
            with pytest.raises(E.CpEngineError, match="badArg"):
                eng.iterate_steps(1)
            eng.iterate_end()
        eng.sync()
        out.append((eng.primal(), eng.dual()))
        eng.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


@pytest.mark.parametrize("reg,bc,m,n,k", [("tv", "neumann", 5, 3, 2), ("tv", "linear", 3, 6, 2), ("tv", "neumann", 2, 2, 16),
                                          ("tv", "neumann", 130, 3, 16), ("tv", "neumann", 7, 9, 1),
                                          ("curvature", "neumann", 3, 4, 2), ("curvature", "dirichlet", 2, 67, 1)])
def test_extreme_shapes(reg, bc, m, n, k):
    """smallest grids (2 pixels per axis), single image (B*L = 0), strips narrower than a wave, K at its maximum"""
    from ma_amd import engine as E
    from oracle import drivers as od
    from oracle.chambolle_pock import chambolle_pock as cp_ref
    rng = np.random.default_rng(m * 100 + n * 10 + k)
    imgs = rng.random((k, m, n))
    u0 = 0.3 * rng.standard_normal((k, 2, n * m))
    u0_ref = np.transpose(u0, (2, 1, 0))
    omega = [0.0, float(m), 0.0, float(n)]
    h = (1.0, 1.0)
    A, b, T_u, gs = od.mf_nn_problem([imgs[i] for i in range(k)], u0_ref, h, omega, reg, bc)
    mu = 0.2 if reg == "tv" else 2.0
    nu, tau, sigma = 0.5 * k, 0.2, 0.2
    kmn = k * m * n
    ncomp = 6 if reg == "tv" else 4
    x0 = np.concatenate([u0.reshape(-1), 0.5 * T_u.reshape(-1, order="F")])
    y0 = 0.01 * rng.standard_normal(ncomp * kmn)
    Fh = lambda y, c, nargout: od.mf_nn_F(y, b, k, h, mu, nu, sigma, c, nargout, reg)
    Gh = lambda x, c, nargout: od.mf_nn_G(x, (m, n, k), c, nargout)
    xr, yr, phr, dhr = cp_ref(Fh, Gh, A, x0, y0, 1.0, tau, sigma, 12, 1e-3)
    eng = E.Engine(E.MODEL_MF_NN_TV if reg == "tv" else E.MODEL_MF_NN_CURV, m, n, k=k, bc=bc, h=h, omega=omega, mu=mu)
    eng.set_images(np.concatenate([imgs[i].reshape(-1, order="F") for i in range(k)]))
    eng.set_primal(x0)
    eng.linearize()
    eng.set_dual(y0)
    eng.set_nu(nu)
    eng.solve(1.0, tau, sigma, 12, 1e-3)
    x, y = eng.primal(), eng.dual()
    ph, dh = eng.history()
    eng.close()
    assert ph.shape == phr.shape
    np.testing.assert_allclose(ph[:, :5], phr[:, :5], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(dh[:, :5], dhr[:, :5], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(ph[:, 5:], phr[:, 5:], rtol=0, atol=1e-6)
    np.testing.assert_allclose(dh[:, 5:], dhr[:, 5:], rtol=0, atol=1e-8)
    assert _rel(x, xr) < TOL and _rel(y, yr) < TOL


@pytest.mark.parametrize("case,k", [("identical_frames", 6), ("zero_dual", 4), ("nu_small", 5), ("nu_large", 5),
                                    ("two_groups", 16)])
def test_degenerate_spectra(case, k):
    """Gram matrices the K x K eigensolve finds hard: repeated eigenvalues (identical frames: rank-1 y3 / B*L = 0; two groups
    of identical frames: rank 2 with K = 16), an exactly zero matrix (y0 = 0 and L0 = 0 in the first row), every singular value
    clipped (tiny nu) and none (huge nu: the point is inside the l1-ball, NNE/l1ball_projection.m:28). Eigenvectors are
    not unique there, W = I - V diag(d) V' is (d is constant on an eigenspace): iterates and histories must match the oracle."""
    from ma_amd import engine as E
    from oracle import drivers as od
    from oracle.chambolle_pock import chambolle_pock as cp_ref
    m, n, reg, bc = 24, 20, "tv", "neumann"
    rng = np.random.default_rng(17 + k)
    base = rng.random((2, m, n))
    if case == "identical_frames":
        imgs = np.repeat(base[:1], k, axis=0)
    elif case == "two_groups":
        imgs = np.stack([base[i % 2] for i in range(k)])
    else:
        imgs = rng.random((k, m, n))
    u0 = 0.3 * rng.standard_normal((k, 2, n * m))
    if case in ("identical_frames", "two_groups"):
        u0 = np.stack([u0[i % 2] if case == "two_groups" else u0[0] for i in range(k)])
    u0_ref = np.transpose(u0, (2, 1, 0))
    omega = [0.0, float(m), 0.0, float(n)]
    h = (1.0, 1.0)
    A, b, T_u, gs = od.mf_nn_problem([imgs[i] for i in range(k)], u0_ref, h, omega, reg, bc)
    mu, tau, sigma = 0.2, 0.2, 0.2
    nu = {"nu_small": 1e-6, "nu_large": 1e6}.get(case, 0.5 * k)
    kmn = k * m * n
    zero = case == "zero_dual"
    x0 = np.concatenate([u0.reshape(-1), (0.0 if zero else 0.5) * T_u.reshape(-1, order="F")])
    y0 = np.zeros(6 * kmn) if zero else 0.01 * rng.standard_normal(6 * kmn)
    if case in ("identical_frames", "two_groups"):      # keep the dual in the same symmetry class
        yb = 0.01 * rng.standard_normal((6, 2, m * n))
        y0 = np.concatenate([np.concatenate([yb[0, i % 2] for i in range(k)]),
                             np.concatenate([np.concatenate([yb[1 + c, i % 2] for c in range(4)]) for i in range(k)]),
                             np.concatenate([yb[5, i % 2] for i in range(k)])])
        if case == "identical_frames":
            y0 = np.concatenate([np.tile(yb[0, 0], k), np.tile(np.concatenate([yb[1 + c, 0] for c in range(4)]), k),
                                 np.tile(yb[5, 0], k)])
    Fh = lambda y, c, nargout: od.mf_nn_F(y, b, k, h, mu, nu, sigma, c, nargout, reg)
    Gh = lambda x, c, nargout: od.mf_nn_G(x, (m, n, k), c, nargout)
    iters = 14
    xr, yr, phr, dhr = cp_ref(Fh, Gh, A, x0, y0, 1.0, tau, sigma, iters, 1e-3)
    eng = E.Engine(E.MODEL_MF_NN_TV, m, n, k=k, bc=bc, h=h, omega=omega, mu=mu)
    eng.set_images(np.concatenate([imgs[i].reshape(-1, order="F") for i in range(k)]))
    eng.set_primal(x0)
    eng.linearize()
    eng.set_dual(y0)
    eng.set_nu(nu)
    eng.solve(1.0, tau, sigma, iters, 1e-3)
    x, y = eng.primal(), eng.dual()
    ph, dh = eng.history()
    eng.close()
    assert np.isfinite(x).all() and np.isfinite(y).all() and np.isfinite(ph).all() and np.isfinite(dh).all()
    assert ph.shape == phr.shape
    scale = max(np.linalg.norm(yr), 1e-300)
    assert _rel(x, xr) < TOL
    assert np.linalg.norm(y - yr) / scale < 1e-9 or np.linalg.norm(y - yr) < 1e-12
    np.testing.assert_allclose(ph[:, :5], phr[:, :5], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(dh[:, :5], dhr[:, :5], rtol=1e-9, atol=1e-9)
    # the constraint columns: (sum s - nu)/nu amplifies rounding by 1/nu for a tiny nu
    np.testing.assert_allclose(ph[:, 5:], phr[:, 5:], rtol=1e-6, atol=1e-6 if case != "nu_small" else 1e-3)
    np.testing.assert_allclose(dh[:, 5:], dhr[:, 5:], rtol=1e-6, atol=1e-8)


def test_long_run_against_c_restatement():
    """300 iterations (warm-started eigensolves with periodic cold restarts, lazy W, predicted means) stay within the
    north_star tolerance of the C restatement"""
    from ma_amd import engine as E
    from oracle import c_oracle
    m, n, k = 96, 80, 16
    imgs, u0, omega, h, A, b, T_u, gs = _setup(m, n, k, "tv", "neumann")
    mu, nu, tau, sigma = 0.2, 1.2 * k, 0.3, 0.3
    kmn = k * m * n
    x0 = np.concatenate([u0.reshape(-1), 0.5 * T_u.reshape(-1, order="F")])
    y0 = np.zeros(6 * kmn)
    g = np.concatenate([gs[i].reshape(-1, order="F") for i in range(k)])
    iters = 300
    xr, yr, phr, dhr = c_oracle.nntv_chambolle_pock(m, n, k, "neumann", h, mu, nu, 1.0, tau, sigma, iters, None, g, b, x0, y0)
    eng = E.Engine(E.MODEL_MF_NN_TV, m, n, k=k, bc="neumann", h=h, omega=omega, mu=mu)
    eng.set_images(np.concatenate([imgs[i].reshape(-1, order="F") for i in range(k)]))
    eng.set_primal(x0)
    eng.linearize()
    eng.set_nu(nu)
    eng.solve(1.0, tau, sigma, iters, 0.0)
    x, y = eng.primal(), eng.dual()
    ph, dh = eng.history()
    eng.close()
    assert ph.shape == phr.shape == (iters + 1, 7)
    np.testing.assert_allclose(ph[:, 0], phr[:, 0], rtol=1e-10)
    np.testing.assert_allclose(dh[:, 0], dhr[:, 0], rtol=1e-10)
    assert _rel(x, xr) < TOL and _rel(y, yr) < TOL


@pytest.mark.parametrize("m,n,fp32,k", [(70, 37, False, 16), (33, 8, False, 16), (130, 21, False, 16), (64, 16, True, 16),
                                          (70, 37, False, 10), (130, 21, False, 7), (64, 19, False, 13), (70, 21, True, 5)])
def test_fused_tv_wave_hand_over_equals_the_recomputation(monkeypatch, m, n, fp32, k):
    """k_mf_fused_tv with 8 waves per workgroup hands the own-pixel result of column j+1 to the wave of column j through LDS
    (cp_mfnn_fused.cuh: hs_give / hs_take) instead of recomputing it from y_n: the same iterates and histories as the 4-wave
    instantiation, which recomputes (the handed-over numbers are bitwise the recomputed ones; the partial sums are grouped by
    workgroup, so means and Gram matrices differ in the last bit: 1e-13), at sizes with partial segments, column groups that end inside a workgroup
    and a single column group, for K = 16 and the K < 16 instantiations (KA = 8, 12, 16); 60 iterations with the stop rule armed"""
    from ma_amd import engine as E, synth
    imgs = synth.dynamic_test_image(m, n, k).numpy()
    u0 = synth.smooth_displacement(m, n, k, amp=1.0).numpy().reshape(-1)
    omega = [0.0, float(m), 0.0, float(n)]

    def run(waves):
        monkeypatch.setenv("CP_TUNE_FUSED_WAVES", waves)
        eng = E.Engine(E.MODEL_MF_NN_TV, m, n, k=k, bc="neumann", h=(1.0, 1.0), omega=omega, mu=0.2, flags=E.FLAG_FP32 if fp32 else 0)
        eng.set_images(np.concatenate([imgs[i].reshape(-1, order="F") for i in range(k)]))
        eng.set_primal(np.concatenate([u0, np.zeros(k * m * n)]))
        eng.linearize()
        eng.set_nu(0.9 * float(np.sum(eng.singular_values(0))))
        est, _ = eng.norm_estimate()
        tau = float(np.sqrt(0.99 / est ** 2))
        res = eng.solve(1.0, tau, tau, 60, 1e-2)
        out = (eng.primal(), eng.dual()) + tuple(eng.history()) + (res.iterations,)
        assert (", 8, false>" in eng.dominant_kernel()) == (waves == "8")
        assert eng.handover_fallbacks() == 0          # no wave ever gave up on the LDS hand-over (cp_handover_fallbacks)
        eng.close()
        return out

    a, b = run("8"), run("4")
    assert a[4] == b[4]
    tol = 1e-6 if fp32 else 1e-13
    for u, v in zip(a[:2], b[:2]):
        assert np.linalg.norm(u - v) <= tol * np.linalg.norm(v)
    for u, v in zip(a[2:4], b[2:4]):
        np.testing.assert_allclose(u, v, rtol=1e3 * tol, atol=10 * tol)


@pytest.mark.parametrize("m,n,k,reg,bc", [(512, 24, 16, "tv", "neumann"), (576, 19, 11, "tv", "neumann"),
                                           (512, 24, 16, "curvature", "dirichlet"), (520, 19, 7, "curvature", "neumann")])
def test_forced_8wave_kernels_against_c_restatement(monkeypatch, m, n, k, reg, bc):
    """The 8-wave instantiations (k_mf_fused_tv with the LDS hand-over, k_mf_fused_curv with 8 columns per workgroup), forced
    at a size with 8-9 segments per column, directly against the C restatement over 120 iterations (they
    were only compared with the 4-wave kernels): K = 16 and a K < 16 instantiation, ragged last segment, a column group that
    ends inside a workgroup; the hand-over never falls back."""
    from ma_amd import engine as E
    from oracle import c_oracle
    monkeypatch.setenv("CP_TUNE_FUSED_WAVES", "8")
    monkeypatch.setenv("CP_TUNE_CURV_WAVES", "8")
    imgs, u0, omega, h, A, b, T_u, gs = _setup(m, n, k, reg, bc)
    mu = 0.2 if reg == "tv" else 20.0
    nu = 1.2 * k
    tau = sigma = 0.3 if reg == "tv" else 0.12
    kmn = k * m * n
    x0 = np.concatenate([u0.reshape(-1), 0.5 * T_u.reshape(-1, order="F")])
    y0 = np.zeros((6 if reg == "tv" else 4) * kmn)
    g = np.concatenate([gs[i].reshape(-1, order="F") for i in range(k)])
    iters = 120
    solve = c_oracle.nntv_chambolle_pock if reg == "tv" else c_oracle.nncurv_chambolle_pock
    xr, yr, phr, dhr = solve(m, n, k, bc, h, mu, nu, 1.0, tau, sigma, iters, None, g, b, x0, y0)
    eng = E.Engine(E.MODEL_MF_NN_TV if reg == "tv" else E.MODEL_MF_NN_CURV, m, n, k=k, bc=bc, h=h, omega=omega, mu=mu)
    name = eng.dominant_kernel()
    assert name.startswith("k_mf_fused_tv<double, false") and name.endswith(", 1, 8, false>") if reg == "tv" else name.endswith(", 8>")
    eng.set_images(np.concatenate([imgs[i].reshape(-1, order="F") for i in range(k)]))
    eng.set_primal(x0)
    eng.linearize()
    eng.set_nu(nu)
    eng.solve(1.0, tau, sigma, iters, 0.0)
    x, y = eng.primal(), eng.dual()
    ph, dh = eng.history()
    assert eng.handover_fallbacks() == 0
    eng.close()
    assert ph.shape == phr.shape == (iters + 1, 7)
    np.testing.assert_allclose(ph[:, 0], phr[:, 0], rtol=1e-10)
    np.testing.assert_allclose(dh[:, 0], dhr[:, 0], rtol=1e-10)
    assert _rel(x, xr) < TOL and _rel(y, yr) < TOL


@pytest.mark.parametrize("reg,bc,m,n,k", [("tv", "neumann", 40, 36, 20), ("tv", "linear", 33, 19, 17), ("tv", "neumann", 70, 21, 32),
                                          ("curvature", "dirichlet", 40, 36, 20), ("curvature", "neumann", 37, 50, 32)])
def test_more_than_16_images_against_oracle(reg, bc, m, n, k):
    """K > 16 (VERDICT r2 missing 1): the reference has no limit on the number of images (NNE/nuclear_norm_constraint.m:34-37,
    the kidney demo's k = size(data, 3)). 17..32 images run on the general path -- split passes built for 32 images, 2 x 2
    MFMA tiles per Gram matrix, 32 x 32 Jacobi (cp_mfnn_big.cuh) -- and must match the numpy oracle like every other size:
    operators' singular values, power iteration, 60 iterations with warm-started eigensolves, histories, stop rule armed."""
    from ma_amd import engine as E
    from oracle import drivers as od
    from oracle.chambolle_pock import chambolle_pock as cp_ref
    imgs, u0, omega, h, A, b, T_u, gs = _setup(m, n, k, reg, bc)
    mu = 0.2 if reg == "tv" else 20.0
    nu, tau = 1.5 * k, 0.3 if reg == "tv" else 0.12
    kmn = k * m * n
    ncomp = 6 if reg == "tv" else 4
    rng = np.random.default_rng(11)
    x0 = np.concatenate([u0.reshape(-1), 0.5 * T_u.reshape(-1, order="F")])
    y0 = 0.01 * rng.standard_normal(ncomp * kmn)
    Fh = lambda y, c, nargout: od.mf_nn_F(y, b, k, h, mu, nu, tau, c, nargout, reg)
    Gh = lambda x, c, nargout: od.mf_nn_G(x, (m, n, k), c, nargout)
    iters = 60
    xr, yr, phr, dhr = cp_ref(Fh, Gh, A, x0, y0, 1.0, tau, tau, iters, 1e-3)
    eng = E.Engine(E.MODEL_MF_NN_TV if reg == "tv" else E.MODEL_MF_NN_CURV, m, n, k=k, bc=bc, h=h, omega=omega, mu=mu)
    assert eng.dominant_kernel() == f"k_mf_dual<double, 0, {0 if reg == 'tv' else 1}, 32>"
    eng.set_images(np.concatenate([imgs[i].reshape(-1, order="F") for i in range(k)]))
    eng.set_primal(x0)
    eng.linearize()
    sv = eng.singular_values(0)                                  # nu_0 of the drivers: svd(B * T(u0)), NNE/mf_nn_tv...:173-175
    Tc = T_u.reshape(m * n, k, order="F")
    sv_ref = np.linalg.svd(Tc - Tc.mean(axis=1, keepdims=True), compute_uv=False)
    np.testing.assert_allclose(sv[:k - 1], sv_ref[:k - 1], rtol=1e-10)
    assert sv[k - 1] <= 1e-9 * sv[0]
    est, n_power = eng.norm_estimate()
    from oracle import operators as oo
    e_ref = oo.matrix_norm(A, x0=od.normest_start(A.shape[1]))   # NNE/matrix_norm.m with the engine's deterministic start vector
    assert n_power == len(e_ref) and abs(est - e_ref[-1]) <= 1e-11 * e_ref[-1]
    eng.set_dual(y0)
    eng.set_nu(nu)
    res = eng.solve(1.0, tau, tau, iters, 1e-3)
    x, y = eng.primal(), eng.dual()
    ph, dh = eng.history()
    eng.close()
    assert ph.shape == phr.shape and dh.shape == dhr.shape and res.iterations == phr.shape[0] - 1
    assert _rel(x, xr) < TOL
    assert _rel(y, yr) < 1e-9
    np.testing.assert_allclose(ph[:, :5], phr[:, :5], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(dh[:, :5], dhr[:, :5], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(ph[:, 5:], phr[:, 5:], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(dh[:, 5:], dhr[:, 5:], rtol=1e-6, atol=1e-8)


@pytest.mark.parametrize("model,bc,m,n,k,fp32", [("nn", "dirichlet", 2, 67, 1, False), ("nn", "neumann", 70, 55, 5, False),
                                                  ("nn", "dirichlet", 130, 9, 3, False), ("nn", "neumann", 640, 7, 2, False),
                                                  ("nn", "dirichlet", 258, 35, 4, True), ("var", "dirichlet", 132, 33, 6, False),
                                                  ("var", "neumann", 516, 5, 3, True)])
def test_curv_y2_two_rows_per_lane(monkeypatch, model, bc, m, n, k, fp32):
    """k_curv_y2_r2 (two rows per lane, 16-byte accesses; the default from 512 even rows on) forced at small and ragged even
    sizes -- a lone pair of rows, a wave with 3 / 35 live lanes, a second 512-row block of 128 rows, strips narrower than the
    load group -- for the NN-curvature and the variance-curvature model, both boundary conditions, fp64 and fp32 storage:
    against the one-row kernel (same arithmetic per pixel; the partial sums associate differently) and, for NN-curvature in
    fp64, against the numpy oracle."""
    from ma_amd import engine as E, synth
    from oracle import drivers as od
    from oracle.chambolle_pock import chambolle_pock as cp_ref
    rng = np.random.default_rng(11 * m + n)
    imgs = synth.dynamic_test_image(m, n, k).numpy() if min(m, n) >= 8 else rng.random((k, m, n))
    u0 = 0.4 * rng.standard_normal((k, 2, n * m))
    omega = [0.0, float(m), 0.0, float(n)]
    h = (1.0, 1.0)
    mu = 20.0 if model == "nn" else 5.0
    tau = sigma = 0.1
    iters = 25
    kmn = k * m * n
    flags = E.FLAG_FP32 if fp32 else 0
    out = {}
    for r2 in ("0", "1"):
        monkeypatch.setenv("CP_TUNE_Y2_R2", r2)
        eng = E.Engine(E.MODEL_MF_NN_CURV if model == "nn" else E.MODEL_VAR_CURV, m, n, k=k, bc=bc, h=h, omega=omega, mu=mu, flags=flags)
        eng.set_images(np.concatenate([imgs[i].reshape(-1, order="F") for i in range(k)]))
        x0 = np.concatenate([u0.reshape(-1), np.zeros(kmn)]) if model == "nn" else u0.reshape(-1)
        eng.set_primal(x0)
        eng.linearize()
        if model == "nn":
            x0[2 * kmn:] = 0.5 * eng.get_linearization()[0]
            eng.set_primal(x0)
            eng.linearize()
            eng.set_nu(1.5 * k)
        y0 = 0.01 * np.random.default_rng(3).standard_normal((4 if model == "nn" else 3) * kmn)
        eng.set_dual(y0)
        eng.solve(1.0, tau, sigma, iters, 0.0)
        out[r2] = (eng.primal(), eng.dual(), *eng.history(), x0, y0)
        eng.close()
    tol = 2e-5 if fp32 else 1e-12
    for a, b in zip(out["0"][:4], out["1"][:4]):
        np.testing.assert_allclose(b, a, rtol=tol, atol=tol * max(1.0, np.abs(a).max()))
    if model == "nn" and not fp32:
        u0_ref = np.transpose(u0, (2, 1, 0))
        A, b, T_u, gs = od.mf_nn_problem([imgs[i] for i in range(k)], u0_ref, h, omega, "curvature", bc)
        x0, y0 = out["1"][4], out["1"][5]
        Fh = lambda y, c, nargout: od.mf_nn_F(y, b, k, h, mu, 1.5 * k, sigma, c, nargout, "curvature")
        Gh = lambda x, c, nargout: od.mf_nn_G(x, (m, n, k), c, nargout)
        xr, yr, phr, dhr = cp_ref(Fh, Gh, A, x0, y0, 1.0, tau, sigma, iters, 0.0)
        assert _rel(out["1"][0], xr) < TOL and _rel(out["1"][1], yr) < TOL
        np.testing.assert_allclose(out["1"][2][:, :5], phr[:, :5], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(out["1"][3][:, :5], dhr[:, :5], rtol=1e-9, atol=1e-9)
