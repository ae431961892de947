"""fp32 mode of the pairwise and mf_nn models (CP_FLAG_FP32: float storage, fp64 arithmetic) against the fp64 mode and
the oracle: north_star tolerance 1e-4 relative on the displacement field and the final energy."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL32 = 1e-4


def _rel(a, b):
    return np.linalg.norm(np.ravel(a) - np.ravel(b)) / max(np.linalg.norm(np.ravel(b)), 1e-300)


def _pair(m, n):
    from ma_amd import synth
    from ma_amd.drivers import normalize
    fr = synth.dynamic_test_image(m, n, 2).numpy()
    return normalize(fr[0]), normalize(fr[1])


@pytest.mark.parametrize("model,lam,tau", [("TV_L2", 50.0, 4.0), ("TV_L1", 4.0, 1.5)])
def test_fp32_registration_matches_fp64_and_oracle(model, lam, tau):
    from ma_amd import engine as E
    from oracle import prox as px, operators as oo
    from oracle.chambolle_pock import chambolle_pock as cp_ref
    m, n = 64, 56
    R, T = _pair(m, n)
    sigma = (1 - 1e-4) / (8 * tau)
    mid = E.MODEL_TV_L2_REG if model == "TV_L2" else E.MODEL_TV_L1_REG
    out = {}
    for name, flags in (("f64", 0), ("f32", E.FLAG_FP32)):
        eng = E.Engine(mid, m, n, lam=lam, flags=flags)
        eng.set_images(T.reshape(-1, order="F"), R.reshape(-1, order="F"))
        eng.set_primal(np.zeros(2 * m * n))
        eng.set_dual(np.zeros(4 * m * n))
        eng.linearize()
        eng.solve(1.0, tau, sigma, 30, 1e-3)
        out[name] = (eng.primal(), eng.dual(), *eng.history())
        eng.close()
    K = oo.pairwise_gradient(m, n, (1.0, 1.0), 2)
    Gfun = px.SSD_registration if model == "TV_L2" else px.SAD_registration
    u0 = np.zeros(2 * m * n)
    F = lambda v, c, nargout: px.TV_registration(v, sigma, c, nargout)
    G = lambda u, c, nargout: Gfun(u, u0, T, R, (1.0, 1.0), lam, tau, c, nargout)
    xr, yr, phr, dhr = cp_ref(F, G, K, u0, np.zeros(4 * m * n), 1.0, tau, sigma, 30)
    x64, y64, ph64, dh64 = out["f64"]
    x32, y32, ph32, dh32 = out["f32"]
    assert ph32.shape == ph64.shape == phr.shape
    assert _rel(x64, xr) < 1e-10
    assert _rel(x32, xr) < TOL32 and _rel(y32, yr) < TOL32
    assert abs(ph32[-1, 0] - phr[-1, 0]) <= TOL32 * abs(phr[-1, 0])
    assert abs(dh32[-1, 0] - dhr[-1, 0]) <= TOL32 * abs(dhr[-1, 0])
    assert _rel(x32, x64) > 0.0                      # it really is a different storage precision
    # what comes back is exactly representable in fp32
    assert np.array_equal(x32, x32.astype(np.float32).astype(np.float64))


def test_fp32_denoise_and_batch():
    from ma_amd import engine as E
    m, n, B = 48, 40, 4
    R, T = _pair(m, n)
    rng = np.random.default_rng(1)
    noisy = np.clip(R + 0.2 * rng.standard_normal((m, n)), 0, 1)
    res = {}
    for flags in (0, E.FLAG_FP32):
        eng = E.Engine(E.MODEL_TV_L1_DENOISE, m, n, lam=1.5, flags=flags)
        eng.set_images(noisy.reshape(-1, order="F"))
        eng.solve(1.0, 0.02, (1 - 1e-4) / (8 * 0.02), 200, 1e-5)
        res[flags] = (eng.primal(), eng.history()[0])
        eng.close()
    assert res[0][1].shape == res[E.FLAG_FP32][1].shape
    assert _rel(res[E.FLAG_FP32][0], res[0][0]) < TOL32
    # batch of B problems in fp32 == the same problems one by one in fp32 (bitwise)
    Ts = np.concatenate([np.roll(T, s, axis=0).reshape(-1, order="F") for s in range(B)])
    Rs = np.tile(R.reshape(-1, order="F"), B)
    eng = E.Engine(E.MODEL_TV_L2_REG, m, n, lam=50.0, batch=B, flags=E.FLAG_FP32)
    eng.set_images(Ts, Rs)
    eng.set_primal(np.zeros(2 * m * n * B))
    eng.linearize()
    eng.solve(1.0, 4.0, (1 - 1e-4) / 32, 20, 1e-3)
    xb = eng.primal().reshape(B, -1)
    eng.close()
    for s in range(B):
        one = E.Engine(E.MODEL_TV_L2_REG, m, n, lam=50.0, flags=E.FLAG_FP32)
        one.set_images(Ts.reshape(B, -1)[s], Rs.reshape(B, -1)[s])
        one.set_primal(np.zeros(2 * m * n))
        one.linearize()
        one.solve(1.0, 4.0, (1 - 1e-4) / 32, 20, 1e-3)
        assert np.array_equal(one.primal(), xb[s])
        one.close()
    E.Engine(E.MODEL_VAR_TV, 16, 16, k=3, flags=E.FLAG_FP32).close()                            # variance models too (below)
    E.Engine(E.MODEL_MF_NN_TV, 16, 8, k=3, n_global=16, j_offset=0, flags=E.FLAG_FP32).close()   # slabs take fp32 storage (tests/test_gpu_slab.py)


@pytest.mark.parametrize("reg,bc,m,n,k", [("tv", "neumann", 48, 40, 3), ("tv", "linear", 33, 50, 4),
                                          ("curvature", "dirichlet", 40, 36, 3), ("tv", "neumann", 64, 64, 16)])
def test_fp32_mf_nn_matches_oracle(reg, bc, m, n, k):
    """mf_nn models in the fp32 storage mode: 40 iterations from a non-trivial start against the fp64 oracle."""
    from ma_amd import engine as E
    from oracle import drivers as od
    from oracle.chambolle_pock import chambolle_pock as cp_ref
    from test_gpu_mfnn import _setup
    imgs, u0, omega, h, A, b, T_u, gs = _setup(m, n, k, reg, bc)
    mu = 0.2 if reg == "tv" else 20.0
    nu, tau = 2.0 * k, 0.3 if reg == "tv" else 0.12
    sigma = tau
    kmn = k * m * n
    ncomp = 6 if reg == "tv" else 4
    rng = np.random.default_rng(5)
    x0 = np.concatenate([u0.reshape(-1), 0.5 * T_u.reshape(-1, order="F")])
    y0 = 0.01 * rng.standard_normal(ncomp * kmn)
    Fh = lambda y, c, nargout: od.mf_nn_F(y, b, k, h, mu, nu, sigma, c, nargout, reg)
    Gh = lambda x, c, nargout: od.mf_nn_G(x, (m, n, k), c, nargout)
    iters = 40
    xr, yr, phr, dhr = cp_ref(Fh, Gh, A, x0, y0, 1.0, tau, sigma, iters, 1e-3)
    model = E.MODEL_MF_NN_TV if reg == "tv" else E.MODEL_MF_NN_CURV
    eng = E.Engine(model, m, n, k=k, bc=bc, h=h, omega=omega, mu=mu, flags=E.FLAG_FP32)
    eng.set_images(np.concatenate([imgs[i].reshape(-1, order="F") for i in range(k)]))
    eng.set_primal(x0)
    eng.linearize()
    Tg, gg, bg = eng.get_linearization()
    assert _rel(bg, b) < 1e-6 and _rel(Tg, T_u.reshape(-1, order="F")) < 1e-6
    sv32 = eng.singular_values(0)
    ne32, _ = eng.norm_estimate(None)
    eng.set_dual(y0)
    eng.set_nu(nu)
    eng.solve(1.0, tau, sigma, iters, 1e-3)
    x, y = eng.primal(), eng.dual()
    ph, dh = eng.history()
    svL = eng.singular_values(1)
    eng.close()
    # the same quantities in the fp64 mode
    e64 = E.Engine(model, m, n, k=k, bc=bc, h=h, omega=omega, mu=mu)
    e64.set_images(np.concatenate([imgs[i].reshape(-1, order="F") for i in range(k)]))
    e64.set_primal(x0)
    e64.linearize()
    sv64 = e64.singular_values(0)
    ne64, _ = e64.norm_estimate(None)
    e64.close()
    print("fp32 x", _rel(x, xr), "u", _rel(x[:2 * kmn], xr[:2 * kmn]), "y", _rel(y, yr),
          "E", abs(ph[-1, 0] - phr[-1, 0]) / abs(phr[-1, 0]), abs(dh[-1, 0] - dhr[-1, 0]) / abs(dhr[-1, 0]))
    assert ph.shape == phr.shape
    assert _rel(x[: 2 * kmn], xr[: 2 * kmn]) < TOL32          # displacement field
    assert _rel(x, xr) < TOL32
    assert _rel(y, yr) < TOL32
    assert abs(ph[-1, 0] - phr[-1, 0]) <= TOL32 * abs(phr[-1, 0])   # final energy
    assert abs(dh[-1, 0] - dhr[-1, 0]) <= TOL32 * abs(dhr[-1, 0])
    assert _rel(x, xr) > 1e-12                                 # a different storage precision indeed
    assert np.array_equal(x, x.astype(np.float32).astype(np.float64))
    np.testing.assert_allclose(sv32, sv64, rtol=1e-5, atol=1e-6 * sv64[0])   # the mean-free block has one zero value
    assert abs(ne32 - ne64) <= 1e-6 * ne64
    Lr = xr[2 * kmn:].reshape(m * n, k, order="F")
    svr = np.linalg.svd(Lr - Lr.mean(axis=1, keepdims=True), compute_uv=False)
    np.testing.assert_allclose(svL, svr, rtol=1e-4, atol=1e-4 * svr[0])


def test_fp32_mf_nn_multilevel_driver():
    """mf_nn_tv_registration_no_ref_ml in the fp32 mode (prolongation, downsampling, power iteration, nu rule) against
    the fp64 mode of the same driver."""
    from ma_amd import synth
    from ma_amd.drivers import mf_nn_tv_registration_no_ref_ml
    k, m, n = 4, 80, 72
    imgs = synth.dynamic_test_image(m, n, k).numpy()
    optPara = dict(theta=1.0, maxIter=60, tol=1e-3, outerIter=[3, 2], mu=0.1, nu_factor=[0.9, 0.95], bc="linear", doPlots=False)
    u64, L64, SV64 = mf_nn_tv_registration_no_ref_ml([imgs[i] for i in range(k)], optPara)
    u32, L32, SV32 = mf_nn_tv_registration_no_ref_ml([imgs[i] for i in range(k)], optPara, fp32=True)
    ru = _rel(u32[-1][1], u64[-1][1])
    rl = _rel(L32[-1][1], L64[-1][1])
    print("driver fp32 vs fp64: u", ru, "L", rl)
    assert ru < 10 * TOL32 and rl < TOL32     # two levels, five re-linearisations: rounding differences compound
    np.testing.assert_allclose(SV32[-1], SV64[-1], rtol=1e-3, atol=1e-4 * SV64[-1].max())


@pytest.mark.parametrize("reg,bc,m,n,k,mu", [("tv", "neumann", 48, 40, 3, 0.1), ("tv", "linear", 33, 50, 4, 0.1),
                                             ("curvature", "dirichlet", 40, 36, 5, 5.0), ("curvature", "neumann", 70, 55, 3, 5.0)])
def test_fp32_variance_models_match_oracle(reg, bc, m, n, k, mu):
    """variance models in the fp32 storage mode (k_var_lazy_tv / k_var_lazy_curv<float>: fields as float, the per-pixel lazy
    fields, the arithmetic and the histories in double): 60 iterations from a dual start that is not mean-free over the
    images, against the fp64 oracle within north_star's 1e-4 (measured: ~1e-7)"""
    from ma_amd import engine as E, synth
    from oracle import drivers as od, prox as px
    from oracle.chambolle_pock import chambolle_pock as cp_ref
    imgs = synth.dynamic_test_image(m, n, k).numpy()
    u0 = synth.smooth_displacement(m, n, k, amp=1.5).numpy().reshape(k, 2, n * m)
    u0_ref = np.transpose(u0, (2, 1, 0))
    omega, h = [0.0, float(m), 0.0, float(n)], (1.0, 1.0)
    A, b, T_u, gs = od.var_problem([imgs[i] for i in range(k)], u0_ref, h, omega, reg, bc)
    tau = sigma = 0.25 if reg == "tv" else 0.1
    ncomp = 5 if reg == "tv" else 3
    rng = np.random.default_rng(3)
    x0 = u0.reshape(-1)
    y0 = 0.01 * rng.standard_normal(ncomp * k * m * n) + 0.005
    iters = 60
    Fh = lambda y, c, nargout: od.var_F(y, -b, k, h, mu, sigma, c, nargout, reg)
    Gh = lambda x, c, nargout: px.mean_zero_indicator(x, (m, n, k), c, nargout)
    xr, yr, phr, dhr = cp_ref(Fh, Gh, A, x0, y0, 1.0, tau, sigma, iters, 0.0)
    eng = E.Engine(E.MODEL_VAR_TV if reg == "tv" else E.MODEL_VAR_CURV, m, n, k=k, bc=bc, h=h, omega=omega, mu=mu, flags=E.FLAG_FP32)
    eng.set_images(np.concatenate([imgs[i].reshape(-1, order="F") for i in range(k)]))
    eng.set_primal(x0)
    eng.linearize()
    eng.set_dual(y0)
    est, npow = eng.norm_estimate()
    eng.solve(1.0, tau, sigma, iters, 0.0)
    x, y = eng.primal(), eng.dual()
    ph, dh = eng.history()
    eng.close()
    print("fp32 var", reg, bc, "x", _rel(x, xr), "y", _rel(y, yr), "p", abs(ph[-1, 0] - phr[-1, 0]) / abs(phr[-1, 0]))
    assert ph.shape == phr.shape
    assert _rel(x, xr) < TOL32 and _rel(y, yr) < TOL32
    assert abs(ph[-1, 0] - phr[-1, 0]) <= TOL32 * abs(phr[-1, 0]) and abs(dh[-1, 0] - dhr[-1, 0]) <= TOL32 * abs(dhr[-1, 0])
    assert np.array_equal(x, x.astype(np.float32).astype(np.float64))          # what comes back is representable in fp32


def test_fp32_variance_multilevel_driver():
    """var_tv_registration_no_ref_ml in the fp32 storage mode (x2 prolongation, power iteration) against the fp64 mode"""
    from ma_amd import synth
    from ma_amd.drivers import _var_ml
    m, n, k = 96, 80, 4
    imgs = synth.dynamic_test_image(m, n, k).numpy()
    optPara = dict(theta=1.0, maxIter=60, tol=1e-3, outerIter=[3, 2], mu=0.1, bc="neumann")
    u64 = _var_ml([imgs[i] for i in range(k)], optPara, "tv")
    u32 = _var_ml([imgs[i] for i in range(k)], optPara, "tv", fp32=True)
    a, b = u64[-1][optPara["outerIter"][1] - 1], u32[-1][optPara["outerIter"][1] - 1]
    print("fp32 var driver", _rel(b, a))
    assert _rel(b, a) < TOL32
