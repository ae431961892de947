"""HIP engine through the C ABI vs the committed fixtures of tests/golden/ (no oracle code runs here: the fixtures are
the oracle's outputs for inputs that include the data files the reference ships, see tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _rel(a, b):
    return np.linalg.norm(np.ravel(a) - np.ravel(b)) / max(np.linalg.norm(np.ravel(b)), 1e-300)


def _vecF(a):
    return np.asarray(a, dtype=np.float64).reshape(-1, order="F")


def test_warp_fixture(golden_dir):
    """IRE/evaluate_displacement.m: warped image and gradient on Data/rect_2.png, displacements far outside the domain
    and exactly on cell boundaries; coarse-level call with h_grid = image spacing"""
    from ma_amd import engine as E
    f = _load(golden_dir, "warp.npz")
    T = f["T"]
    m, n = T.shape
    eng = E.Engine(E.MODEL_TV_L1_REG, m, n, lam=1.0)
    eng.set_images(_vecF(T), _vecF(T))
    eng.linearize(_vecF(f["u"]))
    Tg, gg, _ = eng.get_linearization()
    eng.close()
    np.testing.assert_allclose(Tg, _vecF(f["Tu"]), rtol=0, atol=1e-15)
    np.testing.assert_allclose(gg, _vecF(f["g"]), rtol=0, atol=1e-14)
    assert np.array_equal(Tg == 0.0, _vecF(f["Tu"]) == 0.0)
    Tc = f["Tc"]
    mc, nc = Tc.shape
    eng = E.Engine(E.MODEL_TV_L1_REG, mc, nc, h=tuple(f["hg"]), omega=[0.0, float(m), 0.0, float(n)], lam=1.0)
    eng.set_images(_vecF(Tc), _vecF(Tc))
    eng.linearize(_vecF(f["uc"]))
    Tg, gg, _ = eng.get_linearization()
    eng.close()
    np.testing.assert_allclose(Tg, _vecF(f["Tcu"]), rtol=0, atol=1e-15)
    np.testing.assert_allclose(gg, _vecF(f["gc"]), rtol=0, atol=1e-14)


@pytest.mark.parametrize("m,n", [(9, 6), (8, 8), (7, 11)])
def test_pyramid_fixture(golden_dir, m, n):
    """conv2 + subsampling of the image pyramid and kron(x, ones(2)) prolongation (NNE/mf_nn_tv...:75-76, 113-119),
    odd and even sizes"""
    from ma_amd import engine as E
    f = _load(golden_dir, "pyramid.npz")
    a, d = f[f"img_{m}x{n}"], f[f"down_{m}x{n}"]
    mc, nc = d.shape
    fine = E.Engine(E.MODEL_TV_L1_REG, m, n, lam=1.0)
    coarse = E.Engine(E.MODEL_TV_L1_REG, mc, nc, lam=1.0)
    fine.set_images(_vecF(a), _vecF(a))
    fine.downsample_to(coarse)
    np.testing.assert_allclose(coarse.images()[:mc * nc], _vecF(d), rtol=0, atol=1e-15)
    coarse.set_primal(f[f"xlow_{m}x{n}"])
    fine.prolong_from(coarse, 1.0)
    assert np.array_equal(fine.primal(), f[f"prol_{m}x{n}"])
    fine.close()
    coarse.close()


@pytest.mark.parametrize("tag,a,b,term", [("tvl1_rect", "rect_1", "rect_2", "SAD"),
                                          ("tvl2_rir", "rect_in_rect_1", "rect_in_rect_2", "SSD"),
                                          ("tvl1_slide", "sliding_rect_1", "sliding_rect_2", "SAD")])
def test_pairwise_steps_fixture(golden_dir, tag, a, b, term):
    """PDO/test_TV_L1_registration.m / test_TV_L2_registration.m on the reference's own Data/*.png: outer steps 0, 2
    and 20, each started from the stored state (linearise, 25-30 CP iterations, histories, stop rule)"""
    from ma_amd import engine as E
    from ma_amd.drivers import normalize
    f = _load(golden_dir, "pairwise_e2e.npz")
    png = _load(golden_dir, "data_png.npz")
    R = normalize(png[a].astype(np.float64))
    T = normalize(png[b].astype(np.float64))
    m, n = R.shape
    lam, tau, iters, _ = f[f"{tag}_par"]
    sigma = (1 - 1e-4) / (8 * tau)
    eng = E.Engine(E.MODEL_TV_L1_REG if term == "SAD" else E.MODEL_TV_L2_REG, m, n, lam=float(lam))
    eng.set_images(_vecF(T), _vecF(R))
    for s in (0, 2, 20):
        eng.set_primal(f[f"{tag}_s{s}_u0"])
        eng.set_dual(f[f"{tag}_s{s}_v0"])
        eng.linearize()
        eng.solve(1.0, float(tau), sigma, int(iters), 1e-3)
        ph, dh = eng.history()
        phr, dhr = f[f"{tag}_s{s}_ph"], f[f"{tag}_s{s}_dh"]
        assert ph.shape == phr.shape, (s, ph.shape, phr.shape)
        np.testing.assert_allclose(ph, phr, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(dh, dhr, rtol=1e-9, atol=1e-6 * max(1.0, np.abs(dhr).max()))
        assert _rel(eng.primal(), f[f"{tag}_s{s}_u"]) < TOL, s
        assert _rel(eng.dual(), f[f"{tag}_s{s}_v"]) < TOL, s
    eng.close()


def test_c1_denoising_fixture(golden_dir):
    """PDO/test_TV_L1_denoising.m on Data/rect1.png with the committed salt & pepper mask, 1000 iterations"""
    from ma_amd import drivers as D
    f = _load(golden_dir, "pairwise_e2e.npz")
    x, y, ph, dh, _ = D.tv_l1_denoising(f["c1_noisy"], lam=1.5, tau=0.02, maxIter=1000, tol=1e-5)
    assert ph.shape == f["c1_ph"].shape
    np.testing.assert_allclose(ph, f["c1_ph"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(dh, f["c1_dh"], rtol=1e-9, atol=1e-9)
    assert _rel(x, f["c1_x"]) < TOL and _rel(y, f["c1_y"]) < TOL


@pytest.mark.parametrize("reg,bc,mu", [("tv", "neumann", 0.2), ("curvature", "dirichlet", 20.0)])
def test_mfnn_driver_fixture(golden_dir, reg, bc, mu):
    """NNE/mf_nn_{tv,curvature}_registration_no_ref_ml.m end to end (2 levels, outer iterations, nu schedule, power
    iteration, histories) against the stored run"""
    from ma_amd import drivers as D
    f = _load(golden_dir, "mfnn_e2e.npz")
    imgs = f["imgs"]
    k = imgs.shape[0]
    opt = dict(theta=1.0, maxIter=60, tol=1e-3, outerIter=[3, 2], mu=mu, nu_factor=[0.9, 0.9], bc=bc)
    keep = {}
    u, L, SV = D._mf_nn_ml([imgs[i] for i in range(k)], opt, reg, keep=keep)
    levels = int(f[f"{reg}_levels"][0])
    assert len(u) == levels
    for lev in range(levels):
        np.testing.assert_allclose(SV[lev], f[f"{reg}_SV{lev}"], rtol=1e-9, atol=1e-10)
        for o in range(len(u[lev])):
            key = f"{reg}_u_{lev}_{o}"
            if key not in f:
                assert u[lev][o] is None
                continue
            nu, tau, norm_est, n_power, rows = f[f"{reg}_sched_{lev}_{o}"]
            kk = keep[(lev, o)]
            assert abs(kk["nu"] - nu) <= 1e-11 * abs(nu) and abs(kk["norm_est"] - norm_est) <= 1e-11 * norm_est
            assert kk["n_power"] == int(n_power) and kk["primal_history"].shape[0] == int(rows)
            np.testing.assert_allclose(kk["primal_history"][:, :5], f[f"{reg}_ph_{lev}_{o}"][:, :5], rtol=1e-8, atol=1e-9)
            np.testing.assert_allclose(kk["dual_history"][:, :5], f[f"{reg}_dh_{lev}_{o}"][:, :5], rtol=1e-8, atol=1e-9)
            assert _rel(u[lev][o], f[key]) < 1e-9, (lev, o)
            assert _rel(L[lev][o], f[f"{reg}_L_{lev}_{o}"]) < 1e-9, (lev, o)


@pytest.mark.parametrize("reg,bc,mu", [("tv", "neumann", 0.1), ("curvature", "neumann", 5.0)])
def test_var_driver_fixture(golden_dir, reg, bc, mu):
    """NNE/var_{tv,curvature}_registration_no_ref_ml.m end to end (2 levels, x2 prolongation) against the stored run"""
    from ma_amd import drivers as D
    f = _load(golden_dir, "var_e2e.npz")
    imgs = _load(golden_dir, "mfnn_e2e.npz")["imgs"]
    k = imgs.shape[0]
    opt = dict(theta=1.0, maxIter=60, tol=1e-3, outerIter=[3, 2], mu=mu, bc=bc)
    keep = {}
    u = D._var_ml([imgs[i] for i in range(k)], opt, reg, keep=keep)
    assert len(u) == int(f[f"{reg}_levels"][0])
    for lev in range(len(u)):
        for o in range(len(u[lev])):
            key = f"{reg}_u_{lev}_{o}"
            if key not in f:
                assert u[lev][o] is None
                continue
            tau, norm_est, n_power, rows = f[f"{reg}_sched_{lev}_{o}"]
            kk = keep[(lev, o)]
            assert abs(kk["norm_est"] - norm_est) <= 1e-11 * norm_est and kk["n_power"] == int(n_power)
            assert kk["primal_history"].shape[0] == int(rows)
            np.testing.assert_allclose(kk["primal_history"], f[f"{reg}_ph_{lev}_{o}"], rtol=1e-8, atol=1e-9)
            np.testing.assert_allclose(kk["dual_history"], f[f"{reg}_dh_{lev}_{o}"], rtol=1e-8, atol=1e-9)
            assert _rel(u[lev][o], f[key]) < 1e-9, (lev, o)


def test_single_level_drivers_fixture(golden_dir):
    """NNE/mf_nn_tv_registration_no_ref.m (F1 = SAD with weight 1, F2 = mu*||.||_{2,1} per image, :206-216,230-241 --
    CP_FLAG_MFNN_NORM21) and NNE/var_tv_registration_no_ref.m (swapped SSD prox arguments, :173) against the stored runs:
    3 outer iterations, linear boundary rows, nu schedule, power iteration, histories, stop rule"""
    from ma_amd import drivers as D
    f = _load(golden_dir, "single_level_e2e.npz")
    imgs = f["imgs"]
    k = imgs.shape[0]
    opt = dict(theta=1.0, maxIter=50, tol=1e-3, outerIter=3, mu=0.15, nu_factor=0.9, bc="linear")
    keep = {}
    u, L, SV = D.mf_nn_tv_registration_no_ref([imgs[i] for i in range(k)], opt, keep=keep)
    assert SV.shape == f["mfnn_SV"].shape == (k, 3)
    np.testing.assert_allclose(SV, f["mfnn_SV"], rtol=1e-9, atol=1e-10)
    for o in range(3):
        nu, tau, norm_est, n_power, rows = f[f"mfnn_sched_{o}"]
        kk = keep[(0, o)]
        assert abs(kk["nu"] - nu) <= 1e-11 * abs(nu) and abs(kk["norm_est"] - norm_est) <= 1e-11 * norm_est
        assert kk["n_power"] == int(n_power) and kk["primal_history"].shape[0] == int(rows)
        np.testing.assert_allclose(kk["primal_history"][:, :5], f[f"mfnn_ph_{o}"][:, :5], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(kk["dual_history"], f[f"mfnn_dh_{o}"], rtol=1e-8, atol=1e-8)
        assert _rel(u[o], f[f"mfnn_u_{o}"]) < 1e-9 and _rel(L[o], f[f"mfnn_L_{o}"]) < 1e-9, o
    # the functional matters: the _ml arithmetic (pointwise 2x2 nuclear norm) ends somewhere else
    assert _rel(f["mfnn_ml_u_2"], f["mfnn_u_2"]) > 1e-3
    optv = dict(theta=1.0, maxIter=50, tol=1e-3, outerIter=3, mu=0.1, bc="neumann")
    keep = {}
    uv = D.var_tv_registration_no_ref([imgs[i] for i in range(k)], optv, keep=keep)
    for o in range(3):
        tau, norm_est, n_power, rows = f[f"var_sched_{o}"]
        kk = keep[(0, o)]
        assert abs(kk["norm_est"] - norm_est) <= 1e-11 * norm_est and kk["n_power"] == int(n_power)
        assert kk["primal_history"].shape[0] == int(rows)
        np.testing.assert_allclose(kk["primal_history"], f[f"var_ph_{o}"], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(kk["dual_history"], f[f"var_dh_{o}"], rtol=1e-8, atol=1e-9)
        assert _rel(uv[o], f[f"var_u_{o}"]) < 1e-9, o


@pytest.mark.parametrize("tag,a,b,term", [("tvl1_rect", "rect_1", "rect_2", "SAD"),
                                          ("tvl2_rir", "rect_in_rect_1", "rect_in_rect_2", "SSD"),
                                          ("tvl1_slide", "sliding_rect_1", "sliding_rect_2", "SAD")])
def test_pairwise_free_running_script_fixture(golden_dir, tag, a, b, term):
    """The whole script of PDO/test_TV_L1_registration.m:94-122 / test_TV_L2_registration.m (40 outer steps, warm starts,
    the reference's parameter sets :28-54) free-running on the device, against the stored energies of every outer step.
    The script is ill-conditioned across outer steps: `{tag}_sens` holds what ONE ulp on the template does to the oracle
    itself (tests/golden/make_golden.py) -- nothing for a number of steps, then a jump to 1e-4 .. 5e-3 when the first pixel
    re-selects its interpolation cell. Bar: 1e-10 on every step before that onset; afterwards the engine may be as far
    from the oracle as the perturbed oracle is (2 x the largest distance seen), never further."""
    from ma_amd import engine as E
    from ma_amd.drivers import normalize
    f = _load(golden_dir, "pairwise_e2e.npz")
    png = _load(golden_dir, "data_png.npz")
    R = normalize(png[a].astype(np.float64))
    T = normalize(png[b].astype(np.float64))
    m, n = R.shape
    lam, tau, iters, steps = f[f"{tag}_par"]
    sigma = (1 - 1e-4) / (8 * tau)
    en, sens = f[f"{tag}_energies"], f[f"{tag}_sens"]
    onset = int(np.argmax(sens[:, 0] > 1e-9)) if (sens[:, 0] > 1e-9).any() else int(steps)
    assert 1 <= onset                                        # the first outer step is always well conditioned
    loose_u, loose_e = 2.0 * sens[:, 0].max(), 2.0 * sens[:, 1].max()
    eng = E.Engine(E.MODEL_TV_L1_REG if term == "SAD" else E.MODEL_TV_L2_REG, m, n, lam=float(lam))
    eng.set_images(_vecF(T), _vecF(R))
    worst_pre, worst_post = 0.0, 0.0
    for s in range(int(steps)):
        eng.linearize()                                      # u0 <- u_star (:119-120)
        eng.solve(1.0, float(tau), sigma, int(iters), 1e-3)
        ph, _ = eng.history()
        assert ph.shape[0] == int(en[s, 2]), s               # iteration counts of every outer step
        de = abs(ph[-1, 0] - en[s, 0]) / abs(en[s, 0])
        du = None
        if f"{tag}_s{s}_u" in f:
            du = _rel(eng.primal(), f[f"{tag}_s{s}_u"])
        elif s == int(steps) - 1:
            du = _rel(eng.primal(), f[f"{tag}_u_final"])
        if s < onset - 1:
            assert de <= TOL and (du is None or du <= TOL), (s, de, du)
            worst_pre = max(worst_pre, de, du or 0.0)
        else:
            assert de <= max(TOL, loose_e) and (du is None or du <= max(TOL, loose_u)), (s, de, du, loose_e, loose_u)
            worst_post = max(worst_post, de, du or 0.0)
    eng.close()
    print(f"{tag}: onset of the oracle's own 1-ulp divergence at outer step {onset}; engine-oracle gap before {worst_pre:.1e}, "
          f"after {worst_post:.1e} (bound {max(loose_e, loose_u):.1e})")


def test_evaluate_displacement_s_grid_fixture(golden_dir):
    """IRE/evaluate_displacement.m with an output resolution different from the image's (s_grid, :27-31): Data/rect_2.png
    (70 x 55) resampled to 96 x 80 with u = 0 -- the recipe of BASELINE config 2 -- against the stored oracle output, and a
    displaced coarse grid against the oracle directly"""
    from ma_amd import engine as E
    from oracle import operators as oo
    f = _load(golden_dir, "warp.npz")
    T = f["T"]
    m, n = T.shape
    Tup, _ = E.evaluate_displacement(T, (1.0, 1.0), None, [0.0, float(m), 0.0, float(n)], (96, 80))
    np.testing.assert_allclose(Tup, f["Tup"], rtol=0, atol=1e-15)
    assert np.array_equal(Tup == 0.0, f["Tup"] == 0.0)
    rng = np.random.default_rng(5)
    u = 4.0 * rng.standard_normal((33 * 41, 2))
    u[:20] -= 90.0
    Tr, gr = oo.evaluate_displacement(T, (2.0, 0.5), u, [1.0, 120.0, -3.0, 30.0], (33, 41))
    Tg, gg = E.evaluate_displacement(T, (2.0, 0.5), u, [1.0, 120.0, -3.0, 30.0], (33, 41))
    np.testing.assert_allclose(Tg, Tr, rtol=0, atol=5e-15)        # products contracted to fma on the device: a few ulp
    np.testing.assert_allclose(gg, gr, rtol=0, atol=5e-14)
    assert np.array_equal(Tg == 0.0, Tr == 0.0)                   # the index / validity logic is exact
    # default region and default grid: the plain warp of the image at its own resolution
    Tr, gr = oo.evaluate_displacement(T, (1.0, 1.0), f["u"])
    Tg, gg = E.evaluate_displacement(T, (1.0, 1.0), f["u"])
    np.testing.assert_allclose(Tg, Tr, rtol=0, atol=1e-15)
    np.testing.assert_allclose(gg, gr, rtol=0, atol=1e-14)
