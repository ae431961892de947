"""BASELINE.json's configurations at their full sizes: the headline (NN-TV 4096^2 x K=16, fp64) and config 4 (NN-curvature
4096^2 x 16) against the C restatement on the host for ten iterations, plus size-independent properties of the iterates
(feasibility, energy identities, bitwise run-to-run determinism); config 2 (TV-L1, rect_1 -> rect_2 resampled to 2048^2
with the reference's own s_grid facility), config 3 (multi-level K = 10 at 1024^2: the two coarsest levels against the
numpy oracle, properties on the finest) and config 5 (256 TV-L2 problems of 512^2, both storage modes)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(m, n, k):
    import torch
    from ma_amd import engine as E, synth
    dev = torch.device("cuda", 0)
    imgs = synth.dynamic_test_image(m, n, k, device=dev)
    imgs_host = imgs.transpose(1, 2).contiguous().cpu().numpy().reshape(-1)
    u0 = synth.smooth_displacement(m, n, k, amp=2.0, device=dev).cpu().numpy().reshape(-1)
    del imgs
    torch.cuda.empty_cache()
    eng = E.Engine(E.MODEL_MF_NN_TV, m, n, k=k, bc="neumann", h=(1.0, 1.0), omega=[0.0, float(m), 0.0, float(n)], mu=0.2)
    eng.set_images(imgs_host)
    x0 = np.concatenate([u0, np.zeros(k * m * n)])
    eng.set_primal(x0)
    eng.linearize()
    return eng, x0


def test_full_size_matches_c_restatement():
    from oracle import c_oracle
    m = n = 4096
    k = 16
    eng, x0 = _setup(m, n, k)
    nu = 0.9 * float(np.sum(eng.singular_values(0)))
    eng.set_nu(nu)
    tau = sigma = 0.25
    iters = 10                                 # one start-up dual step + ten launches of the bench kernel (was 2)
    _, g, b = eng.get_linearization()          # inputs of the iteration (the warp itself is covered by the golden tests)
    assert eng.dominant_kernel() == "k_mf_fused_tv<double, false, true, 16, false, 1, 8, false>"   # the kernel bench.py times
    eng.solve(1.0, tau, sigma, iters, 1e-3)
    ph, dh = eng.history()
    x = eng.primal()
    y = eng.dual()
    assert eng.handover_fallbacks() == 0
    eng.close()
    y0 = np.zeros(6 * k * m * n)
    xr, yr, phr, dhr = c_oracle.nntv_chambolle_pock(m, n, k, "neumann", (1.0, 1.0), 0.2, nu, 1.0, tau, sigma, iters, 1e-3,
                                                    g, b, x0, y0)
    del x0, y0, g, b
    assert ph.shape == phr.shape
    scale = np.abs(phr[:, 0]).max()                                                # energies: north_star 1e-10 relative
    np.testing.assert_allclose(ph[:, :5], phr[:, :5], rtol=1e-10, atol=1e-10 * scale)
    np.testing.assert_allclose(dh[:, :5], dhr[:, :5], rtol=1e-10, atol=1e-10 * scale)
    np.testing.assert_allclose(ph[:, 5:], phr[:, 5:], rtol=0, atol=1e-7)
    np.testing.assert_allclose(dh[:, 5:], dhr[:, 5:], rtol=0, atol=1e-9)
    ex = np.linalg.norm(x - xr) / np.linalg.norm(xr)
    ey = np.linalg.norm(y - yr) / np.linalg.norm(yr)
    print(f"4096^2 x 16: rel err x {ex:.2e} y {ey:.2e}")
    assert ex < 1e-10 and ey < 1e-10


def test_full_size_properties():
    m = n = 4096
    k = 16
    hist = []
    for rep in range(2):
        eng, _ = _setup(m, n, k)
        eng.set_nu(30.0)
        eng.solve(1.0, 0.25, 0.25, 12, 0.0)
        ph, dh = eng.history()
        eng.close()
        hist.append((ph, dh))
    ph, dh = hist[0]
    assert np.array_equal(ph, hist[1][0]) and np.array_equal(dh, hist[1][1])       # fixed-order reductions
    assert np.all(np.isfinite(ph)) and np.all(np.isfinite(dh)) and ph.shape == (13, 7)
    assert np.all(ph[1:, 6] <= 1e-12)            # mean-zero constraint holds after every primal step
    assert np.all(dh[1:, 5] <= 1e-12)            # |y1| <= h1h2, s1(y2) <= mu h1h2 after every dual step
    np.testing.assert_allclose(ph[:, 0], ph[:, 1] + ph[:, 2] + ph[:, 3] + ph[:, 4], rtol=1e-15)


def test_c4_curvature_full_size_matches_c_restatement():
    """BASELINE config 4 (single-GPU part): mf_nn_curvature at 4096^2 x 16, dirichlet rows, mu = 20
    (NNE/demo_mf_nn_curvature...:72-79): ten iterations against oracle/cp_nntv.c, then properties over 10 iterations"""
    import torch
    from ma_amd import engine as E, synth
    from oracle import c_oracle
    m = n = 4096
    k = 16
    dev = torch.device("cuda", 0)
    imgs_host = synth.dynamic_test_image(m, n, k, device=dev).transpose(1, 2).contiguous().cpu().numpy().reshape(-1)
    u0 = synth.smooth_displacement(m, n, k, amp=2.0, device=dev).cpu().numpy().reshape(-1)
    torch.cuda.empty_cache()
    eng = E.Engine(E.MODEL_MF_NN_CURV, m, n, k=k, bc="dirichlet", h=(1.0, 1.0), omega=[0.0, float(m), 0.0, float(n)], mu=20.0)
    eng.set_images(imgs_host)
    del imgs_host
    x0 = np.concatenate([u0, np.zeros(k * m * n)])
    eng.set_primal(x0)
    eng.linearize()
    nu = 0.9 * float(np.sum(eng.singular_values(0)))
    eng.set_nu(nu)
    tau = sigma = 0.1
    _, g, b = eng.get_linearization()
    eng.solve(1.0, tau, sigma, 10, 1e-3)
    ph, dh = eng.history()
    x, y = eng.primal(), eng.dual()
    xr, yr, phr, dhr = c_oracle.nncurv_chambolle_pock(m, n, k, "dirichlet", (1.0, 1.0), 20.0, nu, 1.0, tau, sigma, 10, 1e-3,
                                                      g, b, x0, np.zeros(4 * k * m * n))
    del g, b
    assert ph.shape == phr.shape
    scale = np.abs(phr[:, 0]).max()
    np.testing.assert_allclose(ph[:, :5], phr[:, :5], rtol=1e-10, atol=1e-10 * scale)
    np.testing.assert_allclose(dh[:, :5], dhr[:, :5], rtol=1e-10, atol=1e-10 * max(scale, np.abs(dhr[:, 0]).max()))
    np.testing.assert_allclose(ph[:, 5:], phr[:, 5:], rtol=0, atol=1e-7)
    np.testing.assert_allclose(dh[:, 5:], dhr[:, 5:], rtol=1e-9, atol=1e-9)
    ex = np.linalg.norm(x - xr) / np.linalg.norm(xr)
    ey = np.linalg.norm(y - yr) / np.linalg.norm(yr)
    print(f"C4 curvature 4096^2 x 16: rel err x {ex:.2e} y {ey:.2e}")
    assert ex < 1e-10 and ey < 1e-10
    del x, y, xr, yr
    hist = []
    for rep in range(2):
        eng.set_primal(x0)
        eng.set_dual(None)
        eng.solve(1.0, tau, sigma, 10, 0.0)
        hist.append(eng.history())
    eng.close()
    ph, dh = hist[0]
    assert np.array_equal(ph, hist[1][0]) and np.array_equal(dh, hist[1][1])       # fixed-order reductions
    assert np.all(np.isfinite(ph)) and np.all(np.isfinite(dh)) and ph.shape == (11, 7)
    assert np.all(ph[1:, 6] <= 1e-12) and np.all(dh[1:, 5] <= 1e-12)               # mean-zero u, |y1| <= h1h2
    np.testing.assert_allclose(ph[:, 0], ph[:, 1] + ph[:, 2] + ph[:, 3] + ph[:, 4], rtol=1e-15)
    np.testing.assert_allclose(dh[:, 0], -(dh[:, 1] + dh[:, 2] + dh[:, 3] + dh[:, 4]), rtol=1e-15)


def _c2_images(size, golden_dir):
    """rect_1 -> rect_2 (Data/*.png, 70 x 55) resampled to size^2 with evaluate_displacement(img, [1, 1], 0, omega,
    [size, size]) (IRE/evaluate_displacement.m:27-31), the recipe of BASELINE config 2"""
    import os
    from ma_amd import engine as E
    from ma_amd.drivers import normalize
    png = np.load(os.path.join(golden_dir, "data_png.npz"), allow_pickle=False)
    out = []
    for name in ("rect_1", "rect_2"):
        a = normalize(png[name].astype(np.float64))
        up, _ = E.evaluate_displacement(a, (1.0, 1.0), None, [0.0, float(a.shape[0]), 0.0, float(a.shape[1])], (size, size),
                                        want_grad=False)
        out.append(up)
    return out


def test_c2_tv_l1_registration_2048(golden_dir):
    """BASELINE config 2: TV-L1 registration (TV_registration + SAD_registration), lambda = 0.7, tau = 5
    (PDO/test_TV_L1_registration.m:30-36). Parity with the numpy oracle on the same recipe at 256^2 (one outer step);
    at 2048^2: bitwise determinism, dual feasibility, energy identities, monotone data of the outer loop."""
    from ma_amd import drivers as D, engine as E
    from oracle import drivers as od, operators as oo
    lam, tau, iters = 0.7, 5.0, 30
    R, T = _c2_images(256, golden_dir)
    Rr, _ = oo.evaluate_displacement(od.normalize(np.load(golden_dir + "/data_png.npz")["rect_1"].astype(float)), (1.0, 1.0),
                                     np.zeros((256 * 256, 2)), [0.0, 70.0, 0.0, 55.0], (256, 256))
    np.testing.assert_allclose(R, Rr, rtol=0, atol=1e-15)
    ur, vr, hr, _ = od.tv_registration(R, T, lam, tau, iters, 1, "SAD", keep_histories=True)
    u, v, hg, _ = D.tv_registration(R, T, lam, tau, iters, 1, "SAD", keep_histories=True)
    np.testing.assert_allclose(hg[0][0], hr[0][0], rtol=1e-9, atol=1e-9)
    assert np.linalg.norm(u - ur) <= 1e-10 * np.linalg.norm(ur) and np.linalg.norm(v - vr) <= 1e-10 * np.linalg.norm(vr)
    R, T = _c2_images(2048, golden_dir)
    runs = []
    for rep in range(2):
        u, v, hist, info = D.tv_registration(R, T, lam, tau, iters, 3, "SAD", keep_histories=True)
        runs.append((u, v, hist))
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1])
    for (ph, dh), (ph2, dh2) in zip(runs[0][2], runs[1][2]):
        assert np.array_equal(ph, ph2) and np.array_equal(dh, dh2)
        assert ph.shape == (iters + 1, 5) and np.all(np.isfinite(ph)) and np.all(np.isfinite(dh))
        assert np.all(dh[1:, 3] <= 1e-12)                           # ||v_ij|| <= 1 after every dual step (TV_registration.m:60-63)
        np.testing.assert_allclose(ph[:, 0], ph[:, 1] + ph[:, 2], rtol=1e-14)
    vv = runs[0][1].reshape((2048 * 2048, 4), order="F")
    assert np.sqrt((vv ** 2).sum(axis=1)).max() <= 1.0 + 1e-12


def test_c3_multilevel_k10_1024():
    """BASELINE config 3: mf_nn_tv_registration_no_ref_ml, K = 10 synthetic 1024^2 templates, 6 levels. The two coarsest
    levels (32^2, 64^2: outer iterations, nu schedule, power iteration, histories) against the numpy oracle; on every
    level: feasibility and energy identities; the whole driver twice: bitwise determinism of the result."""
    from ma_amd import drivers as D, synth
    from oracle import drivers as od
    m = n = 1024
    k = 10
    imgs = [synth.dynamic_test_image(m, n, k).numpy()[i] for i in range(k)]
    opt = dict(theta=1.0, maxIter=40, tol=1e-3, outerIter=[3, 2], mu=0.2, nu_factor=[0.9, 0.9], bc="neumann")
    keep_r = {}
    ur, Lr, SVr = od.mf_nn_registration_no_ref_ml(imgs, opt, "tv", keep=keep_r, max_levels=2)
    res = []
    for rep in range(2):
        keep_g = {}
        u, L, SV = D._mf_nn_ml(imgs, opt, "tv", keep=keep_g, download_all=(rep == 0))
        res.append((u, L, SV, keep_g))
    u, L, SV, keep_g = res[0]
    assert len(u) == 6
    for key in sorted(keep_r):
        a, b = keep_g[key], keep_r[key]
        assert abs(a["nu"] - b["nu"]) <= 1e-11 * abs(b["nu"]) and abs(a["norm_est"] - b["norm_est"]) <= 1e-11 * b["norm_est"]
        assert a["n_power"] == b["n_power"] and a["primal_history"].shape == b["primal_history"].shape
        np.testing.assert_allclose(a["primal_history"][:, :5], b["primal_history"][:, :5], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(a["dual_history"][:, :5], b["dual_history"][:, :5], rtol=1e-8, atol=1e-9)
        lev, o = key
        assert np.linalg.norm(u[lev][o] - ur[lev][o]) <= 1e-9 * np.linalg.norm(ur[lev][o])
        assert np.linalg.norm(L[lev][o] - Lr[lev][o]) <= 1e-9 * np.linalg.norm(Lr[lev][o])
    for key, a in keep_g.items():
        ph, dh = a["primal_history"], a["dual_history"]
        assert np.all(np.isfinite(ph)) and np.all(np.isfinite(dh))
        assert np.all(ph[1:, 6] <= 1e-12) and np.all(dh[1:, 5] <= 1e-12)
        np.testing.assert_allclose(ph[:, 0], ph[:, 1] + ph[:, 2] + ph[:, 3] + ph[:, 4], rtol=1e-15)
    u2, L2, SV2, _ = res[1]
    assert np.array_equal(u2[5][1], u[5][1]) and np.array_equal(L2[5][1], L[5][1])
    assert all(np.array_equal(SV2[lev], SV[lev]) for lev in range(6))
    assert u[5][1].shape == (m * n, 2, k) and np.abs(u[5][1].mean(axis=(0, 2))).max() <= 1e-10


@pytest.mark.parametrize("fp32", [False, True])
def test_c5_batch_256_tv_l2_512(fp32, monkeypatch):
    """BASELINE config 5: 256 independent TV-L2 (SSD_registration) problems of 512^2 in one handle, both storage modes:
    per-problem results equal a single-problem handle bitwise (problems 0, 100, 255), histories are finite and feasible,
    two runs agree bitwise; fp32 storage stays within 1e-4 of fp64 (north_star)."""
    from ma_amd import engine as E, synth
    m = n = 512
    B = 256
    lam, tau = 50.0, 4.0
    sigma = (1 - 1e-4) / (8 * tau)
    fr = synth.dynamic_test_image(m, n, 2).numpy()
    st, shifts = 12345, []
    for _ in range(B):
        st = (1103515245 * st + 12345) % (1 << 31)
        shifts.append(st % 9 - 4)
    Ts = [np.roll(fr[1], shifts[p], axis=0).reshape(-1, order="F") for p in range(B)]
    Rv = fr[0].reshape(-1, order="F")
    flags = E.FLAG_FP32 if fp32 else 0

    def run_batch():
        eng = E.Engine(E.MODEL_TV_L2_REG, m, n, lam=lam, batch=B, flags=flags)
        eng.set_images(np.concatenate(Ts), np.tile(Rv, B))
        for _ in range(2):
            eng.linearize()
            eng.solve(1.0, tau, sigma, 20, 1e-3)
        x = eng.primal().reshape(B, -1)
        ph, dh = eng.history()
        eng.close()
        return x, ph, dh

    x, ph, dh = run_batch()
    x2, ph2, dh2 = run_batch()
    assert np.array_equal(x, x2) and all(np.array_equal(a, b) for a, b in zip(ph, ph2))
    for p in range(B):
        assert np.all(np.isfinite(ph[p])) and np.all(np.isfinite(dh[p])) and np.all(dh[p][1:, 3] <= 1e-6 if fp32 else dh[p][1:, 3] <= 1e-12)
    ref64 = {}
    # a single 512^2 problem would run on the chunk kernel (too few waves for the marching one): give it the batch's
    # kernel and strip width, so that the per-problem reduction order -- and with it every bit -- is the batch's
    for p in (0, 100, 255):
        for fl in ((flags,) if not fp32 else (flags, 0)):
            monkeypatch.setenv("CP_TUNE_PW_STRIP", "32" if (fl & E.FLAG_FP32) else "6")   # cp_create's choice for the batch (k_pw_march2)
            e1 = E.Engine(E.MODEL_TV_L2_REG, m, n, lam=lam, flags=fl)
            e1.set_images(Ts[p], Rv)
            for _ in range(2):
                e1.linearize()
                e1.solve(1.0, tau, sigma, 20, 1e-3)
            x1 = e1.primal()
            h1 = e1.history()
            e1.close()
            if fl == flags:
                assert np.array_equal(x1, x[p]) and np.array_equal(h1[0], ph[p]) and np.array_equal(h1[1], dh[p])
            else:
                ref64[p] = (x1, h1[0])
    if fp32:
        for p, (x64, ph64) in ref64.items():
            assert np.linalg.norm(x[p] - x64) <= 1e-4 * np.linalg.norm(x64)
            assert abs(ph[p][-1, 0] - ph64[-1, 0]) <= 1e-4 * abs(ph64[-1, 0])
