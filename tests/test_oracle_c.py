"""The C + OpenMP restatement (oracle/cp_nntv.c) against the numpy restatement: two independently written forms of the
NN-TV path (matrix-free loops + Gram/Jacobi vs assembled sparse A + LAPACK SVD) must agree to rounding."""
import numpy as np
import pytest


@pytest.mark.parametrize("bc,m,n,k", [("neumann", 24, 20, 3), ("linear", 17, 26, 4), ("neumann", 32, 32, 16)])
def test_c_restatement_matches_numpy(bc, m, n, k):
    from ma_amd import synth
    from oracle import c_oracle, drivers as od
    from oracle.chambolle_pock import chambolle_pock as cp_ref
    imgs = synth.dynamic_test_image(m, n, k).numpy()
    u0 = synth.smooth_displacement(m, n, k, amp=1.0).numpy().reshape(k, 2, n * m)
    u0_ref = np.transpose(u0, (2, 1, 0))
    omega = [0.0, float(m), 0.0, float(n)]
    h = (1.0, 1.0)
    A, b, T_u, gs = od.mf_nn_problem([imgs[i] for i in range(k)], u0_ref, h, omega, "tv", bc)
    mu, nu, tau, sigma = 0.2, 1.5 * k, 0.3, 0.3
    kmn = k * m * n
    rng = np.random.default_rng(3)
    x0 = np.concatenate([u0.reshape(-1), 0.5 * T_u.reshape(-1, order="F")])
    y0 = 0.01 * rng.standard_normal(6 * kmn)
    Fh = lambda y, c, nargout: od.mf_nn_F(y, b, k, h, mu, nu, sigma, c, nargout, "tv")
    Gh = lambda x, c, nargout: od.mf_nn_G(x, (m, n, k), c, nargout)
    xr, yr, phr, dhr = cp_ref(Fh, Gh, A, x0, y0, 1.0, tau, sigma, 15, 1e-3)
    g = np.concatenate([gs[i].reshape(-1, order="F") for i in range(k)])       # per image (gx; gy)
    x, y, ph, dh = c_oracle.nntv_chambolle_pock(m, n, k, bc, h, mu, nu, 1.0, tau, sigma, 15, 1e-3, g, b, x0, y0)
    assert ph.shape == phr.shape
    np.testing.assert_allclose(ph[:, :5], phr[:, :5], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(dh[:, :5], dhr[:, :5], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(ph[:, 5:], phr[:, 5:], rtol=0, atol=1e-7)       # constraint columns: Gram vs LAPACK
    np.testing.assert_allclose(dh[:, 5:], dhr[:, 5:], rtol=0, atol=1e-9)
    rel = lambda a, c: np.linalg.norm(a - c) / np.linalg.norm(c)
    assert rel(x, xr) < 1e-10 and rel(y, yr) < 1e-10
    assert c_oracle.threads() >= 1



@pytest.mark.parametrize("bc,m,n,k,h", [("dirichlet", 24, 20, 3, (1.0, 1.0)), ("neumann", 17, 26, 4, (2.0, 0.5)), ("dirichlet", 32, 32, 16, (1.0, 1.0))])
def test_c_restatement_curvature_matches_numpy(bc, m, n, k, h):
    """NN-curvature model (Laplacian, SSD on y2; NNE/mf_nn_curvature_registration_no_ref_ml.m:182-202,230-332)"""
    from ma_amd import synth
    from oracle import c_oracle, drivers as od
    from oracle.chambolle_pock import chambolle_pock as cp_ref
    imgs = synth.dynamic_test_image(m, n, k).numpy()
    u0 = synth.smooth_displacement(m, n, k, amp=1.0).numpy().reshape(k, 2, n * m)
    u0_ref = np.transpose(u0, (2, 1, 0))
    omega = [0.0, m * h[0], 0.0, n * h[1]]
    A, b, T_u, gs = od.mf_nn_problem([imgs[i] for i in range(k)], u0_ref, h, omega, "curvature", bc)
    mu, nu, tau, sigma = 20.0, 1.5 * k, 0.1, 0.1
    kmn = k * m * n
    rng = np.random.default_rng(4)
    x0 = np.concatenate([u0.reshape(-1), 0.5 * T_u.reshape(-1, order="F")])
    y0 = 0.01 * rng.standard_normal(4 * kmn)
    Fh = lambda y, c, nargout: od.mf_nn_F(y, b, k, h, mu, nu, sigma, c, nargout, "curvature")
    Gh = lambda x, c, nargout: od.mf_nn_G(x, (m, n, k), c, nargout)
    xr, yr, phr, dhr = cp_ref(Fh, Gh, A, x0, y0, 1.0, tau, sigma, 15, 1e-3)
    g = np.concatenate([gs[i].reshape(-1, order="F") for i in range(k)])
    x, y, ph, dh = c_oracle.nncurv_chambolle_pock(m, n, k, bc, h, mu, nu, 1.0, tau, sigma, 15, 1e-3, g, b, x0, y0)
    assert ph.shape == phr.shape
    np.testing.assert_allclose(ph[:, :5], phr[:, :5], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(dh[:, :5], dhr[:, :5], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(ph[:, 5:], phr[:, 5:], rtol=0, atol=1e-7)
    np.testing.assert_allclose(dh[:, 5:], dhr[:, 5:], rtol=1e-9, atol=1e-9)
    rel = lambda a, c: np.linalg.norm(a - c) / np.linalg.norm(c)
    assert rel(x, xr) < 1e-10 and rel(y, yr) < 1e-10
