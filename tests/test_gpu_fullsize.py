"""The headline configuration itself (NN-TV 4096^2 x K=16, fp64) against the C restatement on the host, plus
size-independent properties of the iterates (feasibility, bitwise run-to-run determinism)."""
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
    iters = 2
    _, g, b = eng.get_linearization()          # inputs of the iteration (the warp itself is covered by the golden tests)
    eng.solve(1.0, tau, sigma, iters, 1e-3)
    ph, dh = eng.history()
    x = eng.primal()
    y = eng.dual()
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
