"""End-to-end drivers on the GPU vs the oracle (multi-level, outer loop, nu schedule, power iteration)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.mark.parametrize("reg,bc,mu,k", [("tv", "neumann", 0.2, 4), ("curvature", "dirichlet", 20.0, 4),
                                           ("tv", "neumann", 0.2, 18), ("curvature", "dirichlet", 20.0, 21)])   # K > 16: the general path
def test_mf_nn_ml_driver(reg, bc, mu, k):
    from ma_amd import drivers as D, synth
    from oracle import drivers as od
    m, n = 70, 66                 # two levels (35x33 -> 70x66), odd coarse size
    imgs = synth.dynamic_test_image(m, n, k).numpy()
    opt = dict(theta=1.0, maxIter=40, tol=1e-3, outerIter=[3, 2], mu=mu, nu_factor=[0.9, 0.9], bc=bc)
    keep_r, keep_g = {}, {}
    ur, Lr, SVr = od.mf_nn_registration_no_ref_ml([imgs[i] for i in range(k)], opt, reg, keep=keep_r)
    ug, Lg, SVg = D._mf_nn_ml([imgs[i] for i in range(k)], opt, reg, keep=keep_g)
    assert set(keep_r) == set(keep_g)
    for key in sorted(keep_r):
        a, b = keep_g[key], keep_r[key]
        print(key, "nu", a["nu"], b["nu"], "norm", a["norm_est"], b["norm_est"], a["n_power"], b["n_power"],
              "rows", a["primal_history"].shape, b["primal_history"].shape)
        assert abs(a["nu"] - b["nu"]) <= 1e-11 * abs(b["nu"])
        assert abs(a["norm_est"] - b["norm_est"]) <= 1e-11 * b["norm_est"]
        assert a["n_power"] == b["n_power"]
        assert a["primal_history"].shape == b["primal_history"].shape
        np.testing.assert_allclose(a["primal_history"][:, :5], b["primal_history"][:, :5], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(a["dual_history"][:, :5], b["dual_history"][:, :5], rtol=1e-8, atol=1e-9)
    for lev in range(len(ur)):
        for o in range(len(ur[lev])):
            if ur[lev][o] is None:
                continue
            assert _rel(ug[lev][o], ur[lev][o]) < 1e-9, (lev, o)
            assert _rel(Lg[lev][o], Lr[lev][o]) < 1e-9, (lev, o)
        np.testing.assert_allclose(SVg[lev], SVr[lev], rtol=1e-9, atol=1e-10)
