"""Operator-level parity through the C ABI against the committed fixture tests/golden/operators.npz (SURVEY.md section
8c items (1) and (2)): the linear operators D, D', Laplacian, B via cp_apply_A / cp_apply_At, and the operator
evaluations the solver uses (value, conjugate value, violation, conjugate prox of SAD, pointwise_2x2_nn,
nuclear_norm_constraint, mean_zero_indicator, zero_function, SSD, norm21) via cp_set_linearization + one CP iteration
whose other terms vanish. Where the fixture's input cannot be produced by the block operator (e.g. a value at D*u), the
expected number comes from the oracle evaluated on the same input."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-12


def _rel(a, b):
    return np.linalg.norm(np.ravel(a) - np.ravel(b)) / max(np.linalg.norm(np.ravel(b)), 1e-300)


@pytest.fixture(scope="module")
def fx(golden_dir):
    import os
    return dict(np.load(os.path.join(golden_dir, "operators.npz")))


@pytest.mark.parametrize("bc", ["neumann", "linear"])
def test_forward_difference_and_adjoint(fx, bc):
    """NNE/finite_difference_operator.m:29-50; h = (0.5, 2) are powers of two, so 1/h is exact: bit-exact."""
    from ma_amd import engine as E
    m, n, k = (int(v) for v in fx["shape"])
    kmn = k * m * n
    eng = E.Engine(E.MODEL_MF_NN_TV, m, n, k=k, bc=bc, h=tuple(fx["h"]), mu=1.0)
    y = eng.apply_A(np.concatenate([fx["x_u"], fx["L"]]))
    assert np.array_equal(y[:kmn], fx["L"])                        # grad T = 0: y1 = L
    assert np.array_equal(y[kmn:5 * kmn], fx[f"D_{bc}_x"])         # y2 = D u, order [Dx ux; Dy ux; Dx uy; Dy uy] per image
    assert _rel(y[5 * kmn:], fx["B_L"]) < 1e-15                    # y3 = B L   (NNE/mean_free_operator.m:20)
    x = eng.apply_At(np.concatenate([np.zeros(kmn), fx["y_d"], np.zeros(kmn)]))
    np.testing.assert_allclose(x[:2 * kmn], fx[f"Dt_{bc}_y"], rtol=0, atol=4e-16 * np.abs(fx[f"Dt_{bc}_y"]).max())
    assert not x[2 * kmn:].any()
    # the variance model's block operator shares D:  y = [B(grad T . u); D u]
    ev = E.Engine(E.MODEL_VAR_TV, m, n, k=k, bc=bc, h=tuple(fx["h"]), mu=1.0)
    yv = ev.apply_A(fx["x_u"])
    assert not yv[:kmn].any() and np.array_equal(yv[kmn:], fx[f"D_{bc}_x"])
    ev.close()
    eng.close()


@pytest.mark.parametrize("bc", ["neumann", "dirichlet"])
def test_laplacian_and_adjoint(fx, bc):
    """NNE/discrete_laplacian.m:23-46 (symmetric: adjoint = itself)"""
    from ma_amd import engine as E
    from oracle import operators as oo
    m, n, k = (int(v) for v in fx["shape"])
    kmn = k * m * n
    eng = E.Engine(E.MODEL_MF_NN_CURV, m, n, k=k, bc=bc, h=tuple(fx["h"]), mu=1.0)
    y = eng.apply_A(np.concatenate([fx["x_u"], fx["L"]]))
    assert _rel(y[kmn:3 * kmn], fx[f"Lap_{bc}_x"]) < 1e-15
    assert _rel(y[3 * kmn:], fx["B_L"]) < 1e-15
    x = eng.apply_At(np.concatenate([np.zeros(kmn), fx["y_l"], np.zeros(kmn)]))
    assert _rel(x[:2 * kmn], oo.discrete_laplacian(m, n, tuple(fx["h"]), k, bc).T @ fx["y_l"]) < 1e-15
    # the state of the handle is untouched by the products
    assert not eng.primal().any() and not eng.dual().any()
    eng.close()


def test_block_operator_with_gradient(fx):
    """A and A' with a gradient field: NNE/mf_nn_tv...:183-188 assembled by the oracle."""
    from ma_amd import engine as E
    from oracle import operators as oo
    import scipy.sparse as sp
    m, n, k = (int(v) for v in fx["shape"])
    mn, kmn = m * n, k * m * n
    rng = np.random.default_rng(3)
    g = rng.standard_normal(2 * kmn)
    b = rng.standard_normal(kmn)
    eng = E.Engine(E.MODEL_MF_NN_TV, m, n, k=k, bc="linear", h=tuple(fx["h"]), mu=1.0)
    eng.set_linearization(None, g, b)
    Tg, gg, bg = eng.get_linearization()
    assert np.array_equal(gg, g) and np.array_equal(bg, b)
    G = sp.block_diag([oo.grad_matrix(g[2 * i * mn: 2 * (i + 1) * mn].reshape(mn, 2, order="F")) for i in range(k)])
    D = oo.finite_difference_operator(m, n, tuple(fx["h"]), k, "linear")
    B = oo.mean_free_operator(m, n, k)
    A = sp.bmat([[-G, sp.identity(kmn)], [D, None], [None, B]]).tocsr()
    x = np.concatenate([fx["x_u"], fx["L"]])
    y = rng.standard_normal(6 * kmn)
    assert _rel(eng.apply_A(x), A @ x) < 1e-15
    assert _rel(eng.apply_At(y), A.T @ y) < 1e-15
    eng.close()


def test_sad_and_pointwise_nn_evaluations(fx):
    """One image (k = 1), h1*h2 = mu of the fixture: NNE/SAD.m:42-83 and NNE/pointwise_2x2_nn.m:12-57 evaluated by the
    engine on the fixture's inputs (value at A*x0, conjugate value at y0, conjugate prox in the first dual step)."""
    from ma_amd import engine as E
    from oracle import prox as px, operators as oo
    m, n, _ = (int(v) for v in fx["shape"])
    mn = m * n
    mu, sig = float(fx["mu"]), float(fx["sig"])
    h = (mu, 1.0)                                         # prod(h) = mu: the weight of SAD and of the 2x2 nuclear norm
    eng = E.Engine(E.MODEL_MF_NN_TV, m, n, k=1, bc="neumann", h=h, mu=1.0)
    eng.set_linearization(None, np.zeros(2 * mn), fx["I"])      # grad T = 0, b = I
    eng.set_nu(1.0)
    # dual side: y0 = [Lv; v4; 0], x0 = 0  ->  A*xbar = 0, so y_1 = prox_{sigma F*}(y0) blockwise
    eng.set_primal(np.zeros(3 * mn))
    eng.set_dual(np.concatenate([fx["Lv"], fx["v4"], np.zeros(mn)]))
    eng.solve(1.0, 0.4, sig, 1, 1e-3)
    y = eng.dual()
    ph, dh = eng.history()
    assert _rel(y[:mn], fx["SAD_cprox"]) < TOL
    assert _rel(y[mn:5 * mn], fx["nn2_cprox"]) < TOL
    assert not y[5 * mn:].any()
    assert abs(dh[0, 1] - fx["SAD_cval"][0]) <= TOL * abs(fx["SAD_cval"][0])          # F1*(y0)
    assert abs(dh[0, 2] - fx["nn2_cval"][0]) <= TOL * max(abs(fx["nn2_cval"][0]), 1.0)  # F2*(y0)
    assert abs(dh[0, 5] - max(fx["SAD_cval"][1], fx["nn2_cval"][1], 0.0)) <= 1e-12    # max of the violations
    # primal side: x0 = [u; Lv] -> (A x0)_1 = Lv, (A x0)_2 = D u
    u = np.random.default_rng(11).standard_normal(2 * mn)
    eng.set_primal(np.concatenate([u, fx["Lv"]]))
    eng.set_dual(np.zeros(6 * mn))
    eng.solve(1.0, 0.4, sig, 0, 1e-3)
    ph, dh = eng.history()
    assert abs(ph[0, 1] - fx["SAD_val"][0]) <= TOL * abs(fx["SAD_val"][0])            # F1(A x0) = mu * sum |Lv - I|
    Du = oo.finite_difference_operator(m, n, h, 1, "neumann") @ u
    want = px.pointwise_2x2_nn(Du, mu, sig, False, 2)[0]
    assert abs(ph[0, 2] - want) <= TOL * abs(want)                                    # F2(A x0) = mu * sum (s1 + s2)
    # G(x0): mean_zero_indicator violation (NNE/mean_zero_indicator.m:62-65)
    assert abs(ph[0, 6] - px.mean_zero_indicator(u, (m, n, 1), False, 2)[1]) <= 1e-14
    eng.close()


def test_nuclear_norm_constraint_evaluations(fx):
    """NNE/nuclear_norm_constraint.m:34-94 with NNE/l1ball_projection.m:23-68: conjugate prox (first dual step from
    y3 = Y, x0 = 0), conjugate value nu * s_max, and the violation (sum s - nu)/nu at B*L."""
    from ma_amd import engine as E
    from oracle import prox as px, operators as oo
    m, n, k = (int(v) for v in fx["shape"])
    mn, kmn = m * n, k * m * n
    nu, tau = float(fx["nnc_nu"]), float(fx["nnc_tau"])
    eng = E.Engine(E.MODEL_MF_NN_TV, m, n, k=k, bc="neumann", h=(1.0, 1.0), mu=1.0)
    eng.set_linearization(None, np.zeros(2 * kmn), np.zeros(kmn))
    eng.set_nu(nu)
    eng.set_primal(np.zeros(3 * kmn))
    eng.set_dual(np.concatenate([np.zeros(5 * kmn), fx["nnc_Y"]]))
    eng.solve(1.0, 0.3, tau, 1, 1e-3)                     # the conjugate prox runs with the dual step (sigma = tau of the fixture)
    y = eng.dual()
    ph, dh = eng.history()
    assert _rel(y[5 * kmn:], fx["nnc_cprox"]) < 1e-11
    assert not y[:5 * kmn].any()
    assert abs(dh[0, 3] - fx["nnc_cval"][0]) <= 1e-12 * fx["nnc_cval"][0]             # F3*(y0) = nu * s_max
    # violation at B*L for L = Y
    eng.set_primal(np.concatenate([np.zeros(2 * kmn), fx["nnc_Y"]]))
    eng.set_dual(np.zeros(6 * kmn))
    eng.solve(1.0, 0.3, tau, 0, 1e-3)
    ph, dh = eng.history()
    BL = oo.mean_free_operator(m, n, k) @ fx["nnc_Y"]
    want = px.nuclear_norm_constraint(BL, k, tau, nu, False, 2)[1]
    # F_con = max over the three components; SAD and the 2x2 norm have none
    assert abs(ph[0, 5] - want) <= 1e-10 * max(want, 1.0)
    # G*(-A'y0) at y0 = [L; 0; 0]: zero_function violation max |L| (NNE/zero_function.m:32)
    eng.set_primal(np.zeros(3 * kmn))
    eng.set_dual(np.concatenate([fx["L"], np.zeros(5 * kmn)]))
    eng.solve(1.0, 0.3, tau, 0, 1e-3)
    ph, dh = eng.history()
    assert abs(dh[0, 6] - fx["zf_cval"][1]) <= 1e-15
    eng.close()


def test_mean_zero_indicator_prox(fx):
    """NNE/mean_zero_indicator.m:45-53: with y = 0, grad T = 0, L = 0 and fields constant per image the dual step
    stays zero and x_1 = prox_G(x_0) = x_0 - mean."""
    from ma_amd import engine as E
    from oracle import prox as px
    m, n, k = (int(v) for v in fx["shape"])
    mn, kmn = m * n, k * m * n
    c = np.random.default_rng(2).standard_normal(2 * k)
    u = np.repeat(c, mn)
    eng = E.Engine(E.MODEL_MF_NN_TV, m, n, k=k, bc="neumann", h=(1.0, 1.0), mu=1.0)
    eng.set_linearization(None, np.zeros(2 * kmn), np.zeros(kmn))
    eng.set_nu(1.0)
    eng.set_primal(np.concatenate([u, np.zeros(kmn)]))
    eng.solve(1.0, 0.3, 0.3, 1, 1e-3)
    x = eng.primal()
    assert _rel(x[:2 * kmn], px.mean_zero_indicator(u, (m, n, k), False, 3)[2]) < 1e-14
    assert not eng.dual().any()
    eng.close()


def test_variance_model_evaluations(fx):
    """NNE/SSD.m:43-67 and NNE/norm21.m:52-88 as the variance drivers use them: conjugate prox of the first dual step
    (x0 = 0), conjugate values and violations at y0."""
    from ma_amd import engine as E
    m, n, _ = (int(v) for v in fx["shape"])
    mn = m * n
    mu, sig = float(fx["mu"]), float(fx["sig"])
    # k = 1 would make B = 0; use k = 2 with the fixture vector in image 1 and zeros in image 2. The drivers pass -B*b
    # as the SSD argument (NNE/var_tv..._ml.m:172) and the device centres b itself: b = [-2I; 0] gives -B*b = [I; -I].
    # h1*h2 = mu of the fixture is the SSD weight (:221), mu_engine = 1 makes the norm21 weight mu*h1*h2 the same.
    from oracle import prox as px
    k = 2
    eng = E.Engine(E.MODEL_VAR_TV, m, n, k=k, bc="neumann", h=(mu, 1.0), mu=1.0)
    eng.set_linearization(None, np.zeros(2 * k * mn), np.concatenate([-2.0 * fx["I"], np.zeros(mn)]))
    eng.set_primal(np.zeros(2 * k * mn))
    y1 = np.concatenate([fx["Lv"], np.zeros(mn)])
    eng.set_dual(np.concatenate([y1, fx["v4"], np.zeros(4 * mn)]))
    eng.solve(1.0, 0.4, sig, 1, 1e-3)
    y = eng.dual()
    ph, dh = eng.history()
    assert _rel(y[:mn], fx["SSD_cprox"]) < TOL                                        # image 1: SSD(Lv, I, mu, sig, conj)
    gI = np.concatenate([fx["I"], -fx["I"]])
    assert _rel(y[:k * mn], px.SSD(y1, gI, mu, sig, True, 3)[2]) < TOL
    assert _rel(y[k * mn: k * mn + 4 * mn], fx["n21_cprox"]) < TOL                    # NNE/norm21.m conjugate prox
    want = px.SSD(y1, gI, mu, sig, True, 2)[0]
    assert abs(dh[0, 1] - want) <= TOL * abs(want)                                    # F1*(y0)
    assert abs(dh[0, 2] - fx["n21_cval"][0]) <= 1e-14                                 # F2*(y0)
    assert abs(dh[0, 4] - max(fx["n21_cval"][1], 0.0)) <= 1e-13                       # FS_con = max ||v|| - mu
    eng.close()
