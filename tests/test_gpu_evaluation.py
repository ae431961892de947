"""SURVEY.md section 8 f4 on the device: landmark_transform / landmark_accuracy against the numpy restatement
(oracle/evaluation.py; parity unpinned -- the reference stores no landmark vectors), and the demos' evaluate + save
sections end to end on a small driver run."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _field(m, n, amp, seed, noise=0.3):
    """smooth part + noise whose slope stays below 1 per cell, so that x -> x + (y - F(x)) is a contraction (a field that
    is not one makes the iteration chaotic and it amplifies the last bit of the interpolation weights)"""
    rng = np.random.default_rng(seed)
    ii, jj = np.meshgrid(np.arange(m) / m, np.arange(n) / n, indexing="ij")
    u = np.zeros((m, n, 2))
    u[:, :, 0] = amp * np.sin(2 * np.pi * ii) * np.cos(np.pi * jj) + noise * rng.standard_normal((m, n))
    u[:, :, 1] = amp * np.cos(np.pi * ii) * np.sin(2 * np.pi * jj) + noise * rng.standard_normal((m, n))
    return u


@pytest.mark.parametrize("m,n,omega", [(48, 40, None), (33, 57, [0.0, 16.5, 0.0, 28.5]), (64, 64, [-1.0, 1.0, -1.0, 1.0])])
def test_landmark_transform_matches_oracle(m, n, omega):
    from ma_amd import evaluation as ev
    from oracle import evaluation as oe
    if omega is None:
        omega = [0.0, float(m), 0.0, float(n)]
    sx, sy = omega[1] - omega[0], omega[3] - omega[2]
    u = _field(m, n, 0.04 * min(sx, sy), 3, noise=0.15 * min(sx / m, sy / n))
    rng = np.random.default_rng(7)
    # inside, near the border, outside the region (zero padding of the interpolation), and small coordinates for which the
    # literal termination expression sum(F(x) - y.^2) stays positive (the reference's warning path)
    LM = np.column_stack([omega[0] + sx * rng.uniform(-0.05, 1.05, 24), omega[2] + sy * rng.uniform(-0.05, 1.05, 24)])
    with warnings.catch_warnings(record=True) as wo:
        warnings.simplefilter("always")
        ref = oe.landmark_transform(LM, u, omega)
    with warnings.catch_warnings(record=True) as wd:
        warnings.simplefilter("always")
        out = ev.landmark_transform(LM, u, omega)
    assert out.shape == ref.shape
    assert len(wd) == len(wo)                                   # the same landmarks fall back to the initial guess
    assert np.max(np.abs(out - ref)) <= 1e-11 * max(abs(omega[1]), abs(omega[3]), 1.0)


def test_landmark_transform_inverts_a_constant_shift_and_first_minimum():
    """a constant shift is inverted by one fixed-point step; with u = 0 and a landmark half-way between four grid points
    MATLAB's min takes the first of the equidistant candidates (column-major order)"""
    from ma_amd import engine as E
    m, n = 32, 40
    omega = [0.0, float(m), 0.0, float(n)]
    u = np.zeros((m, n, 2))
    u[:, :, 0] = 1.5
    u[:, :, 1] = -2.25
    LM = np.array([[10.0, 20.0], [16.5, 11.25], [20.0, 30.5]])
    out, failed = E.landmark_transform(LM, u, omega)
    assert not failed.any()
    assert np.allclose(out, LM - np.array([1.5, -2.25]), atol=1e-12)
    from oracle import evaluation as oe
    LM2 = np.array([[8.0, 12.0], [0.2, 0.3]])
    ref = oe.landmark_transform(LM2, np.zeros((m, n, 2)), omega)
    out2, _ = E.landmark_transform(LM2, np.zeros((m, n, 2)), omega)
    assert np.array_equal(out2, ref)


def test_landmark_accuracy_matches_oracle():
    from ma_amd import evaluation as ev, synth
    from oracle import evaluation as oe
    LM = synth.dynamic_test_landmarks(60, 50, 5)
    cells = [LM[:, :, i] for i in range(5)]
    assert np.allclose(ev.landmark_accuracy(cells), oe.landmark_accuracy(cells), rtol=1e-14, atol=1e-14)
    assert np.allclose(ev.landmark_accuracy([cells[0]] * 3), 0.0)
    assert np.allclose(ev.landmark_accuracy([np.array([[0.0, 0.0]]), np.array([[2.0, 0.0]])]), [1.0])


def test_demo_evaluate_and_save_sections(tmp_path):
    """driver -> evaluate results -> save results -> load: the file carries the demo's workspace
    (NNE/demo_mf_nn_tv_registration_no_ref_ml.m:106-142)"""
    from ma_amd import drivers, evaluation as ev, synth
    import time
    m, n, k = 40, 36, 3
    img = [a for a in synth.dynamic_test_image(m, n, k).numpy()]
    LMs = synth.dynamic_test_landmarks(m, n, k)
    LM = [LMs[:, :, i] for i in range(k)]
    optPara = dict(theta=1.0, maxIter=60, tol=1e-3, outerIter=[2, 2], mu=0.1, bc="neumann", nu_factor=[0.9, 0.9], doPlots=False)
    t0 = time.perf_counter()
    u, Lc, SV_history = drivers.mf_nn_tv_registration_no_ref_ml(img, optPara)
    rt = time.perf_counter() - t0
    uStar, L = u[-1][optPara["outerIter"][1] - 1], Lc[-1][optPara["outerIter"][1] - 1]      # demo :110-111
    res = ev.evaluate_results(img, uStar, L, LM=LM)
    assert len(res["img_u"]) == k and res["img_u"][0].shape == (m, n) and res["LStar"].shape == (m, n, k)
    assert res["LM_acc"].shape == (17,) and res["LM_transformed_acc"].shape == (17,)
    path = ev.save_results(str(tmp_path / "results"), "synthetic", "demo_mf_nn_tv_registration_no_ref_ml", img, optPara, uStar,
                           L, SV_history, rt, res, LM=LM)
    assert path.endswith("res_synthetic_mf_nn_tv.mat")
    ws = ev.load_results(path)
    for name in ("dataset", "method", "img", "optPara", "uStar", "L", "SV_history", "runtime_registration", "LStar", "img_u",
                 "LM", "LM_transformed", "LM_acc", "LM_transformed_acc", "omega", "m", "n", "k"):
        assert name in ws, name
    assert ws["uStar"].shape == (m * n, 2, k) and ws["L"].shape == (m * n, k)
    assert np.array_equal(ws["uStar"], np.asarray(uStar).reshape((m * n, 2, k), order="F"))
    assert ws["img_u"].shape == (k, 1) and np.array_equal(ws["img_u"][1, 0], res["img_u"][1])
    assert ws["LM"].shape == (1, k) and np.array_equal(ws["LM_transformed"][2, 0], res["LM_transformed"][2])
    assert ws["optPara"][0, 0].bc[0] == "neumann"
