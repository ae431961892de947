"""the soft nuclear norm (NNE/nuclear_norm.m) and the L-only Chambolle-Pock problems of the landscape
experiment (NNE/test_nuclear_norm_measure.m) on the device, against the numpy oracle. fp64: 1e-10 relative."""
import numpy as np
import pytest

from oracle import drivers as od, prox as px
from oracle.chambolle_pock import chambolle_pock as cp_ref

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)


@pytest.mark.parametrize("npix,k", [(37 * 29, 2), (64 * 48, 5), (5000, 16), (300, 1)])
@pytest.mark.parametrize("conj", [False, True])
def test_nuclear_norm_operator(npix, k, conj):
    from ma_amd import engine as E
    rng = np.random.default_rng(npix + k)
    # a few dominant directions plus full-rank noise: thresholding removes some singular values and keeps others
    L = rng.standard_normal((npix, min(k, 2))) @ rng.standard_normal((min(k, 2), k)) + 0.3 * rng.standard_normal((npix, k))
    S = np.linalg.svd(L, compute_uv=False)
    mu, tau = 0.7, 0.6 * S[min(1, k - 1)] / 0.7 if not conj else 0.9
    if conj:
        mu = 0.8 * S[0]                       # max(svd(L / mu)) > 1: the violation measure is exercised
    r1, r2, r3 = E.nuclear_norm(L.reshape(-1, order="F"), k, tau, mu, conj, nargout=3)
    o1, o2, o3 = px.nuclear_norm(L.reshape(-1, order="F"), k, tau, mu, conj, nargout=3)
    assert abs(r1 - o1) <= 1e-10 * max(abs(o1), 1.0)
    assert abs(r2 - o2) <= 1e-10 * max(abs(o2), 1.0)
    assert _rel(r3, o3) < 1e-10
    q1, q2, q3 = E.nuclear_norm(L.reshape(-1, order="F"), k, tau, mu, conj, nargout=2)
    assert q3 is None and q1 == r1 and q2 == r2


def test_nuclear_norm_errors():
    from ma_amd import engine as E
    with pytest.raises(E.CpEngineError, match="unsupported"):
        E.nuclear_norm(np.zeros(17 * 4), 17, 1.0, 1.0)
    with pytest.raises(E.CpEngineError, match="badArg"):
        E.nuclear_norm(np.zeros(8), 2, 1.0, -1.0)
    with pytest.raises(E.CpEngineError, match="badArg"):
        E.nn_measure(4, np.zeros(8), 2, 1.0, 1.0, 1.0, 1.0, 5, 1e-3)


def _frames(m, n, k, seed):
    from ma_amd import synth
    imgs = synth.dynamic_test_image(m, n, max(k, 2)).numpy()[:k]
    rng = np.random.default_rng(seed)
    return np.stack([im.reshape(-1, order="F") for im in imgs], axis=1) + 0.02 * rng.standard_normal((m * n, k))


@pytest.mark.parametrize("version", [1, 2, 3])
@pytest.mark.parametrize("m,n,k", [(32, 24, 2), (40, 36, 4)])
def test_nn_measure_against_oracle(version, m, n, k):
    from ma_amd import engine as E
    I = _frames(m, n, k, 10 * version + k)
    tau = sigma = np.sqrt(1 - 1e-4)
    if version == 1:
        param = 0.25 * np.sqrt(m * n)
    elif version == 2:
        param = 0.9 * np.sum(np.linalg.svd(I, compute_uv=False))
    else:
        param = 0.9 * np.sum(np.linalg.svd(I - I.mean(axis=1, keepdims=True), compute_uv=False))
    rng = np.random.default_rng(5)
    L0 = 0.1 * rng.standard_normal(m * n * k)
    P0 = 0.1 * rng.standard_normal(m * n * k)
    if version == 3:
        P0 = (P0.reshape((m * n, k), order="F") - P0.reshape((m * n, k), order="F").mean(axis=1, keepdims=True)).reshape(-1, order="F")
    F, G, K = od.nn_measure_problem(version, I, k, m, n, tau, sigma, param)
    xr, yr, phr, dhr = cp_ref(F, G, K, L0, P0, 1.0, tau, sigma, 60, 1e-12)
    x, y, ph, dh, res = E.nn_measure(version, I, k, param, 1.0, tau, sigma, 60, 1e-12, L0=L0, P0=P0)
    assert ph.shape == phr.shape == (61, 5) and res.iterations == 60 and res.stop_reason == 0
    assert _rel(x, xr) < 1e-10 and _rel(y, yr) < 1e-10
    # values relative to the energy scale, constraint measures absolute (they are differences of O(1) quantities)
    scale = np.max(np.abs(phr[:, 0]))
    assert np.max(np.abs(ph[:, :3] - phr[:, :3])) < 1e-10 * scale
    assert np.max(np.abs(dh[:, :3] - dhr[:, :3])) < 1e-10 * max(scale, np.max(np.abs(dhr[:, 0])))
    assert np.max(np.abs(ph[:, 3:] - phr[:, 3:])) < 1e-9 and np.max(np.abs(dh[:, 3:] - dhr[:, 3:])) < 1e-9


@pytest.mark.parametrize("version", [1, 2, 3])
def test_nn_measure_stop_rule_and_defaults(version):
    """zero start (the script's L0, P0), the script's tolerance: same stopping iteration and rows as the oracle"""
    from ma_amd import engine as E
    m, n, k = 32, 32, 2
    I = _frames(m, n, k, 3)
    tau = sigma = np.sqrt(1 - 1e-4)
    param = {1: np.sqrt(m * n), 2: 0.9 * np.sum(np.linalg.svd(I, compute_uv=False)),
             3: 0.9 * np.sum(np.linalg.svd(I - I.mean(axis=1, keepdims=True), compute_uv=False))}[version]
    F, G, K = od.nn_measure_problem(version, I, k, m, n, tau, sigma, param)
    z = np.zeros(m * n * k)
    xr, yr, phr, dhr = cp_ref(F, G, K, z, z, 1.0, tau, sigma, 200, 1e-4)
    x, y, ph, dh, res = E.nn_measure(version, I, k, param, 1.0, tau, sigma, 200, 1e-4)
    assert ph.shape == phr.shape and res.iterations == phr.shape[0] - 1
    assert res.stop_reason == (1 if phr.shape[0] - 1 < 200 else 0)
    assert _rel(x, xr) < 1e-9 and abs(ph[-1, 0] - phr[-1, 0]) <= 1e-10 * abs(phr[-1, 0])


@pytest.mark.parametrize("version,answer", [(1, 1), (2, 2), (3, 4), (1, 3)])
def test_landscape_experiment(version, answer):
    from ma_amd import drivers as D, synth
    m = n = 32
    imgs = synth.dynamic_test_image(m, n, 2).numpy()
    img = [od.normalize(imgs[0]), od.normalize(imgs[1])]
    kd, ko = {}, {}
    dt, ssd, tv = D.nuclear_norm_measure(img, version, answer, numFrames=5, maxIter=60, tol=1e-4, keep=kd)
    dr, ssdr, tvr = od.nuclear_norm_measure(img, version, answer, numFrames=5, maxIter=60, tol=1e-4, keep=ko)
    for i in range(5):
        assert _rel(kd[i]["I"], ko[i]["I"]) < 1e-12
        assert abs(kd[i]["param"] - ko[i]["param"]) <= 1e-10 * ko[i]["param"]
        assert kd[i]["ph"].shape == ko[i]["ph"].shape
    assert np.max(np.abs(dt - dr)) <= 1e-9 * max(np.max(np.abs(dr)), 1.0)
    np.testing.assert_allclose(ssd, ssdr, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(tv, tvr, rtol=1e-12, atol=1e-10)
