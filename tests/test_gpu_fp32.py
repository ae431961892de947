"""fp32 mode of the pairwise models (CP_FLAG_FP32: float storage, fp64 arithmetic) against the fp64 mode and the oracle:
north_star tolerance 1e-4 relative on the displacement field and the final energy."""
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
    with pytest.raises(E.CpEngineError, match="unsupported"):
        E.Engine(E.MODEL_MF_NN_TV, 16, 16, k=3, flags=E.FLAG_FP32)
