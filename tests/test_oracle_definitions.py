"""The oracle's prox operators against the DEFINITION of a prox, not against a second restatement.

In the reference every functional has two independent code paths: the value (nargout = 2) and the prox (nargout = 3).
For a convex F, p = prox_{tF}(v) if and only if for every x

    F(x) + |x - v|^2 / (2t)  >=  F(p) + |p - v|^2 / (2t) + |x - p|^2 / (2t)

(strong convexity of the prox objective around its minimiser). The tests evaluate both sides with the oracle's own value
path (and the oracle's own feasibility residual for indicator functions) at random x near p, near v and far away: a prox
branch, threshold or scaling restated wrongly (or a value path that does not belong to it) fails the inequality. This
ties the two paths of each restated .m file to the mathematics the reference's header comments state; it cannot
replace vectors produced by MATLAB (there are none: "parity unpinned")."""
import numpy as np
import pytest

from oracle import prox as px


def _samples(rng, p, v, count=60):
    n = p.size
    for s in range(count):
        scale = (1e-3, 1e-2, 0.1, 1.0, 5.0)[s % 5]
        base = (p, v, np.zeros(n))[s % 3]
        yield base + scale * rng.standard_normal(n)
    yield v.copy()
    yield p.copy()


def _check(val, p, v, t, xs, tol=1e-10):
    base = val(p) + np.sum((p - v) ** 2) / (2 * t)
    worst = 0.0
    for x in xs:
        lhs = val(x) + np.sum((x - v) ** 2) / (2 * t)
        rhs = base + np.sum((x - p) ** 2) / (2 * t)
        worst = min(worst, (lhs - rhs) / max(1.0, abs(rhs)))
    assert worst >= -tol, worst


def test_sad_and_its_conjugate():
    rng = np.random.default_rng(1)
    n, mu, s = 40, 0.6, 0.35
    I, v = rng.standard_normal(n), 2 * rng.standard_normal(n)
    p = px.SAD(v, I, mu, s, False, 3)[2]
    _check(lambda x: px.SAD(x, I, mu, s, False, 2)[0], p, v, s, _samples(rng, p, v))
    q = px.SAD(v, I, mu, s, True, 3)[2]
    assert px.SAD(q, I, mu, s, True, 2)[1] <= 1e-15                    # feasible: |q| <= mu
    feas = (np.clip(x, -mu, mu) for x in _samples(rng, q, v))
    _check(lambda x: px.SAD(x, I, mu, s, True, 2)[0], q, v, s, feas)


def test_ssd_and_its_conjugate():
    rng = np.random.default_rng(2)
    n, lam, t = 30, 1.7, 0.4
    g, v = rng.standard_normal(n), 2 * rng.standard_normal(n)
    p = px.SSD(v, g, lam, t, False, 3)[2]
    _check(lambda x: px.SSD(x, g, lam, t, False, 2)[0], p, v, t, _samples(rng, p, v))
    q = px.SSD(v, g, lam, t, True, 3)[2]
    _check(lambda x: px.SSD(x, g, lam, t, True, 2)[0], q, v, t, _samples(rng, q, v))


def test_norm21_and_the_projection_of_its_conjugate():
    rng = np.random.default_rng(3)
    npx, mu, s = 25, 0.8, 0.3
    v = 2 * rng.standard_normal(4 * npx)
    p = px.norm21(v, mu, s, False, 3)[2]
    _check(lambda x: px.norm21(x, mu, s, False, 2)[0], p, v, s, _samples(rng, p, v))
    q = px.norm21(v, mu, s, True, 3)[2]
    assert px.norm21(q, mu, s, True, 2)[1] <= 1e-15

    def feas():
        for x in _samples(rng, q, v):
            X = x.reshape((-1, 4), order="F")
            nr = np.maximum(np.sqrt(np.sum(X ** 2, axis=1)) / mu, 1.0)
            yield (X / nr[:, None]).reshape(-1, order="F")
    _check(lambda x: 0.0, q, v, s, feas())


def test_pointwise_2x2_nuclear_norm_and_the_projection_of_its_conjugate():
    rng = np.random.default_rng(4)
    npx, mu, t = 25, 0.7, 0.45
    v = 2 * rng.standard_normal(4 * npx)
    v[:4 * 3:npx] = 0.0                                                 # some rank-deficient / zero rows
    p = px.pointwise_2x2_nn(v, mu, t, False, 3)[2]
    # the value path against numpy's SVD as well: rows are [m11 m21 m12 m22]
    M = v.reshape((-1, 4), order="F")
    nuc = sum(np.linalg.svd(np.array([[r[0], r[2]], [r[1], r[3]]]), compute_uv=False).sum() for r in M)
    assert np.isclose(px.pointwise_2x2_nn(v, mu, t, False, 2)[0], mu * nuc, rtol=1e-12)
    _check(lambda x: px.pointwise_2x2_nn(x, mu, t, False, 2)[0], p, v, t, _samples(rng, p, v))
    q = px.pointwise_2x2_nn(v, mu, t, True, 3)[2]
    assert px.pointwise_2x2_nn(q, mu, t, True, 2)[1] <= 1e-12          # spectral norm <= mu at every pixel

    def feas():
        for x in _samples(rng, q, v):
            X = x.reshape((-1, 4), order="F").copy()
            for r in X:
                s1 = np.linalg.svd(np.array([[r[0], r[2]], [r[1], r[3]]]), compute_uv=False)[0]
                if s1 > mu:
                    r *= mu / s1 * (1 - 1e-15)
            yield X.reshape(-1, order="F")
    _check(lambda x: 0.0, q, v, t, feas())


def test_tv_conjugate_is_the_projection_onto_unit_balls():
    rng = np.random.default_rng(5)
    for f, c in ((px.TV_registration, 4), (px.TV_denoise, 2)):
        v = 2 * rng.standard_normal(c * 20)
        q = f(v, 0.3, True, 3)[2]
        assert f(q, 0.3, True, 2)[1] <= 1e-15

        def feas():
            for x in _samples(rng, q, v):
                X = x.reshape((-1, c), order="F")
                nr = np.maximum(np.sqrt(np.sum(X ** 2, axis=1)), 1.0)
                yield (X / nr[:, None]).reshape(-1, order="F")
        _check(lambda x: 0.0, q, v, 1.0, feas())


def test_sad_denoise():
    rng = np.random.default_rng(6)
    n, lam, t = 35, 1.3, 0.25
    g, v = rng.standard_normal(n), 2 * rng.standard_normal(n)
    p = px.SAD_denoise(v, g, lam, t, False, 3)[2]
    _check(lambda x: px.SAD_denoise(x, g, lam, t, False, 2)[0], p, v, t, _samples(rng, p, v))


@pytest.mark.parametrize("k", [3, 6])
def test_nuclear_norm_ball_projection_and_conjugate(k):
    rng = np.random.default_rng(7)
    mn, t = 20, 0.4
    v = rng.standard_normal(mn * k)
    S = np.linalg.svd(v.reshape((mn, k), order="F"), compute_uv=False)
    nu = 0.6 * S.sum()
    p = px.nuclear_norm_constraint(v, k, t, nu, False, 3)[2]
    assert px.nuclear_norm_constraint(p, k, t, nu, False, 2)[1] <= 1e-12        # inside the ball (on its boundary)
    assert np.isclose(np.linalg.svd(p.reshape((mn, k), order="F"), compute_uv=False).sum(), nu, rtol=1e-12)

    def feas():
        for x in _samples(rng, p, v):
            s = np.linalg.svd(x.reshape((mn, k), order="F"), compute_uv=False).sum()
            yield x * min(1.0, nu / s * (1 - 1e-15))
    _check(lambda x: 0.0, p, v, 1.0, feas())
    # conjugate: F*(y) = nu * s_max(y), no constraint
    q = px.nuclear_norm_constraint(v, k, t, nu, True, 3)[2]
    _check(lambda x: px.nuclear_norm_constraint(x, k, t, nu, True, 2)[0], q, v, t, _samples(rng, q, v))
    # a point already inside the ball is left alone
    small = 0.5 * nu / S.sum() * v
    assert np.allclose(px.nuclear_norm_constraint(small, k, t, nu, False, 3)[2], small, atol=1e-14)


def test_soft_nuclear_norm():
    rng = np.random.default_rng(8)
    mn, k, t, mu = 18, 4, 0.7, 0.9
    v = rng.standard_normal(mn * k)
    p = px.nuclear_norm(v, k, t, mu, False, 3)[2]
    _check(lambda x: px.nuclear_norm(x, k, t, mu, False, 2)[0], p, v, t, _samples(rng, p, v))


def test_mean_zero_projection():
    rng = np.random.default_rng(9)
    m, n, k = 5, 4, 3
    v = rng.standard_normal(2 * m * n * k) + 0.3
    p = px.mean_zero_indicator(v, (m, n, k), False, 3)[2]
    assert px.mean_zero_indicator(p, (m, n, k), False, 2)[1] < 1e-15
    feas = (px.mean_zero_indicator(x, (m, n, k), False, 3)[2] for x in _samples(rng, p, v))
    _check(lambda x: 0.0, p, v, 1.0, feas)


@pytest.mark.parametrize("name", ["SAD_registration", "SSD_registration"])
def test_pairwise_data_terms(name):
    rng = np.random.default_rng(10)
    m, n = 9, 8
    yy, xx = np.meshgrid(np.arange(m), np.arange(n), indexing="ij")
    T = np.exp(-((yy - 4.2) ** 2 + (xx - 3.1) ** 2) / 9.0)
    R = np.exp(-((yy - 3.6) ** 2 + (xx - 3.9) ** 2) / 9.0)
    T[0, :] = 0.0                                                       # a flat row: zero gradient there (the ngs branches)
    h, lam, t = (1.0, 1.0), 0.8, 0.5
    u0 = 0.3 * rng.standard_normal(2 * m * n)
    v = u0 + 0.5 * rng.standard_normal(2 * m * n)
    f = getattr(px, name)
    p = f(v, u0, T, R, h, lam, t, False, 3)[2]
    _check(lambda x: f(x, u0, T, R, h, lam, t, False, 2)[0], p, v, t, _samples(rng, p, v, count=40))
