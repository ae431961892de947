"""Prox / value / conjugate operators of the reference (TEST INFRASTRUCTURE, see oracle/__init__.py).

Every function returns (res1, res2, res3); entries not computed for the given
`nargout` are None, mirroring the MATLAB nargout branches.
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from .operators import evaluate_displacement, grad_matrix


def _cols(v, c):
    return np.asarray(v, dtype=np.float64).reshape((-1, c), order="F")


# --------------------------------------------------------------------------
# NNE/SAD.m:32-87      mu * || L - I ||_1
def SAD(L, I, mu, sigma, conjugate_flag=False, nargout=2):
    L = np.asarray(L, dtype=np.float64)
    I = np.asarray(I, dtype=np.float64)
    if not conjugate_flag:
        if nargout == 3:
            res3 = np.zeros_like(L)
            diff_LI = L - I
            idx1 = diff_LI > (sigma * mu)                                        # (:44)
            idx2 = diff_LI < (-1) * (sigma * mu)                                 # (:45)
            idx3 = ~(idx1 | idx2)
            res3[idx1] = L[idx1] - (sigma * mu)
            res3[idx2] = L[idx2] + (sigma * mu)
            res3[idx3] = I[idx3]
            return None, None, res3
        return mu * np.sum(np.abs(L - I)), 0.0, None                             # (:54-57)
    if nargout == 3:
        _, _, prox = SAD(L / sigma, I, mu, 1.0 / sigma, False, 3)                # (:71)
        return None, None, L - sigma * prox                                      # (:72)
    mx = np.max(np.abs(L))
    res2 = mx - mu if mx > mu else 0.0                                           # (:77-81)
    res1 = mu * np.dot(L / mu, I)                                                # (:83)
    return res1, res2, None


# --------------------------------------------------------------------------
# NNE/SSD.m:35-71      0.5 * lambda * || x - g ||^2
def SSD(x, g, lam, tau, conjugate_flag, nargout=2):
    x = np.asarray(x, dtype=np.float64)
    res2 = 0.0
    if not conjugate_flag:
        if nargout == 3:
            return None, res2, (x + (lam * tau) * g) / (1 + (lam * tau))         # (:43)
        return lam * 0.5 * np.sum((x - g) ** 2), res2, None                      # (:48)
    if nargout == 3:
        _, _, prox = SSD(x / tau, g, lam, 1.0 / tau, False, 3)                   # (:60)
        return None, res2, x - tau * prox                                        # (:61)
    res1 = lam * (0.5 * np.sum((x / lam) ** 2) + np.dot(x / lam, g))             # (:66-67)
    return res1, res2, None


# --------------------------------------------------------------------------
# NNE/norm21.m:31-92   mu * || . ||_{2,1} over 4 components
def norm21(v, mu, sigma, conjugate_flag=False, nargout=2):
    v = np.asarray(v, dtype=np.float64)
    if not conjugate_flag:
        if nargout == 3:
            _, _, conj_prox = norm21(v.reshape(-1) / sigma, mu, 1.0 / sigma, True, 3)   # (:39)
            return None, None, v.reshape(-1) - sigma * conj_prox
        V = _cols(v, 4)
        norm_v = np.sqrt(np.sum(V ** 2, axis=1))
        return mu * np.sum(norm_v), 0.0, None                                    # (:56)
    V = _cols(v, 4)
    norm_v = np.sqrt(np.sum(V ** 2, axis=1))
    if nargout == 3:
        nn = np.maximum(norm_v, mu)                                              # (:70)
        return None, None, ((mu * V) / nn[:, None]).reshape(-1, order="F")       # (:71-72)
    mx = np.max(norm_v)
    return 0.0, (mx - mu if mx > mu else 0.0), None                              # (:81-88)


# --------------------------------------------------------------------------
# NNE/svd_2x2.m:8-47   rows of M are [m11 m21 m12 m22]
def svd_2x2(M):
    M = _cols(M, 4)
    a = M[:, 0]
    c = M[:, 1]
    b = M[:, 2]
    d = M[:, 3]
    s1 = a ** 2 + b ** 2 + c ** 2 + d ** 2
    s2 = np.sqrt((a ** 2 + b ** 2 - c ** 2 - d ** 2) ** 2 + 4 * (a * c + b * d) ** 2)
    sigma1 = np.sqrt(np.maximum((s1 + s2) / 2, 0))
    sigma2 = np.sqrt(np.maximum((s1 - s2) / 2, 0))
    theta = np.arctan2(2 * a * c + 2 * b * d, a ** 2 + b ** 2 - c ** 2 - d ** 2) / 2
    u11 = np.cos(theta)
    u21 = np.sin(theta)
    u12 = -np.sin(theta)
    u22 = np.cos(theta)
    phi = np.arctan2(2 * a * b + 2 * c * d, a ** 2 - b ** 2 + c ** 2 - d ** 2) / 2
    c_phi = np.cos(phi)
    s_phi = np.sin(phi)
    s11 = (a * u11 + c * u21) * c_phi + (b * u11 + d * u21) * s_phi
    s22 = (a * u21 - c * u11) * s_phi + (-b * u21 + d * u11) * c_phi
    sign1 = np.sign(s11)
    sign2 = np.sign(s22)
    v11 = sign1 * c_phi
    v21 = sign1 * s_phi
    v12 = -sign2 * s_phi
    v22 = sign2 * c_phi
    U = np.stack([u11, u21, u12, u22], axis=1)
    Sigma = np.stack([sigma1, sigma2], axis=1)
    V = np.stack([v11, v21, v12, v22], axis=1)
    return U, Sigma, V


# NNE/pointwise_2x2_nn.m:4-61   mu * sum_i || J_i ||_*
def pointwise_2x2_nn(M, mu, tau, conjugate_flag, nargout=2):
    if not conjugate_flag:
        if nargout == 3:
            Mm = _cols(M, 4)
            U, Sigma, V = svd_2x2(Mm)
            St = np.maximum(Sigma - mu * tau, 0)                                 # (:16)
            res3 = np.zeros_like(Mm)
            res3[:, 0] = St[:, 0] * U[:, 0] * V[:, 0] + St[:, 1] * U[:, 2] * V[:, 2]
            res3[:, 1] = St[:, 0] * U[:, 1] * V[:, 0] + St[:, 1] * U[:, 3] * V[:, 2]
            res3[:, 2] = St[:, 0] * U[:, 0] * V[:, 1] + St[:, 1] * U[:, 2] * V[:, 3]
            res3[:, 3] = St[:, 0] * U[:, 1] * V[:, 1] + St[:, 1] * U[:, 3] * V[:, 3]
            return None, None, res3.reshape(-1, order="F")
        _, Sigma, _ = svd_2x2(M)
        return mu * np.sum(Sigma), 0.0, None                                     # (:32-33)
    if nargout == 3:
        Mv = np.asarray(M, dtype=np.float64).reshape(-1, order="F")
        _, _, conj_prox = pointwise_2x2_nn(Mv / tau, mu, 1.0 / tau, False, 3)    # (:46-47)
        return None, None, Mv - tau * conj_prox                                  # (:48)
    _, Sigma, _ = svd_2x2(M)
    res2 = max(np.max((Sigma[:, 0] - mu) / mu), 0.0)                             # (:56)
    return 0.0, res2, None


# --------------------------------------------------------------------------
# NNE/l1ball_projection.m:23-68   Duchi et al., pivot = first unprocessed index
def l1ball_projection(x):
    x = np.asarray(x, dtype=np.float64).reshape(-1)
    v = np.abs(x)
    n = x.size
    if np.sum(v) <= 1:
        return x.copy()
    s = 0.0
    rho = 0
    U = np.ones(n, dtype=bool)
    while U.any():
        k = int(np.argmax(U))                                                    # find(U, 1)
        G = (v >= v[k]) & U
        Lset = (~G) & U
        d_rho = int(np.sum(G))
        d_s = float(np.sum(v[G]))
        if (s + d_s) - (rho + d_rho) * v[k] < 1:                                 # (:52)
            s = s + d_s
            rho = rho + d_rho
            U = Lset
        else:
            U = G
            U[k] = False
    theta = (s - 1) / rho                                                        # (:64)
    w = np.maximum(v - theta, 0)
    return np.sign(x) * w


# NNE/nuclear_norm_constraint.m:34-96   delta{ || . ||_* <= nu }
def nuclear_norm_constraint(L, numImg, tau, nu, conjugate_flag=False, nargout=2):
    Lm = np.asarray(L, dtype=np.float64).reshape((-1, numImg), order="F")
    U, S, Vt = np.linalg.svd(Lm, full_matrices=False)                            # svd(L,'econ') (:37)
    if not conjugate_flag:
        if nargout == 3:
            res3 = (U * (nu * l1ball_projection(S / nu))) @ Vt                   # (:45)
            return None, None, res3.reshape(-1, order="F")
        sS = np.sum(S)
        return 0.0, ((sS - nu) / nu if sS > nu else 0.0), None                   # (:55-62)
    if nargout == 3:
        mu = nu * tau                                                            # (:72)
        S_prox = S - mu * l1ball_projection(S / mu)                              # (:76)
        res3 = (U * S_prox) @ Vt                                                 # (:79)
        return None, None, res3.reshape(-1, order="F")
    return nu * np.max(S), 0.0, None                                             # (:89-92)


# --------------------------------------------------------------------------
# NNE/mean_zero_indicator.m:27-86
def mean_zero_indicator(u, s, conjugate_flag, nargout=2):
    u = np.asarray(u, dtype=np.float64)
    m, n, k = int(s[0]), int(s[1]), int(s[2])
    mn = m * n
    x_idx = np.tile(np.concatenate([np.ones(mn, bool), np.zeros(mn, bool)]), k)  # (:33)
    y_idx = ~x_idx
    u_x = u[x_idx]
    u_y = u[y_idx]
    N = float(mn * k)
    if not conjugate_flag:
        if nargout == 3:
            res3 = np.zeros_like(u)
            res3[x_idx] = u_x - (np.sum(u_x) / N)                                # (:50)
            res3[y_idx] = u_y - (np.sum(u_y) / N)                                # (:53)
            return None, None, res3
        return 0.0, max(abs(np.mean(u_x)), abs(np.mean(u_y))), None              # (:62-65)
    res3 = np.zeros_like(u)
    res3[x_idx] = np.sum(u_x) / N                                                # (:78)
    res3[y_idx] = np.sum(u_x) / N          # reference uses u_x here too          (:81)
    res2 = max(np.max(np.abs(u_x - res3[x_idx])), np.max(np.abs(u_y - res3[y_idx])))   # (:84)
    return 0.0, res2, res3


# NNE/zero_function.m:26-34
def zero_function(x, conjugate_flag, nargout=2):
    x = np.asarray(x, dtype=np.float64)
    if not conjugate_flag:
        return 0.0, 0.0, (x.copy() if nargout == 3 else None)
    return 0.0, float(np.max(np.abs(x))), (np.zeros_like(x) if nargout == 3 else None)


# --------------------------------------------------------------------------
# PDO/TV_registration.m:34-70 (ncomp=4), PDO/TV_denoise.m:34-70 (ncomp=2)
def _TV(v, sigma, conjugate_flag, nargout, ncomp):
    V = _cols(v, ncomp)
    norm_v = np.sqrt(np.sum(V ** 2, axis=1))
    res2 = 0.0
    if not conjugate_flag:
        res1 = np.sum(norm_v)                                                    # (:44)
        if nargout == 3:
            # the reference calls an undefined function TV(...) here (:51): dead branch
            raise NotImplementedError("prox of TV (non-conjugate) is a dead branch in the reference")
        return res1, res2, None
    mx = np.max(norm_v)
    if mx > 1:
        res2 = mx - 1                                                            # (:60-62)
    res3 = None
    if nargout == 3:
        nn = np.maximum(norm_v, 1)                                               # (:68)
        res3 = (V / nn[:, None]).reshape(-1, order="F")
    return 0.0, res2, res3


def TV_registration(v, sigma, conjugate_flag=False, nargout=2):
    return _TV(v, sigma, conjugate_flag, nargout, 4)


def TV_denoise(v, sigma, conjugate_flag=False, nargout=2):
    return _TV(v, sigma, conjugate_flag, nargout, 2)


# --------------------------------------------------------------------------
# PDO/SAD_denoise.m:36-79
def SAD_denoise(u, g, lam, tau, conjugate_flag=False, nargout=2):
    u = np.asarray(u, dtype=np.float64)
    res2 = 0.0
    if not conjugate_flag:
        res1 = lam * np.sum(np.abs(u - g))                                       # (:40)
        res3 = None
        if nargout == 3:
            res3 = np.zeros_like(u)
            diff_ug = u - g
            idx1 = diff_ug > lam * tau
            idx2 = diff_ug < (-1) * lam * tau
            idx3 = ~(idx1 | idx2)
            res3[idx1] = u[idx1] - lam * tau
            res3[idx2] = u[idx2] + lam * tau
            res3[idx3] = g[idx3]
        return res1, res2, res3
    u = u / lam                                                                  # (:60)
    mx = np.max(np.abs(u))
    if mx > 1:
        res2 = mx - 1
    res1 = lam * np.dot(u, g)                                                    # (:66)
    res3 = None
    if nargout == 3:
        _, _, prox = SAD_denoise(u / (lam * tau), g, lam, 1.0 / (lam * tau), False, 3)
        res3 = u - lam * tau * prox
    return res1, res2, res3


# --------------------------------------------------------------------------
# PDO/SAD_registration.m:36-153
def SAD_registration(u, u0, T, R, h, lam, tau, conjugate_flag=False, nargout=2):
    u = np.asarray(u, dtype=np.float64)
    u0 = np.asarray(u0, dtype=np.float64)
    res2 = 0.0
    # the reference re-warps T on every call (:39-40)
    Tu, g = evaluate_displacement(T, h, u0.reshape((-1, 2), order="F"))
    k = g.shape[0]
    gu0 = g[:, 0] * u0[:k] + g[:, 1] * u0[k:]
    b = Tu.reshape(-1, order="F") - gu0 - np.asarray(R, dtype=np.float64).reshape(-1, order="F")
    if not conjugate_flag:
        phi = b + (g[:, 0] * u[:k] + g[:, 1] * u[k:])                            # (:46)
        res1 = lam * np.sum(np.abs(phi))
        res3 = None
        if nargout == 3:
            ngs = np.sum(g ** 2, axis=1)
            idx1 = phi < (-tau) * lam * ngs                                      # (:58)
            idx2 = phi > tau * lam * ngs                                         # (:59)
            idx3 = ~(idx1 | idx2)
            idx3_1 = idx3 & (ngs > 1e-14)                                        # (:63)
            idx3_2 = idx3 & (ngs <= 1e-14)
            U = u.reshape((k, 2), order="F")
            r = np.zeros_like(U)
            r[idx1] = U[idx1] + tau * lam * g[idx1]
            r[idx2] = U[idx2] - tau * lam * g[idx2]
            r[idx3_1] = U[idx3_1] - g[idx3_1] * (phi[idx3_1] / ngs[idx3_1])[:, None]
            r[idx3_2] = U[idx3_2]
            res3 = r.reshape(-1, order="F")
        return res1, res2, res3
    # conjugate (:86-138)
    u = u / lam
    U = u.reshape((-1, 2), order="F")
    # spdiags(grad_T) (:97) drops an all-zero diagonal; that can only happen when a whole
    # gradient component vanishes identically, which the restatement treats as zeros.
    ngs = np.sum(g ** 2, axis=1)
    res1 = np.zeros(k)
    r2 = np.zeros(k)
    idx1 = ngs < 1e-14                                                           # (:101)
    res1[idx1] = (-1) * np.abs(b[idx1])
    r2[idx1] = np.sqrt(np.sum(U[idx1] ** 2, axis=1))
    idx2 = ~idx1
    p = (-b[idx2] / ngs[idx2])[:, None] * g[idx2]                                # (:114)
    res1[idx2] = np.sum(U[idx2] * p, axis=1)
    norm_grad = np.sqrt(ngs)
    u_rot = np.stack([g[:, 0] * U[:, 0] + g[:, 1] * U[:, 1],
                      -g[:, 1] * U[:, 0] + g[:, 0] * U[:, 1]], axis=1)           # (:120-121)
    u_rot[idx2] = u_rot[idx2] / norm_grad[idx2][:, None]
    idx2_1 = idx2 & (u_rot[:, 0] > norm_grad)
    r2[idx2_1] = np.sum((U[idx2_1] - g[idx2_1]) ** 2, axis=1)                    # squared (:126)
    idx2_2 = idx2 & (u_rot[:, 0] < -norm_grad)
    r2[idx2_2] = np.sum((U[idx2_2] + g[idx2_2]) ** 2, axis=1)                    # squared (:130)
    idx2_3 = idx2 & ~(idx2_1 | idx2_2)
    r2[idx2_3] = np.abs(u_rot[idx2_3, 1])                                        # plain   (:134)
    res1 = lam * np.sum(res1)
    res2 = np.max(r2)
    res3 = None
    if nargout == 3:
        _, _, prox = SAD_registration(u / (lam * tau), u0, T, R, h, lam, 1.0 / (lam * tau), False, 3)
        res3 = u - (lam * tau) * prox
    return res1, res2, res3


# --------------------------------------------------------------------------
# PDO/SSD_registration.m:36-120
def SSD_registration(u, u0, T, R, h, lam, tau, conjugate_flag=False, nargout=2):
    u = np.asarray(u, dtype=np.float64)
    u0 = np.asarray(u0, dtype=np.float64)
    res2 = 0.0
    k = np.asarray(T).size
    Tu, g = evaluate_displacement(T, h, u0.reshape((k, 2), order="F"))           # (:43)
    b = Tu.reshape(-1, order="F") - (g[:, 0] * u0[:k] + g[:, 1] * u0[k:]) \
        - np.asarray(R, dtype=np.float64).reshape(-1, order="F")
    if not conjugate_flag:
        residual = (g[:, 0] * u[:k] + g[:, 1] * u[k:]) + b
        res1 = 0.5 * lam * np.dot(residual, residual)                            # (:51)
        res3 = None
        if nargout == 3:
            D1 = sp.diags(g[:, 0])
            D2 = sp.diags(g[:, 1])
            I = sp.identity(k)
            A = sp.bmat([[I + (lam * tau) * D1.multiply(D1), (lam * tau) * D1.multiply(D2)],
                         [(lam * tau) * D1.multiply(D2), I + (lam * tau) * D2.multiply(D2)]], format="csc")
            c = u - (lam * tau) * np.concatenate([g[:, 0] * b, g[:, 1] * b])     # (:62)
            res3 = spla.spsolve(A, c)                                            # A \ c (:63)
        return res1, res2, res3
    u = u / lam                                                                  # (:70)
    ngs = np.sum(g ** 2, axis=1)
    idx = ngs > 1e-7                                                             # (:81)
    U = u.reshape((-1, 2), order="F")
    r1 = 0.5 * np.sum((U - b[:, None] * g) * g, axis=1) ** 2                     # (:84)
    r1[idx] = r1[idx] / (ngs[idx] ** 2)
    r1 = r1 - 0.5 * b ** 2
    res1 = lam * np.sum(r1)
    r2 = np.zeros(k)
    r2[~idx] = np.sqrt(np.sum(U[~idx] ** 2, axis=1))                             # (:96)
    gn = g.copy()
    gn[idx] = gn[idx] / np.sqrt(ngs[idx])[:, None]
    g_orth = np.stack([-gn[:, 1], gn[:, 0]], axis=1)
    r2[idx] = np.abs(np.sum(g_orth[idx] * U[idx], axis=1))                       # (:102)
    res2 = np.max(r2)
    if nargout == 3:
        # reference branch (:113-116) has a shape bug (u is mn x 2 there) and is never reached
        raise NotImplementedError("prox of SSD_registration conjugate is unreachable in the reference")
    return res1, res2, None


# --------------------------------------------------------------------------
# NNE/nuclear_norm.m:1-84   mu * || L ||_*  (soft constraint used by the archived drivers / test_nuclear_norm_measure.m)
def nuclear_norm(L, numImg, tau, mu, conjugate_flag=False, nargout=2):
    Lm = np.asarray(L, dtype=np.float64).reshape((-1, numImg), order="F")
    res2 = 0.0
    if not conjugate_flag:
        U, S, Vt = np.linalg.svd(Lm, full_matrices=False)
        res1 = mu * np.sum(np.abs(S))
        if nargout == 3:
            St = np.maximum(S - mu * tau, 0)
            return res1, res2, ((U * St) @ Vt).reshape(-1, order="F")
        return res1, res2, None
    Lm = Lm / mu
    S = np.linalg.svd(Lm, compute_uv=False)
    res1 = mu * 0.0
    if np.max(S) > 1:
        res2 = np.max(S) - 1
    if nargout == 3:
        Lv = Lm.reshape(-1, order="F")
        _, _, prox = nuclear_norm(Lv / (mu * tau), numImg, 1.0 / tau, 1.0 / mu, False, 3)
        return res1, res2, Lv - mu * tau * prox
    return res1, res2, None
