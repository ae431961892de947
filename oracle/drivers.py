"""Drivers / scripts of the reference restated (TEST INFRASTRUCTURE, see oracle/__init__.py)."""
import numpy as np
import scipy.sparse as sp

from . import operators as ops
from . import prox as px
from .chambolle_pock import chambolle_pock


def normalize(x):
    x = np.asarray(x, dtype=np.float64)
    return (x - x.min()) / (x.max() - x.min())


def normest_start(n):
    """Deterministic replacement for randn in NNE/matrix_norm.m:22 (the reference is
    non-deterministic there).  x_i = ((i * 2654435761) mod 2^32) / 2^31 - 1, i = 0..n-1,
    exactly representable in fp64, so the engine reproduces it bit for bit."""
    i = np.arange(n, dtype=np.uint64)
    hsh = (i * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)
    return hsh.astype(np.float64) / 2147483648.0 - 1.0


# --------------------------------------------------------------------------
# PDO/test_TV_L1_registration.m:56-122 / PDO/test_TV_L2_registration.m:56-122
def tv_registration(R, T, lam, tau, maxIter, numSteps, data_term="SAD", h=(1.0, 1.0), theta=1.0,
                    keep_histories=False):
    R = normalize(R)
    T = normalize(T)
    m, n = R.shape
    K = ops.pairwise_gradient(m, n, h, 2)
    L_squared = 4 * (1 / h[0] ** 2 + 1 / h[1] ** 2)                              # (:75)
    u0 = np.zeros(m * n * 2)
    v0 = np.zeros(m * n * 4)
    sigma = (1 - 1e-4) / (L_squared * tau)                                       # (:81)
    Gfun = px.SAD_registration if data_term == "SAD" else px.SSD_registration
    F = lambda v, c, nargout: px.TV_registration(v, sigma, c, nargout)
    hist = []
    for _ in range(numSteps):                                                    # (:94)
        G = (lambda u0_: (lambda u, c, nargout: Gfun(u, u0_, T, R, h, lam, tau, c, nargout)))(u0.copy())
        u_star, v_star, ph, dh = chambolle_pock(F, G, K, u0, v0, theta, tau, sigma, maxIter)   # tol default
        u0, v0 = u_star, v_star                                                  # (:119)
        if keep_histories:
            hist.append((ph, dh))
    return u0, v0, hist, dict(sigma=sigma, R=R, T=T)


# PDO/test_TV_L1_denoising.m:29-63
def tv_l1_denoising(img_noisy, lam=1.5, tau=0.02, maxIter=1000, tol=1e-5, h=(1.0, 1.0), theta=1.0):
    img_noisy = np.asarray(img_noisy, dtype=np.float64)
    m, n = img_noisy.shape
    K = ops.pairwise_gradient(m, n, h, 1)
    L_squared = 4 * (1 / h[0] ** 2 + 1 / h[1] ** 2)
    sigma = (1 - 1e-4) / (L_squared * tau)
    g = img_noisy.reshape(-1, order="F")
    G = lambda u, c, nargout: px.SAD_denoise(u, g, lam, tau, c, nargout)
    F = lambda v, c, nargout: px.TV_denoise(v, sigma, c, nargout)
    return chambolle_pock(F, G, K, np.zeros(m * n), np.zeros(2 * m * n), theta, tau, sigma, maxIter, tol) + (sigma,)


# --------------------------------------------------------------------------
# NNE/mf_nn_tv_registration_no_ref_ml.m / NNE/mf_nn_curvature_registration_no_ref_ml.m
def mf_nn_F(y, b, k, h_grid, mu, nu, sigma, conjugate_flag, nargout, reg, single_level=False):
    """Local function F of the drivers (mf_nn_tv...:235-305; mf_nn_curvature...:230-291).

    single_level: the F of NNE/mf_nn_tv_registration_no_ref.m:184-253 -- SAD(y1, b, 1, sigma) (:206,230) and
    norm21(y2_j, mu, sigma) per image (:211-216,236-241): weights 1 and mu (no prod(h_grid)), mu*||.||_{2,1} where the
    _ml driver has the pointwise 2x2 nuclear norm (mf_nn_tv..._ml.m:263-264,289-290 keep norm21 only as a comment)."""
    ncomp = 6 if reg == "tv" else 4
    mn = y.size // (ncomp * k)
    n2 = ncomp - 2
    y1 = y[: k * mn]
    y2 = y[k * mn: (n2 + 1) * k * mn]
    y3 = y[(n2 + 1) * k * mn:]
    hh = float(np.prod(h_grid))
    F2 = px.pointwise_2x2_nn
    if single_level:
        assert reg == "tv"
        hh = 1.0
        F2 = px.norm21
    if nargout == 3:
        res3 = np.zeros(ncomp * k * mn)
        res3[: k * mn] = px.SAD(y1, b, hh, sigma, conjugate_flag, 3)[2]
        if reg == "tv":
            Y2 = y2.reshape((4 * mn, k), order="F")
            for j in range(k):
                res3[k * mn + j * 4 * mn: k * mn + (j + 1) * 4 * mn] = \
                    F2(Y2[:, j], mu * hh, sigma, conjugate_flag, 3)[2]
        else:
            g0 = np.zeros(2 * k * mn)
            res3[k * mn: 3 * k * mn] = px.SSD(y2, g0, mu * hh, sigma, conjugate_flag, 3)[2]
        res3[(n2 + 1) * k * mn:] = px.nuclear_norm_constraint(y3, k, sigma, nu, conjugate_flag, 3)[2]
        return None, None, res3
    r1_F1, r2_F1, _ = px.SAD(y1, b, hh, sigma, conjugate_flag, 2)
    if reg == "tv":
        Y2 = y2.reshape((4 * mn, k), order="F")
        r1_F2 = 0.0
        r2_F2 = 0.0
        for j in range(k):
            a, c, _ = F2(Y2[:, j], mu * hh, sigma, conjugate_flag, 2)
            r1_F2 = r1_F2 + a
            r2_F2 = max(r2_F2, c)
    else:
        g0 = np.zeros(2 * k * mn)
        r1_F2, r2_F2, _ = px.SSD(y2, g0, mu * hh, sigma, conjugate_flag, 2)
    r1_F3, r2_F3, _ = px.nuclear_norm_constraint(y3, k, sigma, nu, conjugate_flag, 2)
    return np.array([r1_F1, r1_F2, r1_F3]), max(r2_F1, r2_F2, r2_F3), None


def mf_nn_G(x, s, conjugate_flag, nargout):
    """Local function G of the drivers (mf_nn_tv...:307-346)."""
    N = int(np.prod(s))
    x_u = x[: 2 * N]
    x_l = x[2 * N:]
    a1, c1, p1 = px.mean_zero_indicator(x_u, s, conjugate_flag, nargout)
    a2, c2, p2 = px.zero_function(x_l, conjugate_flag, nargout)
    if nargout == 3:
        return None, None, np.concatenate([p1, p2])
    return a1 + a2, max(c1, c2), None


def mf_nn_registration_no_ref_ml(img, optPara, reg="tv", normest_x0=normest_start, keep=None, single_level=False,
                                 max_levels=None):
    """[u, L, SV_history] = mf_nn_{tv,curvature}_registration_no_ref_ml(img, optPara).

    `keep` (dict) receives diagnostics per (lev, o): nu, tau, norm estimate, histories.
    The curvature driver returns only the last iterate (mf_nn_curvature...:8); here both return
    the full cell arrays and the caller picks."""
    k = len(img)
    img = [normalize(im) for im in img]                                          # (:49-51)
    m, n = img[0].shape
    omega = [0.0, float(m), 0.0, float(n)]
    # single_level: NNE/mf_nn_tv_registration_no_ref.m (one level, scalar outerIter / nu_factor, h = 1)
    numLevels = 1 if single_level else ops.num_levels(m, n)                      # (:59)
    if np.isscalar(optPara["outerIter"]):
        optPara = dict(optPara, outerIter=[optPara["outerIter"]] * 2)
    if np.isscalar(optPara["nu_factor"]):
        optPara = dict(optPara, nu_factor=[optPara["nu_factor"]] * 2)
    max_outer = int(max(optPara["outerIter"]))
    u = [[None] * max_outer for _ in range(numLevels)]
    L = [[None] * max_outer for _ in range(numLevels)]
    SV_history = [None] * numLevels
    ML = [ops.pyramid(im, numLevels) for im in img]                              # ML[i][lev]
    x = None
    nu = None
    ncomp = 6 if reg == "tv" else 4
    for lev in range(numLevels):
        if max_levels is not None and lev >= max_levels:     # test aid: stop after the coarsest levels (full-size checks)
            break
        m, n = ML[0][lev].shape
        h_grid = np.array([(omega[1] - omega[0]) / m, (omega[3] - omega[2]) / n])   # (:88)
        theta = optPara["theta"]
        tol = optPara["tol"]
        mu = optPara["mu"]
        bc = optPara["bc"]
        maxIter = optPara["maxIter"]
        sel = 0 if lev == 0 else 1
        outerIter = int(optPara["outerIter"][sel])
        nu_factor = optPara["nu_factor"][sel]
        if lev == 0:
            x = np.zeros(3 * k * m * n)
        else:
            m_low, n_low = ML[0][lev - 1].shape
            x = ops.prolong_fields(x, m_low, n_low, m, n, 3 * k)                 # (:111-119)
        p = np.zeros(ncomp * k * m * n)                                          # (:122)
        if reg == "tv":
            A2 = ops.finite_difference_operator(m, n, h_grid, k, bc)
        else:
            A2 = ops.discrete_laplacian(m, n, h_grid, k, bc)
        kmn = k * m * n
        A3 = sp.csr_matrix((kmn, 2 * kmn))
        A4 = sp.identity(kmn, format="csr")
        A5 = sp.csr_matrix((A2.shape[0], kmn))
        A6 = ops.mean_free_operator(m, n, k)
        s = (m, n, k)
        Gh = lambda xx, c, nargout: mf_nn_G(xx, s, c, nargout)
        u0 = x[: 2 * kmn].reshape((m * n, 2, k), order="F")
        SV_history[lev] = np.zeros((k, outerIter))                                # zeros(k, max(outerIter)), outerIter = this level's (:153)
        for o in range(outerIter):
            b = np.zeros(kmn)
            T_u = np.zeros((m, n, k))
            dT = []
            for i in range(k):
                T_u[:, :, i], g = ops.evaluate_displacement(ML[i][lev], h_grid, u0[:, :, i], omega)
                dT.append(ops.grad_matrix(g))
                b[i * m * n: (i + 1) * m * n] = T_u[:, :, i].reshape(-1, order="F") - \
                    dT[i] @ u0[:, :, i].reshape(-1, order="F")                   # (:167-168)
            if o == 0 and lev == 0:
                D = (A6 @ T_u.reshape(-1, order="F")).reshape((m * n, k), order="F")
                S = np.linalg.svd(D, compute_uv=False)
                nu = nu_factor * np.sum(S)                                       # (:173-175)
            elif o == 0 and lev > 0:
                nu = 2 * nu_factor * nu                                          # (:177)
            else:
                nu = nu_factor * nu                                              # (:179)
            A1 = -sp.block_diag(dT, format="csr")
            A = sp.bmat([[A1, A4], [A2, A5], [A3, A6]], format="csr")            # (:186-188)
            e = ops.matrix_norm(A, normest_x0(A.shape[1]) if normest_x0 is not None else None)
            norm_A_est = e[-1]
            tau = np.sqrt(0.99 / norm_A_est ** 2)                                # (:195-196)
            sigma = np.sqrt(0.99 / norm_A_est ** 2)
            nu_ = nu
            Fh = lambda y, c, nargout: mf_nn_F(y, b, k, h_grid, mu, nu_, sigma, c, nargout, reg, single_level)
            x, p, ph, dh = chambolle_pock(Fh, Gh, A, x, p, theta, tau, sigma, maxIter, tol)
            u0 = x[: 2 * kmn].reshape((m * n, 2, k), order="F")
            L0 = x[2 * kmn:]
            u[lev][o] = u0.copy()
            L[lev][o] = L0.reshape((m, n, k), order="F").copy()
            S = np.linalg.svd((A6 @ L0).reshape((m * n, k), order="F"), compute_uv=False)
            SV_history[lev][:, o] = S                                            # (:214-215)
            if keep is not None:
                keep[(lev, o)] = dict(nu=nu, tau=tau, sigma=sigma, norm_est=norm_A_est, n_power=len(e),
                                      primal_history=ph, dual_history=dh, p=p.copy(), x=x.copy(),
                                      b=b.copy(), m=m, n=n)
    return u, L, SV_history


def mf_nn_problem(ML_lev, u0, h_grid, omega, reg, bc):
    """Assemble (A, b, T_u, g) for one linearisation; helper for operator-level tests.
    Mirrors mf_nn_tv...:159-188."""
    k = len(ML_lev)
    m, n = ML_lev[0].shape
    kmn = k * m * n
    b = np.zeros(kmn)
    T_u = np.zeros((m, n, k))
    dT = []
    gs = []
    for i in range(k):
        T_u[:, :, i], g = ops.evaluate_displacement(ML_lev[i], h_grid, u0[:, :, i], omega)
        gs.append(g)
        dT.append(ops.grad_matrix(g))
        b[i * m * n: (i + 1) * m * n] = T_u[:, :, i].reshape(-1, order="F") - dT[i] @ u0[:, :, i].reshape(-1, order="F")
    if reg == "tv":
        A2 = ops.finite_difference_operator(m, n, h_grid, k, bc)
    else:
        A2 = ops.discrete_laplacian(m, n, h_grid, k, bc)
    A1 = -sp.block_diag(dT, format="csr")
    A = sp.bmat([[A1, sp.identity(kmn)], [A2, sp.csr_matrix((A2.shape[0], kmn))],
                 [sp.csr_matrix((kmn, 2 * kmn)), ops.mean_free_operator(m, n, k)]], format="csr")
    return A, b, T_u, gs


# --------------------------------------------------------------------------
# NNE/var_tv_registration_no_ref_ml.m / NNE/var_curvature_registration_no_ref_ml.m
def var_F(y, b, k, h_grid, mu, sigma, conjugate_flag, nargout, reg, ssd_swap=False):
    """Local function F of the variance drivers (var_tv...:201-260; var_curvature...:199-252); `b` is the second
    SSD argument as passed by the caller (the drivers pass -B*b, var_tv...:172)."""
    ncomp = 5 if reg == "tv" else 3
    mn = y.size // (ncomp * k)
    y1 = y[: k * mn]
    y2 = y[k * mn:]
    hh = float(np.prod(h_grid))
    if nargout == 3:
        res3 = np.zeros(ncomp * k * mn)
        if ssd_swap:    # NNE/var_tv_registration_no_ref.m:173 (single-level driver): lambda and tau exchanged
            res3[: k * mn] = px.SSD(y1, b, sigma, hh, conjugate_flag, 3)[2]
        else:
            res3[: k * mn] = px.SSD(y1, b, hh, sigma, conjugate_flag, 3)[2]
        if reg == "tv":
            Y2 = y2.reshape((4 * mn, k), order="F")
            for j in range(k):
                res3[k * mn + j * 4 * mn: k * mn + (j + 1) * 4 * mn] = \
                    px.norm21(Y2[:, j], mu * hh, sigma, conjugate_flag, 3)[2]
        else:
            res3[k * mn:] = px.SSD(y2, np.zeros(2 * k * mn), mu * hh, sigma, conjugate_flag, 3)[2]
        return None, None, res3
    r1_F1, r2_F1, _ = px.SSD(y1, b, hh, sigma, conjugate_flag, 2)
    if reg == "tv":
        Y2 = y2.reshape((4 * mn, k), order="F")
        r1_F2, r2_F2 = 0.0, 0.0
        for j in range(k):
            a, c, _ = px.norm21(Y2[:, j], mu * hh, sigma, conjugate_flag, 2)
            r1_F2 = r1_F2 + a
            r2_F2 = max(r2_F2, c)
    else:
        r1_F2, r2_F2, _ = px.SSD(y2, np.zeros(2 * k * mn), mu * hh, sigma, conjugate_flag, 2)
    return np.array([r1_F1, r1_F2]), max(r2_F1, r2_F2), None


def var_problem(ML_lev, u0, h_grid, omega, reg, bc):
    """(A, b_centred, T_u, g) for one linearisation: var_tv...:140-161 (A = [B*blkdiag(dT); D | Lap], b <- B*b)."""
    k = len(ML_lev)
    m, n = ML_lev[0].shape
    kmn = k * m * n
    b = np.zeros(kmn)
    T_u = np.zeros((m, n, k))
    dT, gs = [], []
    for i in range(k):
        T_u[:, :, i], g = ops.evaluate_displacement(ML_lev[i], h_grid, u0[:, :, i], omega)
        gs.append(g)
        dT.append(ops.grad_matrix(g))
        b[i * m * n: (i + 1) * m * n] = T_u[:, :, i].reshape(-1, order="F") - dT[i] @ u0[:, :, i].reshape(-1, order="F")
    Kc = ops.mean_free_operator(m, n, k)
    b = Kc @ b                                                                   # (:154)
    A1 = Kc @ sp.block_diag(dT, format="csr")                                    # (:157)
    if reg == "tv":
        A2 = ops.finite_difference_operator(m, n, h_grid, k, bc)                 # (:120)
    else:
        A2 = ops.discrete_laplacian(m, n, h_grid, k, bc)                         # var_curvature...:119 (bc: see driver)
    return sp.vstack([A1, A2], format="csr"), b, T_u, gs


def var_registration_no_ref_ml(img, optPara, reg="tv", normest_x0=normest_start, keep=None, single_level=False):
    """u = var_{tv,curvature}_registration_no_ref_ml(img, optPara)  (var_tv...:7-197).

    Quirks kept: the prolongation multiplies by 2 (:112). NNE/var_curvature...:119 calls discrete_laplacian without
    `bc` (an error in MATLAB as shipped); here optPara['bc'] is used for both regularisers."""
    k = len(img)
    img = [normalize(im) for im in img]
    m, n = img[0].shape
    omega = [0.0, float(m), 0.0, float(n)]
    # single_level: NNE/var_tv_registration_no_ref.m (one level, scalar outerIter, h = 1, swapped SSD prox arguments :173)
    numLevels = 1 if single_level else ops.num_levels(m, n)
    oi = optPara["outerIter"]
    oi = [int(oi), int(oi)] if np.isscalar(oi) else [int(v) for v in oi]
    max_outer = max(oi)
    u = [[None] * max_outer for _ in range(numLevels)]
    ML = [ops.pyramid(im, numLevels) for im in img]
    ncomp = 5 if reg == "tv" else 3
    x = None
    for lev in range(numLevels):
        m, n = ML[0][lev].shape
        h_grid = np.array([(omega[1] - omega[0]) / m, (omega[3] - omega[2]) / n])
        theta, tol, mu, bc, maxIter = optPara["theta"], optPara["tol"], optPara["mu"], optPara["bc"], optPara["maxIter"]
        outerIter = oi[0 if lev == 0 else 1]
        if lev == 0:
            x = np.zeros(2 * k * m * n)
        else:
            m_low, n_low = ML[0][lev - 1].shape
            x = ops.prolong_fields(x, m_low, n_low, m, n, 2 * k, factor=2.0)     # (:109-114)
        p = np.zeros(ncomp * k * m * n)
        s = (m, n, k)
        Gh = lambda xx, c, nargout: px.mean_zero_indicator(xx, s, c, nargout)    # (:126)
        u0 = x.reshape((m * n, 2, k), order="F")
        for o in range(outerIter):
            # evaluate_displacement is called WITHOUT omega here (var_tv...:147-148, var_curvature...:146-147): its
            # default [0, m*h1, 0, n*h2] (IRE/evaluate_displacement.m:34-36) is used, m*(M/m) in floating point
            A, b, _, _ = var_problem([ML[i][lev] for i in range(k)], u0, h_grid, None, reg, bc)
            e = ops.matrix_norm(A, normest_x0(A.shape[1]) if normest_x0 is not None else None)
            norm_A_est = e[-1]
            tau = np.sqrt(0.99 / norm_A_est ** 2)                                # (:168-169)
            sigma = np.sqrt(0.99 / norm_A_est ** 2)
            Fh = lambda y, c, nargout, b_=b, sg=sigma: var_F(y, -b_, k, h_grid, mu, sg, c, nargout, reg, ssd_swap=single_level)   # (:172)
            x, p, ph, dh = chambolle_pock(Fh, Gh, A, x, p, theta, tau, sigma, maxIter, tol)
            u0 = x.reshape((m * n, 2, k), order="F")
            u[lev][o] = u0.copy()
            if keep is not None:
                keep[(lev, o)] = dict(tau=tau, sigma=sigma, norm_est=norm_A_est, n_power=len(e), primal_history=ph,
                                      dual_history=dh, p=p.copy(), x=x.copy(), b=b.copy(), m=m, n=n)
    return u


# ------------------------------------------------------------------------------------------------
def nn_measure_problem(version, I, numImg, m, n, tau, sigma, param):
    """(F, G, K) of NNE/test_nuclear_norm_measure.m:127-151,212-238 for the image matrix I (mn x numImg). The script's
    SAD(L, I(:), sigma, c_flag) calls predate the mu argument of NNE/SAD.m:1; they are restated with mu = 1."""
    import scipy.sparse as sp
    from . import prox as px
    Iv = np.asarray(I, dtype=np.float64).reshape(-1, order="F")
    sad = lambda L, c, nargout: px.SAD(L, Iv, 1.0, sigma, c, nargout)
    if version == 1:
        K = sp.identity(m * n * numImg, format="csr")                                                    # (:129)
        G = lambda L, c, nargout: px.nuclear_norm(L, numImg, tau, param, c, nargout)                     # (:216)
        return sad, G, K
    if version == 2:
        K = sp.identity(m * n * numImg, format="csr")
        G = lambda L, c, nargout: px.nuclear_norm_constraint(L, numImg, tau, param, c, nargout)          # (:224-225)
        return sad, G, K
    K = ops.mean_free_operator(m, n, numImg)                                                             # (:137)
    F = lambda L, c, nargout: px.nuclear_norm_constraint(L, numImg, tau, param, c, nargout)              # (:234-235)
    return F, sad, K


def nuclear_norm_measure(img, version, answer, numFrames=51, maxIter=200, tol=1e-4, keep=None):
    """NNE/test_nuclear_norm_measure.m:73-251 as a function of the (already normalised, downsampled) frames img[0..1]:
    returns data_term (numFrames x 3), SSD, TV."""
    from .chambolle_pock import chambolle_pock
    from . import prox as px
    numImg = len(img)
    img = [np.asarray(a, dtype=np.float64).copy() for a in img]
    m, n = img[0].shape
    h_img = (1.0 / m, 1.0 / n)                                                                           # (:75)
    t = np.linspace(-1, 1, numFrames)                                                                    # (:84)
    omega = [0.0, 1.0, 0.0, 1.0]
    center = np.array([(omega[3] + omega[2]) / 2, (omega[1] + omega[0]) / 2])                            # (:94)
    xx, yy = ops.cell_centered_grid(omega, (m, n))
    X = np.stack([xx.reshape(-1, order="F"), yy.reshape(-1, order="F")], axis=1)
    if answer == 1:
        d = np.array([[1.0, -1.0]])
        d = d / np.max(np.abs(d))                                                                        # (:100-101)
    elif answer == 2:
        R = 0.375
        c_dist = np.sqrt((X[:, 0] - center[0]) ** 2 + (X[:, 1] - center[1]) ** 2)
        tmp = 1 - c_dist / R
        idx = tmp >= 0
        tmp[idx] = tmp[idx] ** 0.5
        tmp[~idx] = 0
        mask = tmp.reshape((m, n), order="F")
        img = [a * mask for a in img]                                                                    # (:106-115)
    h_grid = ((omega[1] - omega[0]) / m, (omega[3] - omega[2]) / n)
    theta, norm_K = 1.0, 1.0
    tau = np.sqrt((1 - 1e-4) / norm_K ** 2)                                                              # (:153-154)
    sigma = tau
    L0 = np.zeros(m * n * numImg)
    P0 = np.zeros(m * n * numImg)
    data_term = np.zeros((numFrames, 3))
    SSD = np.zeros(numFrames)
    TV = np.zeros(numFrames)
    A = ops.finite_difference_operator(m, n, h_grid, 1, "linear")                                        # (:165)
    I = np.zeros((m * n, numImg))
    I[:, 0] = img[0].reshape(-1, order="F")
    for i in range(numFrames):
        for j in range(numImg - 1):
            if answer == 1:
                u = t[i] * np.ones((m * n, 2)) * d[j]                                                    # (:179)
            elif answer == 2:
                alpha = t[i] * np.pi
                u = (X - center) @ np.array([[np.cos(alpha), np.sin(alpha)], [-np.sin(alpha), np.cos(alpha)]])
                u = u + center - X                                                                       # (:183-187)
            elif answer == 3:
                u = (1 - t[i]) ** 2 * (X - center) + center - X                                          # (:191-194)
            else:
                u = X - center
                u[:, 1] = u[:, 1] + t[i] * u[:, 0]
                u = u + center - X                                                                       # (:198-201)
            Tu, _ = ops.evaluate_displacement(img[j + 1], h_img, u, omega)                               # (:206-207)
            I[:, j + 1] = Tu.reshape(-1, order="F")
        if version == 1:
            param = np.sqrt(m * n)                                                                       # (:215)
        elif version == 2:
            param = 0.9 * np.sum(np.linalg.svd(I, compute_uv=False))                                     # (:222-223)
        else:
            Kb = ops.mean_free_operator(m, n, numImg)
            D = (Kb @ I.reshape(-1, order="F")).reshape((m * n, numImg), order="F")
            param = 0.9 * np.sum(np.linalg.svd(D, compute_uv=False))                                     # (:231-233)
        F, G, K = nn_measure_problem(version, I, numImg, m, n, tau, sigma, param)
        _, _, ph, dh = chambolle_pock(F, G, K, L0, P0, theta, tau, sigma, maxIter, tol)                  # (:241-242)
        data_term[i] = ph[-1, 0:3]                                                                       # (:243)
        SSD[i] = np.sum((I[:, 0] - I[:, 1]) ** 2)                                                        # (:246)
        TV[i] = px.TV_registration(A @ u.reshape(-1, order="F"), None, False, 2)[0]                      # (:249)
        if keep is not None:
            keep[i] = dict(I=I.copy(), param=param, ph=ph, dh=dh, tau=tau, sigma=sigma)
    return data_term, SSD, TV
