"""PDO/chambolle_pock.m restated (TEST INFRASTRUCTURE, see oracle/__init__.py)."""
import numpy as np
import scipy.sparse as sp


def chambolle_pock(F, G, K, x0, y0, theta=None, tau=None, sigma=None, maxIter=None, tol=None, log=None):
    """Returns (x_star, y_star, primal_history, dual_history); PDO/chambolle_pock.m:54-228.

    F, G: callables op(v, conjugate_flag, nargout) -> (res1, res2, res3); res1 may be a scalar or a
    row vector of component values (summed by the solver, :91,:170)."""
    K = sp.csr_matrix(K)
    if tol is None:
        tol = 1e-3                                                               # (:54)
    if maxIter is None:
        maxIter = 300                                                            # (:55)
    if sigma is None and tau is not None:
        K_abs = abs(K)
        L2 = K_abs.sum(axis=0).max() * K_abs.sum(axis=1).max()                   # (:58-60)
        sigma = 1.0 / (L2 * tau)
    if tau is None:
        K_abs = abs(K)
        Le = np.sqrt(K_abs.sum(axis=0).max() * K_abs.sum(axis=1).max())          # (:64-68)
        tau = 1.0 / Le
        sigma = tau
    if theta is None:
        theta = 1.0                                                              # (:70)

    Kt = sp.csr_matrix(K.T)                                                      # (:73)

    def primal_objective(x):                                                     # (:233-237)
        F_val, F_con, _ = F(K @ x, False, 2)
        G_val, G_con, _ = G(x, False, 2)
        return np.atleast_1d(F_val).astype(float), np.atleast_1d(G_val).astype(float), F_con, G_con

    def dual_objective(y):                                                       # (:239-243)
        FS_val, FS_con, _ = F(y, True, 2)
        GS_val, GS_con, _ = G(-(Kt @ y), True, 2)
        return np.atleast_1d(FS_val).astype(float), np.atleast_1d(GS_val).astype(float), FS_con, GS_con

    i = 0
    GAP = np.zeros(maxIter + 1)
    y_current = np.array(y0, dtype=np.float64)
    x_current = np.array(x0, dtype=np.float64)
    x_bar = x_current.copy()

    F_con = np.zeros(maxIter + 1)
    G_con = np.zeros(maxIter + 1)
    Fh, Gh, F_con[0], G_con[0] = primal_objective(x_current)
    F_history = np.zeros((maxIter + 1, Fh.size))
    G_history = np.zeros((maxIter + 1, Gh.size))
    F_history[0] = Fh
    G_history[0] = Gh
    primal_history = np.zeros(maxIter + 1)
    primal_history[0] = G_history[0].sum() + F_history[0].sum()

    FS_con = np.zeros(maxIter + 1)
    GS_con = np.zeros(maxIter + 1)
    FSh, GSh, FS_con[0], GS_con[0] = dual_objective(y_current)
    FStar_history = np.zeros((maxIter + 1, FSh.size))
    GStar_history = np.zeros((maxIter + 1, GSh.size))
    FStar_history[0] = FSh
    GStar_history[0] = GSh
    dual_history = np.zeros(maxIter + 1)
    dual_history[0] = -(GStar_history[0].sum() + FStar_history[0].sum())

    while True:                                                                  # (:134)
        if i == maxIter:
            break
        with np.errstate(divide="ignore", invalid="ignore"):
            GAP[i] = abs((primal_history[i] - dual_history[i]) / dual_history[i])   # (:142)
        if i >= 4:
            sl = slice(i - 4, i + 1)                                             # MATLAB i-3 : i+1
            if (not np.any(np.isnan(GAP[sl]))) and np.all(GAP[sl] <= tol) and \
               np.all(F_con[sl] <= tol) and np.all(G_con[sl] <= tol) and \
               np.all(FS_con[sl] <= tol) and np.all(GS_con[sl] <= tol):
                break
        i += 1
        x_old = x_current
        _, _, y_current = F(y_current + sigma * (K @ x_bar), True, 3)            # (:162)
        _, _, x_current = G(x_current - tau * (Kt @ y_current), False, 3)        # (:165)
        F_history[i], G_history[i], F_con[i], G_con[i] = primal_objective(x_current)
        primal_history[i] = G_history[i].sum() + F_history[i].sum()
        FStar_history[i], GStar_history[i], FS_con[i], GS_con[i] = dual_objective(y_current)
        dual_history[i] = -(GStar_history[i].sum() + FStar_history[i].sum())
        x_bar = x_current + theta * (x_current - x_old)                          # (:179)
        if log is not None and i % 10 == 0:
            log(i, primal_history[i], dual_history[i])

    n_rows = i + 1                                                               # trim (:205-216)
    primal = np.column_stack([primal_history[:n_rows], F_history[:n_rows], G_history[:n_rows],
                              F_con[:n_rows], G_con[:n_rows]])                   # (:225-226)
    dual = np.column_stack([dual_history[:n_rows], FStar_history[:n_rows], GStar_history[:n_rows],
                            FS_con[:n_rows], GS_con[:n_rows]])                   # (:227-228)
    if i == maxIter:
        # reference does not trim when maxIter is reached; all maxIter+1 rows are filled anyway
        pass
    return x_current, y_current, primal, dual
