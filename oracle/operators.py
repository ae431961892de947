"""Linear operators, warping and pyramid of the reference (TEST INFRASTRUCTURE, see oracle/__init__.py).

Matrices are assembled with kron/diags exactly as the MATLAB sources do, so
index and boundary logic is the reference's by construction.
"""
import numpy as np
import scipy.sparse as sp
import scipy.signal


def _vecF(a):
    return np.asarray(a, dtype=np.float64).reshape(-1, order="F")


# --------------------------------------------------------------------------
# NNE/finite_difference_operator.m:29-50
def _fd_1d(m, h, bc):
    # Dx = (1/h) * spdiags([-ones, ones], 0:1, m, m)           (:29)
    D = sp.lil_matrix((1.0 / h) * sp.diags([-np.ones(m), np.ones(m - 1)], [0, 1], shape=(m, m)))
    if bc == "linear":
        # Dx(m, [m-1, m]) = [-1, 1]   -- NOT scaled by 1/h       (:31)
        D[m - 1, m - 2] = -1.0
        D[m - 1, m - 1] = 1.0
    elif bc == "neumann":
        D[m - 1, m - 1] = 0.0                                  # (:33)
    else:
        raise ValueError("Unknown boundary condition!")        # (:35)
    return sp.csr_matrix(D)


def finite_difference_operator(m, n, h_grid, k=1, bc="linear"):
    """sparse(4kmn x 2kmn), NNE/finite_difference_operator.m:29-50."""
    Dx = _fd_1d(m, h_grid[0], bc)
    Dy = _fd_1d(n, h_grid[1], bc)
    G1 = sp.vstack([sp.kron(sp.identity(n), Dx), sp.kron(Dy, sp.identity(m))])   # (:49)
    G = sp.kron(sp.identity(2), G1)
    D = sp.kron(sp.identity(k), G)                                               # (:50)
    D = sp.csr_matrix(D)
    D.eliminate_zeros()
    return D


def pairwise_gradient(m, n, h, ncomp):
    """K of the pairwise scripts: kron(I_ncomp, [Gx; Gy]) with Neumann rows.
    PDO/test_TV_L1_registration.m:67-72 (ncomp=2), PDO/test_TV_L1_denoising.m:38-43 (ncomp=1)."""
    Dx = _fd_1d(m, h[0], "neumann")
    Dy = _fd_1d(n, h[1], "neumann")
    G1 = sp.vstack([sp.kron(sp.identity(n), Dx), sp.kron(Dy, sp.identity(m))])
    K = sp.csr_matrix(sp.kron(sp.identity(ncomp), G1))
    K.eliminate_zeros()
    return K


# --------------------------------------------------------------------------
# NNE/discrete_laplacian.m:23-46
def _lap_1d(m, h, bc):
    e = np.ones(m)
    D = sp.lil_matrix((1.0 / h) ** 2 * sp.diags([e[:-1], -2 * e, e[:-1]], [-1, 0, 1], shape=(m, m)))
    if bc == "neumann":
        D[0, 0] = -1.0 / h ** 2                                # (:26)
        D[m - 1, m - 1] = -1.0 / h ** 2                        # (:27)
    elif bc == "dirichlet":
        pass                                                   # (:28)
    else:
        # reference only warns (:30) and continues with the dirichlet stencil
        import warnings
        warnings.warn("Unknown boundary condition")
    return sp.csr_matrix(D)


def discrete_laplacian(m, n, h_grid, k, bc):
    """sparse(2kmn x 2kmn), NNE/discrete_laplacian.m:23-46."""
    Dxx = _lap_1d(m, h_grid[0], bc)
    Dyy = _lap_1d(n, h_grid[1], bc)
    L1 = sp.kron(sp.identity(n), Dxx) + sp.kron(Dyy, sp.identity(m))
    return sp.csr_matrix(sp.kron(sp.identity(2 * k), L1))                        # (:46)


# --------------------------------------------------------------------------
# NNE/mean_free_operator.m:20
def mean_free_operator(m, n, k):
    return sp.csr_matrix(sp.kron(sp.identity(k) - np.ones((k, k)) / k, sp.identity(m * n)))


# --------------------------------------------------------------------------
# NNE/matrix_norm.m:15-48
def matrix_norm(S, x0=None, rng=None):
    """Power iteration on S'S.  The reference starts from randn (:22) and is therefore
    non-deterministic; `x0` injects the start vector (it is normalised here as in :23)."""
    tol = 1e-6
    maxIter = 100
    if x0 is None:
        rng = np.random.default_rng() if rng is None else rng
        x = rng.standard_normal(S.shape[1])
    else:
        x = np.array(x0, dtype=np.float64).reshape(-1)
    x = x / np.linalg.norm(x)
    St = sp.csr_matrix(S.T)
    e = []
    i = 0
    while (i < 2) or (abs(e[i - 1] - e[i - 2]) > tol * e[i - 1]):                # (:25)
        i += 1
        y = St @ (S @ x)                                                         # (:31-32)
        normY = np.linalg.norm(y)
        x = y / normY
        e.append(np.sqrt(normY))                                                 # (:40)
        if i == maxIter:
            break
    return np.array(e)


# --------------------------------------------------------------------------
# IRE/cell_centered_grid.m:20-29
def cell_centered_grid(omega, s):
    m, n = int(s[0]), int(s[1])
    h_x = (omega[1] - omega[0]) / m
    h_y = (omega[3] - omega[2]) / n
    xs = omega[0] + h_x * (0.5 + np.arange(m))
    ys = omega[2] + h_y * (0.5 + np.arange(n))
    x = np.repeat(xs[:, None], n, axis=1)
    y = np.repeat(ys[None, :], m, axis=0)
    return x, y


# IRE/bilinear_interpolation.m:22-105
def bilinear_interpolation(img, h, p):
    img = np.asarray(img, dtype=np.float64)
    m, n = img.shape
    l = p.shape[0]
    # q = W * [p, 1]'  with W = [1/h1 0 1/2; 0 1/h2 1/2; 0 0 1]                  (:29-35)
    q = np.empty((l, 2))
    q[:, 0] = (1.0 / h[0]) * p[:, 0] + 0.5
    q[:, 1] = (1.0 / h[1]) * p[:, 1] + 0.5
    img_p = np.zeros(l)
    dimg_p = np.zeros((l, 2))
    sub_tl = np.floor(q)                                                         # (:42)
    chi = q - sub_tl                                                             # (:64)
    imgF = img.reshape(-1, order="F")

    corners = (
        # (di, dj, weight for value, d/dx factor, d/dy factor)                  (:66-105)
        (0, 0, lambda c1, c2: (1 - c1) * (1 - c2), lambda c1, c2: -(1 - c2), lambda c1, c2: -(1 - c1)),
        (1, 0, lambda c1, c2: c1 * (1 - c2),       lambda c1, c2: +(1 - c2), lambda c1, c2: -c1),
        (0, 1, lambda c1, c2: (1 - c1) * c2,       lambda c1, c2: -c2,       lambda c1, c2: +(1 - c1)),
        (1, 1, lambda c1, c2: c1 * c2,             lambda c1, c2: +c2,       lambda c1, c2: +c1),
    )
    for di, dj, wv, wdx, wdy in corners:
        si = sub_tl[:, 0] + di
        sj = sub_tl[:, 1] + dj
        valid = (1 <= si) & (si <= m) & (1 <= sj) & (sj <= n)                    # (:48-55) zero padding
        idx = (si[valid].astype(np.int64) - 1) + (sj[valid].astype(np.int64) - 1) * m
        c1 = chi[valid, 0]
        c2 = chi[valid, 1]
        v = imgF[idx]
        img_p[valid] += wv(c1, c2) * v
        dimg_p[valid, 0] += (1.0 / h[0]) * wdx(c1, c2) * v
        dimg_p[valid, 1] += (1.0 / h[1]) * wdy(c1, c2) * v
    return img_p, dimg_p


# IRE/evaluate_displacement.m:24-49
def evaluate_displacement(img, h, u, omega=None, s_grid=None):
    """Returns (img_u (k x l), g (k*l x 2)).  The reference wraps g into the sparse matrix
    [diag(g(:,1)) diag(g(:,2))] (:47-48); `grad_matrix` builds that form when a caller needs it."""
    img = np.asarray(img, dtype=np.float64)
    m, n = img.shape
    if s_grid is None:
        s_grid = (m, n)
    k, l = int(s_grid[0]), int(s_grid[1])
    if omega is None:
        omega = [0.0, m * h[0], 0.0, n * h[1]]
    x, y = cell_centered_grid(omega, (k, l))
    p = np.stack([_vecF(x), _vecF(y)], axis=1)
    u = np.asarray(u, dtype=np.float64)
    if u.ndim == 0 or u.size == 1:
        p_displaced = p + float(u)
    else:
        p_displaced = p + u.reshape(k * l, 2, order="F") if u.ndim == 1 else p + u
    img_p, dimg_p = bilinear_interpolation(img, h, p_displaced)
    return img_p.reshape((k, l), order="F"), dimg_p


def grad_matrix(g):
    kl = g.shape[0]
    return sp.hstack([sp.diags(g[:, 0]), sp.diags(g[:, 1])], format="csr")


# --------------------------------------------------------------------------
# pyramid: NNE/mf_nn_tv_registration_no_ref_ml.m:59,67-79
def num_levels(m, n):
    return int(min(np.floor(np.log2([m, n]) - 5))) + 1                           # (:59)


def downsample(img):
    """conv2(img, 0.25*ones(2), 'same') then (1:2:end, 1:2:end)                    (:75-76).
    'same' with a 2x2 kernel keeps rows/cols 2..end of the full convolution, i.e. the block
    mean of (i,i+1)x(j,j+1) with zeros past the end."""
    full = scipy.signal.convolve2d(np.asarray(img, dtype=np.float64), 0.25 * np.ones((2, 2)), mode="full")
    same = full[1:, 1:]
    return same[0::2, 0::2].copy()


def pyramid(img, levels):
    ML = [None] * levels
    ML[levels - 1] = np.asarray(img, dtype=np.float64)
    for lev in range(levels - 2, -1, -1):
        ML[lev] = downsample(ML[lev + 1])
    return ML


def prolong_fields(x, m_low, n_low, m, n, nfields, factor=1.0):
    """kron(x_low, ones(2)) cropped to (1:m, 1:n) for every field
    (NNE/mf_nn_tv_registration_no_ref_ml.m:113-119; var drivers multiply by 2,
    NNE/var_tv_registration_no_ref_ml.m:112)."""
    xl = np.asarray(x).reshape((m_low, n_low, nfields), order="F")
    xh = np.zeros((2 * m_low, 2 * n_low, nfields))
    for i in range(nfields):
        xh[:, :, i] = np.kron(xl[:, :, i], np.ones((2, 2)))
    return factor * xh[:m, :n, :].reshape(-1, order="F")
