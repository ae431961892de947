"""TEST INFRASTRUCTURE ONLY (the oracle): numpy restatement of the reference's evaluation utilities, landmark transfer
through a displacement field and the landmark accuracy measure. Same names, arguments and results as
NNE/landmark_transform.m and NNE/landmark_accuracy.m. The product path is ma_amd/evaluation.py -> cp_landmark_transform /
cp_landmark_accuracy (HIP); only tests/ may import this module. Parity unpinned: the reference stores no landmark
vectors."""
import warnings

import numpy as np


def _bilinear(img, h, p):
    """values of the cell-centred image `img` (m x n) at the world points p (N x 2), zero outside
    (IRE/bilinear_interpolation.m:29-64; value only)"""
    m, n = img.shape
    q1 = p[:, 0] / h[0] + 0.5
    q2 = p[:, 1] / h[1] + 0.5
    f1 = np.floor(q1)
    f2 = np.floor(q2)
    c1 = q1 - f1
    c2 = q2 - f2
    out = np.zeros(p.shape[0])
    for di in (0, 1):
        for dj in (0, 1):
            si = f1.astype(np.int64) + di
            sj = f2.astype(np.int64) + dj
            ok = (si >= 1) & (si <= m) & (sj >= 1) & (sj <= n)
            w = (c1 if di else 1.0 - c1) * (c2 if dj else 1.0 - c2)
            v = np.zeros(p.shape[0])
            v[ok] = img[si[ok] - 1, sj[ok] - 1]
            out += w * v
    return out


def landmark_transform(LM, u, omega):
    """LM_transformed = landmark_transform(LM, u, omega)   (NNE/landmark_transform.m:1-67)

    LM: (numLM, 2) landmark positions in the reference frame; u: (m, n, 2) displacement field. Solves
    F(x) = x + u(x) = y by the fixed-point iteration x <- x + (y - F(x)) from the nearest deformed grid point.
    The termination test is the reference's literal expression `sum(F(x) - LM(i,:).^2) < 1e-12` (:47,:52)."""
    LM = np.asarray(LM, dtype=np.float64)
    u = np.asarray(u, dtype=np.float64)
    m, n = u.shape[0], u.shape[1]
    hx = (omega[1] - omega[0]) / m
    hy = (omega[3] - omega[2]) / n

    def F(z):
        z2 = z.reshape(1, 2)
        return z + np.array([_bilinear(u[:, :, 0], (hx, hy), z2)[0], _bilinear(u[:, :, 1], (hx, hy), z2)[0]])

    xs = omega[0] + hx * (np.arange(1, m + 1) - 0.5)                     # IRE/cell_centered_grid.m:20-29
    ys = omega[2] + hy * (np.arange(1, n + 1) - 0.5)
    xx, yy = np.meshgrid(xs, ys, indexing="ij")
    p = np.column_stack([xx.reshape(-1, order="F"), yy.reshape(-1, order="F")])
    g = p + u.reshape((m * n, 2), order="F")
    out = np.zeros_like(LM)
    for i in range(LM.shape[0]):
        min_idx = int(np.argmin(np.sum((g - LM[i]) ** 2, axis=1)))
        y = LM[i]
        x = p[min_idx].copy()
        for _ in range(100):
            x = x + (y - F(x))
            if np.sum(F(x) - LM[i] ** 2) < 1e-12:
                break
        if np.sum(F(x) - LM[i] ** 2) >= 1e-12:
            warnings.warn("Fixed-point iteration failed! Returning initial guess.")
            x = p[min_idx].copy()
        out[i] = x
    return out


def landmark_accuracy(LM):
    """LM_acc = landmark_accuracy(LM)   (NNE/landmark_accuracy.m:1-28): LM is a list of k (numLM, 2) arrays; returns
    the mean distance of every landmark to its mean position over the k images."""
    y = np.stack([np.asarray(a, dtype=np.float64) for a in LM], axis=2)
    y_bar = np.mean(y, axis=2, keepdims=True)
    return np.sum(np.sqrt(np.sum((y - y_bar) ** 2, axis=1)), axis=1) / y.shape[2]
