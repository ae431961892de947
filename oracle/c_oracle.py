"""ctypes binding of oracle/cp_nntv.c (TEST INFRASTRUCTURE ONLY, PARITY UNPINNED -- see oracle/__init__.py).

The C + OpenMP restatement of the NN-TV Chambolle-Pock iteration: second, independently written form of the path
(matrix-free loops instead of assembled sparse matrices) and the multi-threaded CPU baseline of bench.py."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "cp_nntv.c")
LIB = os.path.join(HERE, "_build", "libcp_nntv.so")


def build(force=False):
    """gcc -O3 -fopenmp (no -march=native: the library is built in one container and run on another host)"""
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call(["gcc", "-O3", "-fopenmp", "-fPIC", "-shared", "-std=c11", "-o", LIB, SRC, "-lm"])
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
        dp = C.POINTER(C.c_double)
        _lib.nntv_chambolle_pock.argtypes = [C.c_int] * 4 + [C.c_double] * 7 + [C.c_int, C.c_double] + [dp] * 6
        _lib.nntv_chambolle_pock.restype = C.c_int
        _lib.nncurv_chambolle_pock.argtypes = _lib.nntv_chambolle_pock.argtypes
        _lib.nncurv_chambolle_pock.restype = C.c_int
        _lib.nntv_threads.restype = C.c_int
    return _lib


def threads():
    return int(lib().nntv_threads())


def nntv_chambolle_pock(m, n, k, bc, h, mu, nu, theta, tau, sigma, max_iter, tol, g, b, x0, y0):
    """g: per image (gx; gy) stacked, length 2*k*m*n; b: k*m*n; x0: 3*k*m*n; y0: 6*k*m*n (reference layouts).
    Returns (x, y, primal_history, dual_history) like PDO/chambolle_pock.m; tol=None disables the stop rule."""
    if bc not in ("neumann", "linear"):
        raise ValueError("Unknown boundary condition!")
    dp = C.POINTER(C.c_double)
    g = np.ascontiguousarray(g, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.array(x0, dtype=np.float64)
    y = np.array(y0, dtype=np.float64)
    kmn = k * m * n
    assert g.size == 2 * kmn and b.size == kmn and x.size == 3 * kmn and y.size == 6 * kmn
    hp = np.zeros((max_iter + 1, 7))
    hd = np.zeros((max_iter + 1, 7))
    rows = lib().nntv_chambolle_pock(m, n, k, 1 if bc == "linear" else 0, float(h[0]), float(h[1]), float(mu), float(nu),
                                     float(theta), float(tau), float(sigma), int(max_iter),
                                     -1.0 if tol is None else float(tol), g.ctypes.data_as(dp), b.ctypes.data_as(dp),
                                     x.ctypes.data_as(dp), y.ctypes.data_as(dp), hp.ctypes.data_as(dp),
                                     hd.ctypes.data_as(dp))
    if rows < 0:
        raise RuntimeError(f"nntv_chambolle_pock failed ({rows})")
    return x, y, hp[:rows], hd[:rows]


def nncurv_chambolle_pock(m, n, k, bc, h, mu, nu, theta, tau, sigma, max_iter, tol, g, b, x0, y0):
    """NN-curvature (NNE/mf_nn_curvature_registration_no_ref_ml.m): as nntv_chambolle_pock with y0 of length 4*k*m*n
    ([y1; per image (Lap ux; Lap uy); y3]) and bc in {'neumann', 'dirichlet'} (NNE/discrete_laplacian.m:25-28)."""
    if bc not in ("neumann", "dirichlet"):
        raise ValueError("Unknown boundary condition!")
    dp = C.POINTER(C.c_double)
    g = np.ascontiguousarray(g, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.array(x0, dtype=np.float64)
    y = np.array(y0, dtype=np.float64)
    kmn = k * m * n
    assert g.size == 2 * kmn and b.size == kmn and x.size == 3 * kmn and y.size == 4 * kmn
    hp = np.zeros((max_iter + 1, 7))
    hd = np.zeros((max_iter + 1, 7))
    rows = lib().nncurv_chambolle_pock(m, n, k, 0 if bc == "neumann" else 2, float(h[0]), float(h[1]), float(mu), float(nu),
                                       float(theta), float(tau), float(sigma), int(max_iter),
                                       -1.0 if tol is None else float(tol), g.ctypes.data_as(dp), b.ctypes.data_as(dp),
                                       x.ctypes.data_as(dp), y.ctypes.data_as(dp), hp.ctypes.data_as(dp),
                                       hd.ctypes.data_as(dp))
    if rows < 0:
        raise RuntimeError(f"nncurv_chambolle_pock failed ({rows})")
    return x, y, hp[:rows], hd[:rows]
