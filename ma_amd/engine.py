"""ctypes binding of libcpengine.so (include/cpengine.h).

The library is the product; there is no CPU fallback.  If the shared object is missing the
import of this module fails loudly (build it with `python -c "import __graft_entry__ as g; g.build()"`).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CP_ENGINE_LIB", os.path.join(_HERE, "libcpengine.so"))   # override: experiments only

MODEL_TV_L1_DENOISE, MODEL_TV_L1_REG, MODEL_TV_L2_REG, MODEL_MF_NN_TV, MODEL_MF_NN_CURV = 1, 2, 3, 4, 5
MODEL_VAR_TV, MODEL_VAR_CURV = 6, 7
FLAG_VAR_SSD_SWAP = 1
FLAG_FP32 = 2
FLAG_MFNN_NORM21 = 4
BC = {"neumann": 0, "linear": 1, "dirichlet": 2}
STOP_MAXITER, STOP_TOLERANCE = 0, 1


class ProblemDesc(C.Structure):
    _fields_ = [
        ("model", C.c_int32), ("m", C.c_int32), ("n", C.c_int32), ("k", C.c_int32),
        ("bc", C.c_int32), ("device", C.c_int32), ("batch", C.c_int32), ("flags", C.c_int32),
        ("h", C.c_double * 2), ("omega", C.c_double * 4),
        ("lambda_", C.c_double), ("mu", C.c_double),
        ("n_global", C.c_int32), ("j_offset", C.c_int32),
    ]


class Result(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("stop_reason", C.c_int32), ("rows", C.c_int32), ("cols", C.c_int32)]


class CpEngineError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP engine is not built. Run __graft_entry__.build() "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    dp = C.POINTER(C.c_double)
    vp = C.c_void_p
    lib.cp_last_error.restype = C.c_char_p
    lib.cp_version.restype = C.c_char_p
    sigs = {
        "cp_create": [C.POINTER(ProblemDesc), C.POINTER(vp)],
        "cp_destroy": [vp],
        "cp_set_images": [vp, dp, dp],
        "cp_set_primal": [vp, dp],
        "cp_set_dual": [vp, dp],
        "cp_linearize": [vp, dp],
        "cp_get_linearization": [vp, dp, dp, dp],
        "cp_set_linearization": [vp, dp, dp, dp],
        "cp_apply_A": [vp, dp, dp],
        "cp_apply_At": [vp, dp, dp],
        "cp_set_nu": [vp, C.c_double],
        "cp_singular_values": [vp, C.c_int, dp],
        "cp_norm_estimate": [vp, dp, dp, C.POINTER(C.c_int32)],
        "cp_solve": [vp, C.c_double, C.c_double, C.c_double, C.c_int32, C.c_double, C.POINTER(Result)],
        "cp_get_result": [vp, C.c_int32, C.POINTER(Result)],
        "cp_get_history": [vp, dp, dp],
        "cp_get_primal": [vp, dp],
        "cp_get_dual": [vp, dp],
        "cp_get_images": [vp, dp],
        "cp_prolong": [vp, vp, C.c_double],
        "cp_downsample": [vp, vp],
        "cp_iterate": [vp, C.c_double, C.c_double, C.c_double, C.c_int32],
        "cp_iterate_begin": [vp, C.c_double, C.c_double, C.c_double, C.c_int32],
        "cp_iterate_steps": [vp, C.c_int32],
        "cp_iterate_end": [vp],
        "cp_sync": [vp],
        "cp_kernel_time_ms": [vp, C.c_char_p, dp, C.POINTER(C.c_int64)],
        "cp_set_stream": [vp, vp],
        "cp_dominant_kernel": [vp, C.c_char_p, C.c_int32],
        "cp_handover_fallbacks": [vp, C.POINTER(C.c_int64)],
        "cp_mg_halo_pack": [vp, C.c_int, C.c_int, C.c_int, vp],
        "cp_mg_halo_unpack": [vp, C.c_int, C.c_int, C.c_int, vp],
        "cp_mg_state_pack": [vp, C.c_int, C.c_int, vp, vp],
        "cp_mg_state_unpack": [vp, C.c_int, C.c_int, vp, vp],
        "cp_mg_begin": [vp, C.c_double, C.c_double, C.c_double, C.c_int32],
        "cp_mg_stage": [vp, C.c_int, C.c_int32, vp, C.c_int32, C.c_double, C.c_int32, C.c_double],
        "cp_mg_end": [vp, C.c_int32],
        "cp_mg_stop_state": [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)],
        "cp_mg_gram": [vp, C.c_int, dp, dp],
        "cp_mg_power_start": [vp, dp],
        "cp_singular_values_static": [C.c_int32, vp, C.c_int32, C.c_int32, C.c_int32, vp, vp],
        "cp_stream_copy_gbps": [C.c_int32, C.c_int64, C.c_int32, C.c_int32, dp],
        "cp_evaluate_displacement": [C.c_int32, dp, C.c_int32, C.c_int32, dp, dp, dp, C.c_int32, C.c_int32, dp, dp],
        "cp_landmark_transform": [C.c_int32, dp, C.c_int32, dp, C.c_int32, C.c_int32, dp, dp, C.POINTER(C.c_int32)],
        "cp_landmark_accuracy": [C.c_int32, dp, C.c_int32, C.c_int32, dp],
        "cp_nuclear_norm": [C.c_int32, dp, C.c_int64, C.c_int32, C.c_double, C.c_double, C.c_int32, dp, dp, dp],
        "cp_nn_measure": [C.c_int32, C.c_int32, dp, C.c_int64, C.c_int32, C.c_double, dp, dp, C.c_double, C.c_double,
                          C.c_double, C.c_int32, C.c_double, dp, dp, dp, dp, C.POINTER(Result)],
    }
    for name, args in sigs.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    lib.cp_mg_partial_len.argtypes = []
    lib.cp_mg_partial_len.restype = C.c_int64
    lib.cp_mg_halo_len.argtypes = [vp, C.c_int]
    lib.cp_mg_halo_len.restype = C.c_int64
    return lib


def _torch_first():
    """The PyTorch-ROCm wheel bundles its own HIP runtime next to the system one this library links. The two coexist in a
    process only if torch initialises its device BEFORE libcpengine.so is loaded (loaded the other way round, whichever
    runtime comes second reports that there is no GPU). So: if torch is importable and sees a GPU, let it come first.
    CP_ENGINE_NO_TORCH=1 skips this (a process that never touches torch.cuda does not need it)."""
    if os.environ.get("CP_ENGINE_NO_TORCH", "0") == "1":
        return
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass


_torch_first()
lib = _load()
EXPORTED = ["cp_last_error", "cp_version", "cp_create", "cp_destroy", "cp_set_images", "cp_set_primal", "cp_set_dual",
            "cp_linearize", "cp_get_linearization", "cp_set_linearization", "cp_apply_A", "cp_apply_At", "cp_set_nu", "cp_singular_values", "cp_norm_estimate", "cp_solve",
            "cp_get_result", "cp_get_history", "cp_get_primal", "cp_get_dual", "cp_get_images", "cp_prolong",
            "cp_downsample", "cp_iterate", "cp_iterate_begin", "cp_iterate_steps", "cp_iterate_end", "cp_sync",
            "cp_kernel_time_ms", "cp_set_stream", "cp_dominant_kernel", "cp_handover_fallbacks", "cp_mg_partial_len",
            "cp_mg_halo_len", "cp_mg_halo_pack", "cp_mg_halo_unpack", "cp_mg_state_pack", "cp_mg_state_unpack", "cp_mg_begin", "cp_mg_stage", "cp_mg_end",
            "cp_mg_stop_state", "cp_mg_gram", "cp_mg_power_start", "cp_singular_values_static", "cp_stream_copy_gbps", "cp_evaluate_displacement",
            "cp_nuclear_norm", "cp_nn_measure", "cp_landmark_transform", "cp_landmark_accuracy"]

BUF_X0, BUF_X1, BUF_P0, BUF_P1, BUF_XTMP = 0, 1, 2, 3, 4
SEL_ALL, SEL_X_U, SEL_P_Y12, SEL_P_DY, SEL_P_Y2 = 0, 1, 2, 3, 4
SEL_VEC_X_U, SEL_VEC_P_DY, SEL_VEC_P_Y2 = 5, 6, 7
STAGE_DUAL0, STAGE_EIG0, STAGE_FUSED, STAGE_EFUSED, STAGE_APPLY_A, STAGE_APPLY_AT = 0, 1, 2, 3, 4, 5
STAGE_SPLIT_DUAL, STAGE_SPLIT_EIG, STAGE_SPLIT_PRIMAL, STAGE_SPLIT_FIN = 6, 7, 8, 9
STAGE_ROW0_DUAL, STAGE_ROW0_EIG, STAGE_ROW0_PRIMAL, STAGE_ROW0_FIN = 10, 11, 12, 13
STAGE_CURV_P, STAGE_CURV_Y2, STAGE_CURV_FIN = 14, 15, 16


def _chk(rc):
    if rc != 0:
        raise CpEngineError(lib.cp_last_error().decode())


def _dptr(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _f64(a):
    """host vector in MATLAB (column-major) order as a contiguous float64 array"""
    return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))


class Engine:
    """One handle = one (F, G, K) triple of the reference at one resolution."""

    def __init__(self, model, m, n, k=1, bc="neumann", h=(1.0, 1.0), omega=None, lam=0.0, mu=0.0, device=0,
                 batch=1, n_global=0, j_offset=0, flags=0):
        d = ProblemDesc()
        d.model, d.m, d.n, d.k = model, m, n, k
        if bc not in BC:
            raise CpEngineError("cpengine:badBC Unknown boundary condition!")
        d.bc, d.device, d.batch = BC[bc], device, batch
        d.h[0], d.h[1] = float(h[0]), float(h[1])
        if omega is not None:
            for i in range(4):
                d.omega[i] = float(omega[i])
        d.lambda_, d.mu = float(lam), float(mu)
        d.n_global, d.j_offset = n_global, j_offset
        d.flags = flags
        self.desc = d
        self.model, self.m, self.n, self.k, self.batch = model, m, n, k, batch
        self.var = model in (MODEL_VAR_TV, MODEL_VAR_CURV)
        self.mf = model in (MODEL_MF_NN_TV, MODEL_MF_NN_CURV) or self.var
        if self.var:
            self.nx = 2 * k * m * n
            self.ny = (5 if model == MODEL_VAR_TV else 3) * k * m * n
            self.cols = 6
        elif self.mf:
            self.nx = 3 * k * m * n
            self.ny = (6 if model == MODEL_MF_NN_TV else 4) * k * m * n
            self.cols = 7
        else:
            nc = 1 if model == MODEL_TV_L1_DENOISE else 2
            self.nx = nc * m * n * batch
            self.ny = 2 * nc * m * n * batch
            self.cols = 5
        self._h = C.c_void_p()
        _chk(lib.cp_create(C.byref(d), C.byref(self._h)))

    def close(self):
        if self._h:
            lib.cp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # data in
    def set_images(self, a, b=None):
        a, b = _f64(a), _f64(b)
        _chk(lib.cp_set_images(self._h, _dptr(a), _dptr(b)))

    def set_primal(self, x0=None):
        x0 = _f64(x0)
        _chk(lib.cp_set_primal(self._h, _dptr(x0)))

    def set_dual(self, y0=None):
        y0 = _f64(y0)
        _chk(lib.cp_set_dual(self._h, _dptr(y0)))

    def linearize(self, u0=None):
        u0 = _f64(u0)
        _chk(lib.cp_linearize(self._h, _dptr(u0)))

    def get_linearization(self):
        kk = self.k * self.batch
        T = np.empty(kk * self.m * self.n)
        g = np.empty(2 * kk * self.m * self.n)
        b = np.empty(kk * self.m * self.n)
        _chk(lib.cp_get_linearization(self._h, _dptr(T), _dptr(g), _dptr(b)))
        return T, g, b

    def set_linearization(self, T_u, grad, b):
        """grad T and b given directly (the data arguments of the operators); T_u may be None"""
        T_u, grad, b = _f64(T_u), _f64(grad), _f64(b)
        _chk(lib.cp_set_linearization(self._h, _dptr(T_u), _dptr(grad), _dptr(b)))

    def apply_A(self, x):
        x = _f64(x)
        if x.size != self.nx:
            raise CpEngineError(f"cpengine:badArg x has {x.size} entries, expected {self.nx}")
        y = np.empty(self.ny)
        _chk(lib.cp_apply_A(self._h, _dptr(x), _dptr(y)))
        return y

    def apply_At(self, y):
        y = _f64(y)
        if y.size != self.ny:
            raise CpEngineError(f"cpengine:badArg y has {y.size} entries, expected {self.ny}")
        x = np.empty(self.nx)
        _chk(lib.cp_apply_At(self._h, _dptr(y), _dptr(x)))
        return x

    def set_nu(self, nu):
        _chk(lib.cp_set_nu(self._h, float(nu)))

    def singular_values(self, which):
        sv = np.empty(self.k)
        _chk(lib.cp_singular_values(self._h, int(which), _dptr(sv)))
        return sv

    def norm_estimate(self, x0=None):
        x0 = _f64(x0)
        est = C.c_double()
        it = C.c_int32()
        _chk(lib.cp_norm_estimate(self._h, _dptr(x0), C.byref(est), C.byref(it)))
        return est.value, it.value

    def solve(self, theta, tau, sigma, max_iter, tol):
        r = Result()
        _chk(lib.cp_solve(self._h, theta, tau, sigma, int(max_iter), tol, C.byref(r)))
        return r

    def result(self, problem=0):
        r = Result()
        _chk(lib.cp_get_result(self._h, problem, C.byref(r)))
        return r

    def history(self):
        """(primal_history, dual_history); for a batch: lists with one pair of arrays per problem"""
        res = [self.result(p) for p in range(self.batch)]
        tot = sum(r.rows * r.cols for r in res)
        ph = np.empty(tot)
        dh = np.empty(tot)
        _chk(lib.cp_get_history(self._h, _dptr(ph), _dptr(dh)))
        outp, outd, off = [], [], 0
        for r in res:
            sz = r.rows * r.cols
            outp.append(ph[off:off + sz].reshape((r.rows, r.cols), order="F").copy())
            outd.append(dh[off:off + sz].reshape((r.rows, r.cols), order="F").copy())
            off += sz
        if self.batch == 1:
            return outp[0], outd[0]
        return outp, outd

    def primal(self):
        x = np.empty(self.nx)
        _chk(lib.cp_get_primal(self._h, _dptr(x)))
        return x

    def dual(self):
        y = np.empty(self.ny)
        _chk(lib.cp_get_dual(self._h, _dptr(y)))
        return y

    def images(self):
        n_g = self.desc.n_global if self.desc.n_global > 0 else self.n
        a = np.empty(self.k * self.m * n_g * self.batch)
        _chk(lib.cp_get_images(self._h, _dptr(a)))
        return a

    def prolong_from(self, coarse, factor=1.0):
        _chk(lib.cp_prolong(coarse._h, self._h, float(factor)))

    def downsample_to(self, coarse):
        _chk(lib.cp_downsample(self._h, coarse._h))

    # row-slab multi-GPU hooks (device pointers are plain ints, e.g. torch.Tensor.data_ptr())
    def set_stream(self, stream_ptr):
        _chk(lib.cp_set_stream(self._h, C.c_void_p(stream_ptr)))

    def halo_len(self, sel):
        return int(lib.cp_mg_halo_len(self._h, sel))

    def halo_pack(self, buf, sel, column, dev_ptr):
        _chk(lib.cp_mg_halo_pack(self._h, buf, sel, column, C.c_void_p(dev_ptr)))

    def halo_unpack(self, buf, sel, side, dev_ptr):
        _chk(lib.cp_mg_halo_unpack(self._h, buf, sel, side, C.c_void_p(dev_ptr)))

    def state_pack(self, xbuf, pbuf, to_left_ptr, to_right_ptr):
        """device pointers (0 = skip that side)"""
        _chk(lib.cp_mg_state_pack(self._h, xbuf, pbuf, C.c_void_p(to_left_ptr or None), C.c_void_p(to_right_ptr or None)))

    def state_unpack(self, xbuf, pbuf, from_right_ptr, from_left_ptr):
        _chk(lib.cp_mg_state_unpack(self._h, xbuf, pbuf, C.c_void_p(from_right_ptr or None), C.c_void_p(from_left_ptr or None)))

    def mg_begin(self, theta, tau, sigma, max_iters=4094):
        _chk(lib.cp_mg_begin(self._h, theta, tau, sigma, int(max_iters)))

    def mg_stage(self, stage, n, dev_ptr, world=1, scale=1.0, check_stop=0, tol=0.0):
        _chk(lib.cp_mg_stage(self._h, stage, n, C.c_void_p(dev_ptr), world, scale, check_stop, tol))

    def mg_end(self, iterations):
        _chk(lib.cp_mg_end(self._h, iterations))

    def mg_stop_state(self):
        a, b = C.c_int32(), C.c_int32()
        _chk(lib.cp_mg_stop_state(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def mg_gram(self, which, V=None):
        V = None if V is None else np.ascontiguousarray(V, dtype=np.float64)
        g = np.empty(256)
        _chk(lib.cp_mg_gram(self._h, which, _dptr(V), _dptr(g)))
        return g.reshape(16, 16)

    def mg_power_start(self):
        s = C.c_double()
        _chk(lib.cp_mg_power_start(self._h, C.byref(s)))
        return s.value

    # benchmarking hooks
    def iterate(self, theta, tau, sigma, iters):
        _chk(lib.cp_iterate(self._h, theta, tau, sigma, int(iters)))

    def iterate_begin(self, theta, tau, sigma, max_iters):
        _chk(lib.cp_iterate_begin(self._h, theta, tau, sigma, int(max_iters)))

    def iterate_steps(self, iters):
        _chk(lib.cp_iterate_steps(self._h, int(iters)))

    def iterate_end(self):
        _chk(lib.cp_iterate_end(self._h))

    def sync(self):
        _chk(lib.cp_sync(self._h))

    def kernel_time_ms(self, name):
        ms = C.c_double()
        n = C.c_int64()
        _chk(lib.cp_kernel_time_ms(self._h, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def dominant_kernel(self):
        """name of the kernel the steady-state iteration spends its time in, as rocprofv3 prints it (cp_dominant_kernel)"""
        buf = C.create_string_buffer(256)
        _chk(lib.cp_dominant_kernel(self._h, buf, 256))
        return buf.value.decode()

    def handover_fallbacks(self):
        """hand-over time-outs of the 8-wave kernels since the handle was created (cp_handover_fallbacks); must be 0"""
        n = C.c_int64()
        _chk(lib.cp_handover_fallbacks(self._h, C.byref(n)))
        return n.value


def singular_values_static(device, field_ptr, m, n, k, sv_ptr, stream_ptr=0):
    """cp_singular_values_static: device pointers as plain ints (e.g. torch.Tensor.data_ptr()); asynchronous on the stream"""
    _chk(lib.cp_singular_values_static(int(device), C.c_void_p(field_ptr), int(m), int(n), int(k), C.c_void_p(sv_ptr),
                                       C.c_void_p(stream_ptr or None)))


def stream_copy_gbps(device=0, nbytes=1 << 31, lane_bytes=16, repeats=3):
    """cp_stream_copy_gbps: measured device-to-device copy rate (GB/s, read + write) of the library's own streaming kernel"""
    out = C.c_double()
    _chk(lib.cp_stream_copy_gbps(int(device), int(nbytes), int(lane_bytes), int(repeats), C.byref(out)))
    return out.value


def evaluate_displacement(img, h, u=None, omega=None, s_grid=None, device=0, want_grad=True):
    """[img_u, g] = evaluate_displacement(img, h, u, omega, s_grid)  (IRE/evaluate_displacement.m): img_u is k x l, g is
    (k*l) x 2 (the reference wraps g into a sparse matrix, :47-48)"""
    img = np.asarray(img, dtype=np.float64)
    m, n = img.shape
    k, l = (m, n) if s_grid is None else (int(s_grid[0]), int(s_grid[1]))
    hh = np.ascontiguousarray(np.asarray(h, dtype=np.float64).reshape(2))
    imgv = np.ascontiguousarray(img.reshape(-1, order="F"))
    uv = None
    if u is not None and np.size(u) > 1:
        uv = np.ascontiguousarray(np.asarray(u, dtype=np.float64).reshape((k * l, 2), order="F").reshape(-1, order="F"))
    elif u is not None and float(np.asarray(u).reshape(-1)[0]) != 0.0:
        uv = np.full(2 * k * l, float(np.asarray(u).reshape(-1)[0]))
    om = None if omega is None else np.ascontiguousarray(np.asarray(omega, dtype=np.float64).reshape(4))
    out = np.empty(k * l)
    g = np.empty(2 * k * l) if want_grad else None
    _chk(lib.cp_evaluate_displacement(int(device), _dptr(imgv), m, n, _dptr(hh), _dptr(uv), _dptr(om), k, l, _dptr(out), _dptr(g)))
    return out.reshape((k, l), order="F"), (None if g is None else g.reshape((k * l, 2), order="F"))


def landmark_transform(LM, u, omega, device=0):
    """LM_transformed = landmark_transform(LM, u, omega)  (NNE/landmark_transform.m) on the device; also returns the flags
    of the landmarks whose fixed-point iteration failed (the reference's warning, :55-57)"""
    LM = np.asarray(LM, dtype=np.float64).reshape(-1, 2)
    u = np.asarray(u, dtype=np.float64)
    m, n = u.shape[0], u.shape[1]
    lmv = np.ascontiguousarray(LM.reshape(-1, order="F"))
    uv = np.ascontiguousarray(u.reshape(-1, order="F"))
    om = np.ascontiguousarray(np.asarray(omega, dtype=np.float64).reshape(4))
    out = np.zeros(2 * LM.shape[0])
    failed = np.zeros(max(LM.shape[0], 1), dtype=np.int32)
    _chk(lib.cp_landmark_transform(int(device), _dptr(lmv), LM.shape[0], _dptr(uv), m, n, _dptr(om), _dptr(out),
                                   failed.ctypes.data_as(C.POINTER(C.c_int32))))
    return out.reshape((LM.shape[0], 2), order="F"), failed[: LM.shape[0]].astype(bool)


def landmark_accuracy(LM, device=0):
    """LM_acc = landmark_accuracy(LM)  (NNE/landmark_accuracy.m): LM is a list of k (numLM, 2) arrays"""
    y = np.stack([np.asarray(a, dtype=np.float64).reshape(-1, 2) for a in LM], axis=2)        # numLM x 2 x k
    acc = np.zeros(y.shape[0])
    _chk(lib.cp_landmark_accuracy(int(device), _dptr(np.ascontiguousarray(y.reshape(-1, order="F"))), y.shape[0], y.shape[2],
                                  _dptr(acc)))
    return acc


def nuclear_norm(L, numImg, tau, mu, conjugate_flag=False, nargout=2, device=0):
    """[res1, res2, res3] = nuclear_norm(L, numImg, tau, mu, conjugate_flag)  (NNE/nuclear_norm.m:1-2) on the device"""
    Lv = np.ascontiguousarray(np.asarray(L, dtype=np.float64).reshape(-1, order="F"))
    npix = Lv.size // int(numImg)
    r1, r2 = C.c_double(), C.c_double()
    prox = np.empty_like(Lv) if nargout == 3 else None
    _chk(lib.cp_nuclear_norm(int(device), _dptr(Lv), npix, int(numImg), float(tau), float(mu), 1 if conjugate_flag else 0,
                             C.byref(r1), C.byref(r2), _dptr(prox)))
    return r1.value, r2.value, prox


def nn_measure(version, I, numImg, param, theta, tau, sigma, maxIter, tol, L0=None, P0=None, device=0):
    """[x_star, y_star, primal_history, dual_history] of chambolle_pock(F, G, K, L0, P0, theta, tau, sigma, maxIter, tol) for
    the L-only problems of NNE/test_nuclear_norm_measure.m:212-242 (version 1 soft / 2 hard / 3 mean-free hard; param =
    mu or nu); also returns the Result (iterations, stop_reason)."""
    Iv = np.ascontiguousarray(np.asarray(I, dtype=np.float64).reshape(-1, order="F"))
    npix = Iv.size // int(numImg)
    vec = lambda a: None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1, order="F"))
    L0v, P0v = vec(L0), vec(P0)
    xs, ys = np.empty_like(Iv), np.empty_like(Iv)
    ph, dh = np.empty((maxIter + 1) * 5), np.empty((maxIter + 1) * 5)
    res = Result()
    _chk(lib.cp_nn_measure(int(device), int(version), _dptr(Iv), npix, int(numImg), float(param), _dptr(L0v), _dptr(P0v),
                           float(theta), float(tau), float(sigma), int(maxIter), float(tol), _dptr(xs), _dptr(ys),
                           _dptr(ph), _dptr(dh), C.byref(res)))
    rows = res.rows
    return xs, ys, ph[: rows * 5].reshape((rows, 5), order="F"), dh[: rows * 5].reshape((rows, 5), order="F"), res
