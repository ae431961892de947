"""mex/cp_mex.cpp EXECUTED: compiled unchanged against tests/mex_shim/mex_shim.cpp (a functional stand-in for the MEX
runtime -- test infrastructure, see its header; the image has no MATLAB / Octave) and linked with the real libcpengine.so.
Every call below goes through the gateway's own mexFunction(nlhs, plhs, nrhs, prhs): argument marshalling, output
allocation, handle table, error path (mexErrMsgIdAndTxt = longjmp out of the gateway), mexLock / mexAtExit bookkeeping.
The results must equal the ctypes path (ma_amd/engine.py) bit for bit: both end in the same C ABI calls.
Reference contract: PDO/chambolle_pock.m:1-2 (solver outputs), SURVEY.md section 8(b) "MEX gateway"."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tests", "mex_shim", "_build", "libcp_mex_test.so")
DOUBLE, INT32, UINT64, CHAR = 6, 12, 15, -1


class Mex:
    """thin driver of the shim's mxt_* API: m.call('cmd', args..., nlhs=k) -> list of numpy arrays / strings"""

    def __init__(self):
        from ma_amd import engine as E   # noqa: F401  loads libcpengine.so first (same file the gateway links)
        if not os.path.exists(LIB):
            raise RuntimeError(f"{LIB} missing: run __graft_entry__.build()")
        L = C.CDLL(LIB)
        vp = C.c_void_p
        L.mxt_double.restype = vp
        L.mxt_double.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_double)]
        L.mxt_string.restype = vp
        L.mxt_string.argtypes = [C.c_char_p]
        L.mxt_uint64.restype = vp
        L.mxt_uint64.argtypes = [C.c_uint64]
        L.mxt_free.argtypes = [vp]
        L.mxt_numel.restype = C.c_size_t
        L.mxt_numel.argtypes = [vp]
        L.mxt_dim.restype = C.c_size_t
        L.mxt_dim.argtypes = [vp, C.c_int]
        L.mxt_class.argtypes = [vp]
        L.mxt_data.restype = vp
        L.mxt_data.argtypes = [vp]
        L.mxt_error_id.restype = C.c_char_p
        L.mxt_error_msg.restype = C.c_char_p
        L.mxt_live_arrays.restype = C.c_long
        L.mxt_call.argtypes = [C.c_int, C.POINTER(vp), C.c_int, C.POINTER(vp)]
        self.L = L

    def _to_mx(self, a):
        L = self.L
        if isinstance(a, str):
            return L.mxt_string(a.encode())
        if isinstance(a, Handle):
            return L.mxt_uint64(a.value)
        if a is None:
            return L.mxt_double(0, 0, 1, None)
        arr = np.asarray(a, dtype=np.float64)
        if arr.ndim == 0:
            arr = arr.reshape(1, 1)
        if arr.ndim == 1:
            arr = arr.reshape(-1, 1)
        f = np.asfortranarray(arr)
        dims = list(f.shape) + [1] * (3 - f.ndim)
        return L.mxt_double(dims[0], dims[1], dims[2], f.ravel(order="F").ctypes.data_as(C.POINTER(C.c_double)))

    def _from_mx(self, p):
        L = self.L
        cls, n = L.mxt_class(p), L.mxt_numel(p)
        shape = (L.mxt_dim(p, 0), L.mxt_dim(p, 1))
        if cls == CHAR:
            return C.string_at(L.mxt_data(p), n).decode()
        ct = {DOUBLE: C.c_double, INT32: C.c_int32, UINT64: C.c_uint64}[cls]
        flat = np.ctypeslib.as_array(C.cast(L.mxt_data(p), C.POINTER(ct)), shape=(n,)).copy() if n else np.zeros(0)
        if cls == UINT64:
            return Handle(int(flat[0]))
        return flat.reshape(shape, order="F")

    def call(self, *args, nlhs=1):
        L = self.L
        prhs = [self._to_mx(a) for a in args]
        pr = (C.c_void_p * len(prhs))(*prhs)
        pl = (C.c_void_p * max(nlhs, 1))()
        rc = L.mxt_call(nlhs, pl, len(prhs), pr)
        try:
            if rc != 0:
                raise MexError(L.mxt_error_id().decode(), L.mxt_error_msg().decode())
            return [self._from_mx(pl[i]) if pl[i] else None for i in range(max(nlhs, 1))]
        finally:
            for p in prhs:
                L.mxt_free(p)
            for i in range(max(nlhs, 1)):
                if pl[i]:
                    L.mxt_free(pl[i])


class Handle:
    def __init__(self, value):
        self.value = value


class MexError(RuntimeError):
    def __init__(self, ident, msg):
        super().__init__(f"{ident}: {msg}")
        self.ident, self.msg = ident, msg


@pytest.fixture(scope="module")
def mex():
    return Mex()


def test_version_and_unknown_command(mex):
    from ma_amd import engine as E
    assert mex.call("version")[0] == E.lib.cp_version().decode()
    with pytest.raises(MexError) as ei:
        mex.call("no_such_command", Handle(0))
    assert ei.value.ident == "cpengine:badArg"        # a foreign handle is rejected before the command is looked at
    with pytest.raises(MexError) as ei:
        mex.call(3.0)
    assert ei.value.ident == "cpengine:badArg" and "cp_mex('command'" in ei.value.msg


def test_error_path_carries_the_library_message_id(mex):
    """NNE/finite_difference_operator.m:35 error('Unknown boundary condition!') -> CP_ERR_BAD_BC -> cpengine:badBC through
    mexErrMsgIdAndTxt; the handle table stays empty and unlocked"""
    locks = mex.L.mxt_lock_count()
    with pytest.raises(MexError) as ei:   # NN-TV knows neumann (0) and linear (1) rows only
        mex.call("create", 4, 32, 24, 3, 2, [1.0, 1.0], None, 0.0, 0.2)
    assert ei.value.ident == "cpengine:badBC" and "Unknown boundary condition!" in ei.value.msg
    with pytest.raises(MexError) as ei:
        mex.call("create", 4, 32, 24, 3, 0, [1.0], None, 0.0, 0.2)      # h_grid with one element
    assert ei.value.ident == "cpengine:badArg" and "h_grid has 1 elements, expected 2" in ei.value.msg
    assert mex.L.mxt_lock_count() == locks


@pytest.mark.parametrize("model,m,n,k,bc,mu", [("tv", 48, 40, 3, "neumann", 0.2), ("curvature", 40, 36, 4, "dirichlet", 20.0)])
def test_mf_nn_solve_through_mexfunction_equals_ctypes_bitwise(mex, model, m, n, k, bc, mu):
    from ma_amd import engine as E, synth
    mid = E.MODEL_MF_NN_TV if model == "tv" else E.MODEL_MF_NN_CURV
    imgs = synth.dynamic_test_image(m, n, k).numpy()
    imgv = np.concatenate([imgs[i].reshape(-1, order="F") for i in range(k)])
    u0 = synth.smooth_displacement(m, n, k, amp=1.5).numpy().reshape(-1)
    x0 = np.concatenate([u0, np.zeros(k * m * n)])
    omega = [0.0, float(m), 0.0, float(n)]
    nu, tau, iters, tol = 2.0 * k, 0.3 if model == "tv" else 0.12, 25, 1e-3

    # ---- ctypes path
    eng = E.Engine(mid, m, n, k=k, bc=bc, h=(1.0, 1.0), omega=omega, mu=mu)
    eng.set_images(imgv)
    eng.set_primal(x0)
    eng.linearize()
    sv_c = eng.singular_values(0)
    est_c, np_c = eng.norm_estimate()
    eng.set_nu(nu)
    res = eng.solve(1.0, tau, tau, iters, tol)
    x_c, y_c = eng.primal(), eng.dual()
    ph_c, dh_c = eng.history()
    name_c = eng.dominant_kernel()
    eng.close()

    # ---- the same through mexFunction
    locks = mex.L.mxt_lock_count()
    h, = mex.call("create", mid, m, n, k, E.BC[bc], [1.0, 1.0], omega, 0.0, mu)
    assert isinstance(h, Handle) and mex.L.mxt_lock_count() == locks + 1         # mexLock at the first live handle
    mex.call("set_images", h, imgv, nlhs=0)
    mex.call("set_primal", h, x0, nlhs=0)
    mex.call("linearize", h, nlhs=0)
    sv_m, = mex.call("singular_values", h, 0)
    est_m, np_m = mex.call("norm_estimate", h, nlhs=2)
    mex.call("set_nu", h, nu, nlhs=0)
    it_m, why_m = mex.call("solve", h, 1.0, tau, tau, iters, tol, nlhs=2)
    ph_m, dh_m = mex.call("get_history", h, nlhs=2)
    x_m, = mex.call("get_primal", h)
    y_m, = mex.call("get_dual", h)
    assert mex.call("dominant_kernel", h)[0] == name_c
    assert mex.call("handover_fallbacks", h)[0][0, 0] == 0.0
    with pytest.raises(MexError) as ei:                                           # wrong length -> gateway's own check
        mex.call("set_primal", h, x0[:-1], nlhs=0)
    assert ei.value.ident == "cpengine:badArg" and "x0 has" in ei.value.msg
    mex.call("destroy", h, nlhs=0)
    assert mex.L.mxt_lock_count() == locks                                       # mexUnlock with the last handle
    with pytest.raises(MexError) as ei:
        mex.call("get_primal", h)
    assert "stale or foreign handle" in ei.value.msg

    assert int(it_m[0, 0]) == res.iterations and int(why_m[0, 0]) == res.stop_reason
    assert ph_m.shape == ph_c.shape == (res.iterations + 1, 7)                   # PDO/chambolle_pock.m:224-228, mf_nn: 7 columns
    assert np.array_equal(ph_m, ph_c) and np.array_equal(dh_m, dh_c)
    assert np.array_equal(x_m.ravel(), x_c) and np.array_equal(y_m.ravel(), y_c)
    assert np.array_equal(sv_m.ravel(), sv_c) and est_m[0, 0] == est_c and int(np_m[0, 0]) == np_c


def test_pairwise_batch_history_and_atexit(mex):
    """TV-L2 batch of 3 through the gateway: per-problem histories (the gateway slices the stacked buffer), and the
    mexAtExit hook destroys what is still alive"""
    from ma_amd import engine as E, synth
    m, n, B = 40, 32, 3
    fr = synth.dynamic_test_image(m, n, 2).numpy()
    T = np.concatenate([np.roll(fr[1], s, axis=0).reshape(-1, order="F") for s in (0, 1, -2)])
    R = np.tile(fr[0].reshape(-1, order="F"), B)
    lam, tau = 50.0, 4.0
    sigma = (1 - 1e-4) / (8 * tau)
    eng = E.Engine(E.MODEL_TV_L2_REG, m, n, lam=lam, batch=B)
    eng.set_images(T, R)
    eng.linearize()
    eng.solve(1.0, tau, sigma, 20, 1e-3)
    ph_c, dh_c = eng.history()
    x_c = eng.primal()
    eng.close()
    h, = mex.call("create", E.MODEL_TV_L2_REG, m, n, 1, 0, [1.0, 1.0], None, lam, 0.0, B)
    mex.call("set_images", h, T, R, nlhs=0)
    mex.call("linearize", h, nlhs=0)
    mex.call("solve", h, 1.0, tau, sigma, 20, 1e-3, nlhs=2)
    for p in range(B):
        ph_m, dh_m = mex.call("get_history", h, p, nlhs=2)
        assert np.array_equal(ph_m, ph_c[p]) and np.array_equal(dh_m, dh_c[p])
    x_m, = mex.call("get_primal", h)
    assert np.array_equal(x_m.ravel(), x_c)
    mex.L.mxt_run_atexit()                           # `clear mex`: the hook destroys the live handle
    with pytest.raises(MexError):
        mex.call("get_primal", h)


def test_evaluate_displacement_command(mex):
    """[img_u, g] = cp_mex('evaluate_displacement', img, h, u, omega, s_grid): IRE/evaluate_displacement.m:1 on its own"""
    from ma_amd import engine as E, synth
    img = synth.dynamic_test_image(30, 22, 1).numpy()[0]
    omega = [0.0, 30.0, 0.0, 22.0]
    a, g = E.evaluate_displacement(img, (1.0, 1.0), None, omega, (45, 33))
    am, gm = mex.call("evaluate_displacement", img, [1.0, 1.0], 0.0, omega, [45.0, 33.0], nlhs=2)
    assert am.shape == (45, 33) and np.array_equal(am, a) and np.array_equal(gm, g)
