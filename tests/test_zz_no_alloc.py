"""cp_singular_values_static (the allocation-free entry point of include/cpengine.h) against cp_singular_values on
the same field: two calls in a row, the second with a smaller input (the static scratch keeps the first call's rows)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _handle_sv(E, imgs, m, n, k):
    eng = E.Engine(E.MODEL_MF_NN_TV, m, n, k=k, bc="neumann", h=(1.0, 1.0), mu=0.2)
    eng.set_images(imgs)
    eng.set_primal(None)
    eng.linearize()                      # u0 = 0, h = 1: T(u0) is the image itself (IRE/test_interpolation.m:39-42)
    T, _, _ = eng.get_linearization()
    sv = eng.singular_values(0)
    eng.close()
    return T, sv


def test_static_entry_point_matches_handle_on_two_calls_in_a_row():
    import torch
    from ma_amd import engine as E, synth
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream(dev).cuda_stream
    for (m, n, k) in ((1100, 900, 7), (96, 80, 5), (40, 36, 16)):      # 3868 > 768 workgroups, then smaller inputs
        fr = synth.dynamic_test_image(m, n, k).numpy()
        imgs = np.concatenate([fr[i].reshape(-1, order="F") for i in range(k)])
        T, sv_handle = _handle_sv(E, imgs, m, n, k)
        assert np.array_equal(T, imgs)
        field = torch.from_numpy(imgs).to(dev)
        out = torch.full((16,), -7.0, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        E.singular_values_static(0, field.data_ptr(), m, n, k, out.data_ptr(), stream)
        torch.cuda.synchronize()
        sv = out.cpu().numpy()
        assert np.all(sv[k:] == -7.0)                                   # writes k values, nothing else
        np.testing.assert_allclose(sv[:k], sv_handle, rtol=1e-12, atol=1e-13 * sv_handle[0])
        assert np.all(np.diff(sv[:k]) <= 0)
        # and against LAPACK on the mean-free image matrix (NNE/mf_nn_tv_registration_no_ref_ml.m:173-175)
        Y = imgs.reshape((m * n, k), order="F")
        s_ref = np.linalg.svd(Y - Y.mean(axis=1, keepdims=True), compute_uv=False)
        np.testing.assert_allclose(sv[:k - 1], s_ref[:k - 1], rtol=1e-10, atol=1e-10 * s_ref[0])


def test_static_entry_point_rejects_what_exceeds_its_scratch():
    import torch
    from ma_amd import engine as E
    x = torch.zeros(64, dtype=torch.float64, device="cuda:0")
    with pytest.raises(E.CpEngineError, match="unsupported"):
        E.singular_values_static(0, x.data_ptr(), 2, 2, 17, x.data_ptr())
    with pytest.raises(E.CpEngineError, match="unsupported"):
        E.singular_values_static(0, x.data_ptr(), 40000, 40000, 2, x.data_ptr())
    with pytest.raises(E.CpEngineError, match="badArg"):
        E.singular_values_static(0, 0, 8, 8, 2, x.data_ptr())
