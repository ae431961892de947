"""Error behaviour at the C ABI (status codes + message ids instead of MATLAB error()/warning())."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_create_errors():
    from ma_amd import engine as E
    with pytest.raises(E.CpEngineError, match="badBC"):          # NNE/finite_difference_operator.m:35
        E.Engine(E.MODEL_MF_NN_TV, 16, 16, k=3, bc="dirichlet")
    E.Engine(E.MODEL_MF_NN_TV, 16, 16, k=17).close()            # mf_nn: 17..32 images on the general path (cp_mfnn_big.cuh)
    with pytest.raises(E.CpEngineError, match="unsupported"):
        E.Engine(E.MODEL_MF_NN_TV, 16, 16, k=33)
    with pytest.raises(E.CpEngineError, match="unsupported"):    # ... in fp64 storage, on one GPU
        E.Engine(E.MODEL_MF_NN_TV, 16, 16, k=17, flags=E.FLAG_FP32)
    with pytest.raises(E.CpEngineError, match="unsupported"):
        E.Engine(E.MODEL_MF_NN_CURV, 16, 8, k=20, bc="neumann", n_global=16, j_offset=0)
    E.Engine(E.MODEL_VAR_TV, 16, 16, k=64).close()              # variance models: up to 64 images
    with pytest.raises(E.CpEngineError, match="unsupported"):
        E.Engine(E.MODEL_VAR_TV, 16, 16, k=65)
    with pytest.raises(E.CpEngineError, match="badArg"):
        E.Engine(E.MODEL_TV_L1_REG, 1, 16)
    with pytest.raises(E.CpEngineError, match="badArg"):
        E.Engine(99, 16, 16)
    with pytest.raises(E.CpEngineError, match="badArg"):
        E.Engine(E.MODEL_TV_L1_REG, 16, 16, h=(0.0, 1.0))
    with pytest.raises(E.CpEngineError, match="unsupported"):    # slabs: no linear boundary row, no variance models
        E.Engine(E.MODEL_MF_NN_TV, 16, 8, k=3, bc="linear", n_global=16, j_offset=0)
    with pytest.raises(E.CpEngineError, match="unsupported"):
        E.Engine(E.MODEL_VAR_TV, 16, 8, k=3, n_global=16, j_offset=0)
    with pytest.raises(E.CpEngineError, match="badArg"):
        E.Engine(E.MODEL_MF_NN_TV, 16, 8, k=3, n_global=12, j_offset=8)


def test_state_errors():
    from ma_amd import engine as E
    eng = E.Engine(E.MODEL_MF_NN_TV, 16, 12, k=3)
    with pytest.raises(E.CpEngineError, match="badState"):
        eng.linearize()                                            # no images yet
    eng.set_images(np.random.default_rng(0).random(3 * 16 * 12))
    with pytest.raises(E.CpEngineError, match="badState"):
        eng.solve(1.0, 0.1, 0.1, 5, 1e-3)                          # not linearised
    eng.linearize()
    with pytest.raises(E.CpEngineError, match="badState"):
        eng.solve(1.0, 0.1, 0.1, 5, 1e-3)                          # nu not set
    with pytest.raises(E.CpEngineError, match="badArg"):
        eng.set_nu(0.0)
    eng.set_nu(1.0)
    with pytest.raises(E.CpEngineError, match="badArg"):
        eng.solve(1.0, -0.1, 0.1, 5, 1e-3)
    r = eng.solve(1.0, 0.1, 0.1, 0, 1e-3)                          # maxIter = 0: row 0 only
    ph, dh = eng.history()
    assert r.iterations == 0 and ph.shape == (1, 7)
    with pytest.raises(E.CpEngineError, match="badState"):
        eng.iterate_steps(1)
    eng.close()
    pw = E.Engine(E.MODEL_TV_L1_REG, 16, 12, lam=1.0)
    with pytest.raises(E.CpEngineError, match="badArg"):
        pw.set_nu(1.0)
    with pytest.raises(E.CpEngineError, match="badArg"):
        pw.set_images(np.zeros(16 * 12))                           # reference image missing
    pw.close()
