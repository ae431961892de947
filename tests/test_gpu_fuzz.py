"""Random shapes of the four multi-image models against the numpy oracle (tests/fuzz_cases.py): sizes that are not
multiples of the 64-row segments, 3..66 columns (fewer than the waves of a workgroup, more than one strip of k_curv_y2),
K = 2..16 (2..23 for the variance models), both boundary conditions of each regulariser, anisotropic grid spacings, and
the 4- and 8-wave instantiations forced at these sizes; and of the pairwise TV-L1 / TV-L2 models with random strip widths of the
marching kernel (round 3: k_curv_y2_r2 and k_pw_march2, the two-rows-per-lane kernels, forced at even sizes below their default
thresholds in half of the draws). 750 multi-image and 60 pairwise cases of the same generators ran clean on the final tree of
round 2, 1000 and 450 (three seeds each) on that of round 3 (scratch/fuzz_parity.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12])
def test_random_shapes_match_the_oracle(seed, monkeypatch):
    from fuzz_cases import run_case
    monkeypatch.delenv("CP_TUNE_FUSED_WAVES", raising=False)
    monkeypatch.delenv("CP_TUNE_VAR_WAVES", raising=False)
    rng = np.random.default_rng(seed)
    try:
        for t in range(16):
            r = run_case(t, rng)
            assert r["ok"], r
    finally:
        import os
        os.environ.pop("CP_TUNE_FUSED_WAVES", None)
        os.environ.pop("CP_TUNE_VAR_WAVES", None)


def test_random_pairwise_shapes_match_the_oracle(monkeypatch):
    from fuzz_cases import run_pair_case
    monkeypatch.delenv("CP_TUNE_PW_STRIP", raising=False)
    rng = np.random.default_rng(41)
    try:
        for t in range(14):
            r = run_pair_case(t, rng)
            assert r["ok"], r
    finally:
        import os
        os.environ.pop("CP_TUNE_PW_STRIP", None)
