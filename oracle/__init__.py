"""CPU oracle for the Chambolle-Pock registration hot path of roland1993/MA.

TEST INFRASTRUCTURE ONLY.  This package is a numpy/scipy restatement of the
reference's MATLAB algorithm (file:line cited per function).  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it, and
only as the checker -- never as the thing shipped or measured.  The product
(`ma_amd/`) never imports it and fails loudly when the HIP library is missing.

PARITY UNPINNED (bit level): the reference is MATLAB source; neither MATLAB nor
Octave exists in the build container, and the reference stores no golden vectors
or expected outputs.  The restatement is pinned only by
  * the invariants the reference's own interactive tests check
    (IRE/test_interpolation.m:39-42, NNE/svd_2x2.m:52-56,
    NNE/test_l1ball_projection.m:45-56, PDO/test_chambolle_pock.m:16-41), and
  * the deterministic inputs the reference ships (Data/*.png, committed as
    fixtures under tests/golden/).
Third-party arithmetic the reference delegates to MATLAB builtins
(svd 'econ', sparse backslash, conv2) is restated by its mathematical
definition with LAPACK/scipy.

Conventions: every array is float64; vectors are 1-D; images are (m, n) and all
reshapes are column-major (order='F') exactly like MATLAB.  Operators follow the
reference contract  op(v, conjugate_flag, nargout) -> (res1, res2, res3)  where
`nargout` (2 or 3) selects the outputs computed (PDO/chambolle_pock.m:20-32).
"""
