"""The C-ABI library loads without a GPU and exports every symbol include/cpengine.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "cpengine.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(cp_[A-Za-z_0-9]+)\s*\(", txt)))


def test_header_symbols_exported():
    from ma_amd import engine as E
    names = _declared()
    assert len(names) >= 20
    lib = ctypes.CDLL(E.LIB_PATH)
    for nm in names:
        assert hasattr(lib, nm), f"{nm} declared in include/cpengine.h but not exported"
    assert sorted(E.EXPORTED) == names, "ma_amd/engine.py binds a different set of entry points than the header declares"


def test_version_and_error_string_without_gpu():
    from ma_amd import engine as E
    assert b"cpengine" in E.lib.cp_version()
    assert isinstance(E.lib.cp_last_error(), bytes)


def test_struct_layout_matches_header():
    from ma_amd import engine as E
    # cp_problem_desc: 8 x int32, 2+4+2 doubles, 2 x int32 (+ padding to 8)
    assert ctypes.sizeof(E.ProblemDesc) == 8 * 4 + 8 * 8 + 2 * 4
    assert ctypes.sizeof(E.Result) == 16


def test_bad_arguments_are_reported_not_crashing():
    from ma_amd import engine as E
    import pytest
    with pytest.raises(E.CpEngineError, match="badBC|Unknown boundary"):
        E.Engine(E.MODEL_MF_NN_TV, 32, 32, k=2, bc="periodic")
    h = ctypes.c_void_p()
    d = E.ProblemDesc()
    d.model, d.m, d.n, d.k = 99, 32, 32, 1
    d.h[0] = d.h[1] = 1.0
    assert E.lib.cp_create(ctypes.byref(d), ctypes.byref(h)) != 0
    assert b"badArg" in E.lib.cp_last_error()
    d.model, d.k = E.MODEL_MF_NN_TV, 33      # mf_nn: up to 32 images (17..32 on the general path)
    assert E.lib.cp_create(ctypes.byref(d), ctypes.byref(h)) != 0
    assert b"unsupported" in E.lib.cp_last_error()
