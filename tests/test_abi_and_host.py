"""CPU-only: the C-ABI shared library loads and exports every symbol include/rflu.h declares (no compute calls without a
GPU), and the host-side mirror of the reference interface behaves like src/lu.jl's helpers."""
import os
import re

import numpy as np
import pytest

import __graft_entry__ as entry

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    entry.build()  # hipcc cross-compiles for gfx950 without a GPU; cached by content hash
    from recursivefactorization.jl_amd import _ffi

    return _ffi.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rflu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rflu_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    from recursivefactorization.jl_amd import _ffi

    names = declared_symbols()
    assert len(names) >= 30
    for name in names:
        assert hasattr(lib, name), f"{name} is declared in include/rflu.h but not exported by librflu.so"
        assert name in _ffi.EXPORTS, f"{name} has no ctypes prototype in _ffi.py"
    assert lib.rflu_version() >= 100


def test_no_cpu_fallback_without_a_device(lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import recursivefactorization.jl_amd as rf

    with pytest.raises(rf.RfluError, match="no HIP device|no CPU fallback"):
        rf.Handle(0)
    with pytest.raises(rf.RfluError):
        rf.lu(np.asfortranarray(np.eye(4)))


def test_host_mirror_of_reference_helpers():
    import recursivefactorization.jl_amd as rf

    # normalize_pivot (src/lu.jl:10-17): both spellings
    assert rf.normalize_pivot(rf.Val(True)) is True and rf.normalize_pivot(rf.RowMaximum()) is True
    assert rf.normalize_pivot(rf.Val(False)) is False and rf.normalize_pivot(rf.NoPivot()) is False
    with pytest.raises(TypeError):
        rf.normalize_pivot("yes")
    # NotIPIV (src/lu.jl:27-40): lazy identity, views keep the type
    p = rf.NotIPIV(5)
    assert len(p) == 5 and [p[i] for i in range(5)] == [1, 2, 3, 4, 5] and isinstance(p[1:4], rf.NotIPIV) and len(p[1:4]) == 3
    assert np.array_equal(np.asarray(p), np.arange(1, 6))
    # LU accessors (LinearAlgebra.LU's L, U, p) on a hand-made factorization
    F = rf.LU(np.array([[4.0, 3.0], [0.5, 1.5]], order="F"), np.array([2, 2]), 0)
    assert np.array_equal(F.L, [[1, 0], [0.5, 1]]) and np.array_equal(F.U, [[4, 3], [0, 1.5]]) and list(F.p) == [1, 0]
    assert F.issuccess() and not rf.LU(F.factors, F.ipiv, 3).issuccess()
    # element types outside Float32/Float64 are refused loudly (the reference routes them to generic CPU code)
    with pytest.raises(TypeError):
        rf.lu(np.asfortranarray(np.eye(3, dtype=np.complex128)))
    with pytest.raises(ValueError):
        rf.lu_(np.ascontiguousarray(np.arange(6.0).reshape(2, 3)))  # lu! needs column-major storage
