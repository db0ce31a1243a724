"""Host-pointer entry at sizes where the transfers overlap the factorization (driver.cpp: getrf_host / getrf_host_engine, n >= 8192).
Stream schedules (Float32, NoPivot, fat matrices, RFLU_ENGINE_HOST=0): finished block rows leave through a fourth stream, pinned
bounce buffers and a threaded scatter into the caller's columns while the rest is factored -- same factors, pivots and info as the
device entry, to the bit.  Float64 with pivoting, square or tall (round 5): the matrix also ARRIVES while it is factored, through the
update engine (engine.hip), whose summation order differs from the stream schedules': pivots and info equal, factors equal to rounding."""
import ctypes
import numpy as np
import pytest
import torch

import recursivefactorization.jl_amd as rf
from recursivefactorization.jl_amd import _ffi
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _device_reference(A, pivot, bs):
    W = torch.from_numpy(np.ascontiguousarray(A.T)).to("cuda:0").T      # column-major device view
    F = rf.lu_(W, None, pivot, check=False, blocksize=bs)
    ip = F.ipiv.cpu().numpy() if pivot else None
    return F.factors.cpu().numpy(), ip, F.info


@pytest.mark.parametrize("m,n,dtype,pivot,bs", [
    (8192, 8192, np.float64, True, None),     # all leaf-wise
    (12288, 12288, np.float64, True, None),   # block-column lookahead first, then leaf-wise
    (10000, 8200, np.float64, True, None),    # tall: the rows below the square part are final only at the end
    (8192, 9000, np.float64, True, 256),      # fat
    (8192, 8192, np.float32, True, None),
    (8192, 8192, np.float64, False, None),    # NoPivot
])
def test_host_entry_matches_device_entry(m, n, dtype, pivot, bs, monkeypatch):
    A = O.fill_uniform(m, n, 5 + m + n, dtype)
    if not pivot:
        A[np.arange(min(m, n)), np.arange(min(m, n))] += 10.0
    ref, ipr, infr = _device_reference(A, pivot, bs)
    variants = [{}]
    if (m, n, dtype, pivot) == (8192, 8192, np.float64, True):   # other chunkings / thread counts, and the plain sequence
        variants += [{"RFLU_HOST_EARLY_OUT": "1024", "RFLU_HOST_THREADS": "3"}, {"RFLU_HOST_EARLY_OUT": "0"},
                     {"RFLU_ENGINE_HOST": "0"}, {"RFLU_ENGINE_HOST": "0", "RFLU_HOST_EARLY_OUT": "1024", "RFLU_HOST_THREADS": "3"}]
    for env in variants:
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        H = np.asfortranarray(A.copy())
        F = rf.lu_(H, None, pivot, check=False, blocksize=bs)
        assert F.info == infr
        through_engine = (pivot and m >= n and env.get("RFLU_HOST_EARLY_OUT", "512") != "0" and env.get("RFLU_ENGINE_HOST", "1") != "0")
        if through_engine and dtype == np.float32:
            # Float32 through the engine (round 6): another summation order than the device entry's stream schedule at this size -- held to what
            # the reference holds Float32 to (info, the residual bound, test/runtests.jl:19-20) and a floor of equal leading pivots
            ip = np.asarray(F.ipiv)
            same = int(np.argmax(ip != ipr)) if (ip != ipr).any() else len(ip)
            assert same >= 1024, same
            res, _ = O.residual(A, np.asarray(F.factors), ip)
            assert res <= 4 * 20 * min(m, n) * np.finfo(np.float32).eps * max(1.0, float(np.abs(A).max()))
        else:
            if pivot:
                assert np.array_equal(np.asarray(F.ipiv), ipr)
            if through_engine:
                assert np.abs(np.asarray(F.factors) - ref).max() <= 1e-10 * np.abs(ref).max(), env
            else:
                assert np.array_equal(np.asarray(F.factors), ref), env
        for k in env:
            monkeypatch.delenv(k)


def test_host_entry_headline_size_through_the_engine():
    """n = 16384 Float64 pivoted, the size `host_entry` of the bench line is quoted on: the matrix arrives block column by block column
    while the engine factors what is there (getrf_host_engine), block rows go home as they become final -- pivots and info equal to
    the device entry's, factors equal to rounding (the device entry of this size runs the same engine: same summation order per column
    block, but which workgroup finishes a tile first is not fixed, so `equal to rounding`, not `to the bit`)."""
    n = 16384
    A = O.fill_uniform(n, n, 12, np.float64)
    ref, ipr, infr = _device_reference(A, True, None)
    assert rf.last_path() == "hip-engine" and infr == 0
    H = np.asfortranarray(A)
    del A
    F = rf.lu_(H, None, True, check=False)
    assert rf.last_path() == "hip-engine"
    assert F.info == 0
    assert np.array_equal(np.asarray(F.ipiv), ipr)
    assert np.abs(np.asarray(F.factors) - ref).max() <= 1e-10 * np.abs(ref).max()


def test_host_entry_with_a_column_stride_and_reuse():
    """lda > m (a view into a taller buffer), the same handle and bounce buffers used for a second, smaller matrix."""
    n, lda = 8192, 8192 + 24
    A = O.fill_uniform(n, n, 77, np.float64)
    ref, ipr, _ = _device_reference(A, True, None)
    buf = np.full((lda, n), np.nan, order="F")
    buf[:n, :] = A
    h = _ffi.default_handle(0)
    h.set_stream(None)
    ipiv = np.empty(n, dtype=np.int64)
    info = ctypes.c_int64(0)
    h.call("rflu_getrf_f64", n, n, ctypes.c_void_p(buf.ctypes.data), lda, ctypes.c_void_p(ipiv.ctypes.data), 1, 0, ctypes.byref(info))
    assert info.value == 0
    assert np.array_equal(ipiv, ipr)
    assert np.abs(buf[:n, :] - ref).max() <= 1e-10 * np.abs(ref).max()   # (through the engine: equal to rounding)
    assert np.isnan(buf[n:, :]).all()          # nothing written below the matrix
    B = np.asfortranarray(A[:8192 - 512, :8192 - 512].copy())
    refB, ipB, _ = _device_reference(A[:8192 - 512, :8192 - 512], True, None)
    F = rf.lu_(B, None, True, check=False)   # 7680 rows: below the engine's size, the stream path: bit-identical
    assert np.array_equal(np.asarray(F.factors), refB) and np.array_equal(np.asarray(F.ipiv), ipB)


@pytest.mark.parametrize("ghost_leaf,engine_host", [(100, "1"), (3, "1"), (100, "0"), (3, "0")])
def test_failed_factorization_leaves_the_callers_matrix_untouched(ghost_leaf, engine_host, monkeypatch):
    """Finished block rows travel home while the rest is factored -- but a panel timeout (or a placement error) is only known at the
    end.  RFLU_DEBUG_GHOST_LEAF makes one cooperative leaf wait for a participant that does not exist: its bounded spins run out,
    the call returns RFLU_ERR_TIMEOUT -- and the caller's host matrix must be bit-identical to the input (rows that already went
    home are taken back from the device copy of the input, which is only overwritten after success), so that a host which falls
    back to another solver after the error factors the right matrix (the reference's boundary: lu! of a host array,
    src/lu.jl:116-121)."""
    n = 8192
    A = O.fill_uniform(n, n, 91, np.float64)
    H = np.asfortranarray(A.copy())
    monkeypatch.setenv("RFLU_ENGINE_HOST", engine_host)
    monkeypatch.setenv("RFLU_DEBUG_GHOST_LEAF", str(ghost_leaf))
    with pytest.raises(rf.RfluError) as exc:
        rf.lu_(H, None, True, check=False)
    assert "timed out" in str(exc.value)
    assert np.array_equal(H, A), "the caller's matrix was modified by a failed call"
    monkeypatch.delenv("RFLU_DEBUG_GHOST_LEAF")
    # the handle is usable afterwards and gives the right answer
    ref, ipr, infr = _device_reference(A, True, None)
    F = rf.lu_(H, None, True, check=False)
    assert F.info == infr == 0
    assert np.array_equal(np.asarray(F.ipiv), ipr)
    assert np.abs(np.asarray(F.factors) - ref).max() <= 1e-10 * np.abs(ref).max()
