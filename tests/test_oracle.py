"""Pins the CPU oracle (oracle/rflu_oracle.c) against the properties the reference's own tests assert
(/root/reference/test/runtests.jl) with LAPACK getrf as comparator -- the same comparator the reference uses
(`baselu = LinearAlgebra.lu`, runtests.jl:11).  CPU only."""
import glob
import os

import numpy as np
import pytest
import scipy.linalg as sla

import oracle as O
from helpers import REF_SIZES, nopivot_lu_numpy, rand_matrix, wilkinson


def getrf(A):
    f = sla.lapack.dgetrf if A.dtype == np.float64 else sla.lapack.sgetrf
    lu, piv, info = f(np.asfortranarray(A))
    return lu, piv.astype(np.int64) + 1, int(info)


def tol_E(A):
    # runtests.jl:19   E = 20 * size(A,1) * eps(T)
    return 20 * A.shape[0] * np.finfo(A.dtype).eps


def test_generator_mirror_is_bit_exact():
    for dt in (np.float64, np.float32):
        assert np.array_equal(O.fill_uniform(37, 21, 12, dt), O.np_uniform(37, 21, 12, dt))
    A = O.np_uniform(64, 64, 7)
    assert 0.0 <= A.min() and A.max() < 1.0 and abs(A.mean() - 0.5) < 0.02


def test_nsplit_matches_reference_rule():
    # src/lu.jl:158-162 ; SURVEY 3.3: 512 -> 256 -> 128 -> 64 -> 32 -> 16 -> 8 (Float64)
    chain, n = [], 512
    while n > 8:
        n = O.nsplit(np.float64, n)
        chain.append(n)
    assert chain == [256, 128, 64, 32, 16, 8]
    assert O.nsplit(np.float64, 15) == 7 and O.nsplit(np.float64, 16) == 8 and O.nsplit(np.float64, 300) == 152
    assert O.nsplit(np.float32, 31) == 15 and O.nsplit(np.float32, 32) == 16 and O.nsplit(np.float32, 16) == 8


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("s", REF_SIZES)
def test_pivoted_lu_matches_lapack_properties(dtype, s):
    # runtests.jl:33-68 with pivot = Val(true): square and fat (s, s+2); info equality, residual bound, solve check.
    for m in (s, s + 2):
        A = rand_matrix(s, m, seed=1000 * s + m, dtype=dtype)
        F, ipiv, info = O.lu(A, pivot=True)
        _, lpiv, linfo = getrf(A)
        assert info == linfo == 0
        assert np.array_equal(ipiv, lpiv)  # stronger than the reference's own test (tie-free input)
        mx, _ = O.residual(A, F, ipiv)
        assert mx < tol_E(A)
        if s == m:  # runtests.jl:21-28  b = ldiv!(MF, A[:, end]) ~ e_n
            lu_piv = (F, ipiv - 1)
            b = sla.lu_solve(lu_piv, A[:, -1].astype(dtype))
            if np.all(np.isfinite(b)):
                rhs = np.zeros(s)
                rhs[-1] = 1
                assert np.allclose(b, rhs, atol=100 * tol_E(A), rtol=0)
        # runtests.jl:59-64: zero a column, check=false, info must equal LAPACK's
        i = (7 * s + m) % s
        A2 = A.copy()
        A2[:, i] = 0
        _, ipiv2, info2 = O.lu(A2, pivot=True)
        _, lpiv2, linfo2 = getrf(A2)
        assert info2 == linfo2 == i + 1


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("s", REF_SIZES)
def test_unpivoted_lu_properties(dtype, s):
    # runtests.jl:33-68 with pivot = Val(false): bound 10*sqrt(E); info vs an unpivoted generic LU.
    for m in (s, s + 2):
        A = rand_matrix(s, m, seed=2000 * s + m, dtype=dtype)
        F, ipiv, info = O.lu(A, pivot=False)
        _, ninfo = nopivot_lu_numpy(A)
        assert info == ninfo
        assert np.array_equal(ipiv, np.arange(1, s + 1))
        if info == 0:
            mx, _ = O.residual(A, F, ipiv)
            if np.isfinite(mx):
                assert mx < 10 * np.sqrt(tol_E(A)) * max(1.0, float(np.max(np.abs(F))))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_nopivot_user_ipiv_becomes_identity(dtype):
    # runtests.jl:70-84: poisoned ipiv must come back == 1:n ; matrix rand + 10I
    n = 30
    A = rand_matrix(n, n, seed=3, dtype=dtype) + dtype(10) * np.eye(n, dtype=dtype)
    F, ipiv, info = O.lu(A, pivot=False, poison_ipiv=np.iinfo(np.int64).max - 7)
    assert info == 0 and np.array_equal(ipiv, np.arange(1, n + 1))
    b = rand_matrix(n, 1, seed=4, dtype=dtype)[:, 0]
    x = sla.lu_solve((F, ipiv - 1), b)
    assert np.linalg.norm(A.astype(np.float64) @ x - b) < 1000 * n * np.finfo(dtype).eps


def test_nopivot_zero_pivot_reports_info_and_continues():
    A = rand_matrix(60, 60, seed=5) + 10 * np.eye(60)
    A[0, 0] = 0.0  # first pivot exactly zero -> info = 1, factorization continues (src/lu.jl:321-334)
    _, _, info = O.lu(A, pivot=False)
    assert info == 1
    # a zero pivot deep inside the right half of the recursion exercises the info offset (src/lu.jl:248-255)
    A = np.asfortranarray(np.triu(rand_matrix(100, 100, seed=6)) + 10 * np.eye(100))
    A[70, 70] = 0.0
    _, _, info = O.lu(A, pivot=False)
    assert info == 71


def test_wilkinson_all_ties_lowest_index():
    # runtests.jl:130-140 generator; every pivot search is an exact tie -> lowest index -> identity pivots,
    # growth 2^(k-1) in the last column.
    for n in (50, 130):
        A = wilkinson(n)
        F, ipiv, info = O.lu(A, pivot=True)
        assert info == 0 and np.array_equal(ipiv, np.arange(1, n + 1))
        assert np.array_equal(F[:, -1][: n], 2.0 ** np.arange(n))
        _, lpiv, _ = getrf(A)
        assert np.array_equal(ipiv, lpiv)


def test_nan_never_wins_argmax_and_zero_column():
    # src/lu.jl:298-304: amax starts at 0 with strict '>' -> NaN is never selected; all-zero column -> kp = k
    A = rand_matrix(64, 64, seed=8)
    A[5, 0] = np.nan
    A[9, 0] = 0.999999
    _, ipiv, _ = O.generic_lufact(A)
    assert ipiv[0] == 10
    Z = rand_matrix(20, 20, seed=9)
    Z[:, 3] = 0
    _, ipiv, info = O.generic_lufact(Z)
    assert info == 4 and ipiv[3] == 4


def test_blocksize_and_threshold_do_not_change_pivots():
    A = rand_matrix(300, 300, seed=10)
    base = O.lu(A)[1]
    for bs in (1, 8, 16, 64):
        for th in (0, 40, 48, 1000):
            assert np.array_equal(O.lu(A, blocksize=bs, threshold=th)[1], base)


def test_tall_matrix():
    A = rand_matrix(400, 130, seed=11)
    F, ipiv, info = O.lu(A)
    _, lpiv, _ = getrf(A)
    assert info == 0 and np.array_equal(ipiv, lpiv)
    assert O.residual(A, F, ipiv)[0] < tol_E(A)


GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_oracle_reproduces_golden_fixtures(path):
    from golden.make_golden import build_input

    g = np.load(path, allow_pickle=False)
    A = build_input(g)
    F, ipiv, info = O.lu(A, pivot=bool(g["pivot"]))
    assert info == int(g["info"])
    assert np.array_equal(ipiv, g["ipiv"])
    tol = 64 * np.finfo(A.dtype).eps * max(A.shape) * max(1.0, float(np.max(np.abs(g["lu_sample"]))))
    idx = g["sample_idx"]
    got = F.ravel(order="F")[idx]
    ok = np.isfinite(g["lu_sample"])
    assert np.allclose(got[ok], g["lu_sample"][ok], atol=tol, rtol=0)
    assert len(GOLDEN) > 0
