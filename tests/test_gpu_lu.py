"""-m gpu: parity of the HIP path with the CPU oracle through the reference-shaped API (lu / lu_), mirroring
/root/reference/test/runtests.jl.  Bars: ipiv and info bit-exact (integer work); factors and residual within the
reference's own Float64/Float32 tolerance  max|L*U - A[p,:]| < 20*s*eps  (runtests.jl:19-20), written in each test."""
import glob
import os

import numpy as np
import pytest
import scipy.linalg as sla
import torch

import oracle as O
import recursivefactorization.jl_amd as rf
from gpu_util import to_dev_cm
from helpers import REF_SIZES, rand_matrix, wilkinson

pytestmark = pytest.mark.gpu


def tol_E(A):
    return 20 * A.shape[0] * np.finfo(A.dtype).eps


def check_against_oracle(A, F, pivot=True, factor_tol_mult=50):
    Fo, ipo, infoo = O.lu(A, pivot=pivot)
    lu_host = F.factors.cpu().numpy() if hasattr(F.factors, "cpu") else np.asarray(F.factors)
    ip = np.asarray(F.ipiv.cpu().numpy() if hasattr(F.ipiv, "cpu") else F.ipiv)
    assert abs(F.info) == infoo
    assert np.array_equal(ip, ipo), "ipiv must be bit-exact"
    if infoo == 0:
        mx, fro = O.residual(A, lu_host, ip)
        bound = tol_E(A) if pivot else 10 * np.sqrt(tol_E(A)) * max(1.0, float(np.max(np.abs(Fo))))
        assert mx < bound
        scale = max(1.0, float(np.max(np.abs(Fo))))
        # pivoted: backward-stable, factors agree to a small multiple of E; unpivoted LU is not (the reference itself only
        # asks 10*sqrt(E) of it, runtests.jl:20), so two correct summation orders may differ by that much
        ftol = factor_tol_mult * tol_E(A) if pivot else 10 * np.sqrt(tol_E(A))
        assert np.max(np.abs(lu_host - Fo)) < ftol * scale
    return lu_host, ip


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("s", REF_SIZES)
def test_lu_reference_shapes_pivoted(dtype, s):
    # runtests.jl:33-68, pivot = Val(true) / RowMaximum(), square and fat, host boundary (numpy) and device boundary
    for m in (s, s + 2):
        A = rand_matrix(s, m, seed=1000 * s + m, dtype=dtype)
        F = rf.lu(A, rf.RowMaximum() if s % 2 else rf.Val(True), check=False)
        assert rf.last_path() == "hip-recursive"
        check_against_oracle(A, F)
        Fd = rf.lu_(to_dev_cm(A), None, True, check=False)
        check_against_oracle(A, Fd)
        # singular: zero a column, check=false, info must match (runtests.jl:59-64)
        i = (7 * s + m) % s
        A2 = A.copy()
        A2[:, i] = 0
        F2 = rf.lu(A2, True, check=False)
        assert F2.info == i + 1
        assert np.array_equal(F2.ipiv, O.lu(A2)[1])
        with pytest.raises(rf.SingularException):
            rf.lu(A2, True)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("s", [1, 2, 7, 10, 50, 130, 300])
def test_lu_nopivot(dtype, s):
    # runtests.jl:33-68 with pivot = Val(false)/NoPivot(): bound 10*sqrt(E); NotIPIV result; identity fill of a user ipiv
    for m in (s, s + 2):
        A = (rand_matrix(s, m, seed=2000 * s + m, dtype=dtype) + dtype(10) * np.eye(s, m, dtype=dtype)).astype(dtype, order="F")
        F = rf.lu(A, rf.NoPivot(), check=False)
        assert isinstance(F.ipiv, rf.NotIPIV) and len(F.ipiv) == s
        check_against_oracle(A, rf.LU(F.factors, np.arange(1, s + 1), F.info), pivot=False)
    # runtests.jl:70-84: poisoned user ipiv comes back == 1:n
    n = 30
    A = (rand_matrix(n, n, seed=3, dtype=dtype) + dtype(10) * np.eye(n, dtype=dtype)).astype(dtype, order="F")
    ipiv = np.full(n, np.iinfo(np.int64).max - 7, dtype=np.int64)
    F = rf.lu_(A.copy(order="F"), ipiv, rf.Val(False), rf.Val(False))
    assert F.ipiv is ipiv and np.array_equal(ipiv, np.arange(1, n + 1))
    b = rand_matrix(n, 1, seed=4, dtype=dtype)[:, 0]
    x = sla.lu_solve((F.factors, ipiv - 1), b)
    assert np.linalg.norm(A.astype(np.float64) @ x - b) < 1000 * n * np.finfo(dtype).eps


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("pivot", [True, False])
@pytest.mark.parametrize("s", [3, 10, 50, 130, 300])
def test_lu_adjoint_transpose_wrappers(dtype, pivot, s):
    # test/runtests.jl:55-58: A' = permutedims(A); MF' = lu(A'' ...) -- the wrapper forwards to the parent and wraps the
    # result (src/lu.jl:85-87); testlu then looks at parent(MF') against parent(A'') = A' (tall when A is fat)
    for m in (s, s + 2):
        A = rand_matrix(s, m, seed=4000 * s + m, dtype=dtype)
        if not pivot:
            A = (A + dtype(10) * np.eye(s, m, dtype=dtype)).astype(dtype, order="F")
        At = np.asfortranarray(A.T.copy())           # permutedims(A): an m x s matrix in its own storage
        for wrap in (rf.Adjoint, rf.Transpose):
            MF = rf.lu(wrap(At), rf.Val(pivot), check=False)
            assert isinstance(MF, rf.Adjoint)
            P = MF.parent                            # LU of A' itself
            if pivot:
                check_against_oracle(At, P)
            else:
                assert isinstance(P.ipiv, rf.NotIPIV)
                check_against_oracle(At, rf.LU(P.factors, np.arange(1, min(At.shape) + 1), P.info), pivot=False)
        # in-place form on a device matrix: lu!(A'') factors the parent's storage
        dAt = to_dev_cm(At)
        MF = rf.lu_(rf.Adjoint(dAt), None, pivot, check=False)
        assert MF.parent.factors is dAt
        if pivot:
            check_against_oracle(At, MF.parent)


def test_nopivot_zero_pivot_sign_convention():
    A = np.asfortranarray(np.triu(rand_matrix(100, 100, seed=6)) + 10 * np.eye(100))
    A[70, 70] = 0.0
    F = rf.lu(A, rf.NoPivot(), check=False)
    assert F.info == (-71 if rf.NOPIVOT_NEGATIVE_INFO else 71)  # src/lu.jl:249-254, 323-326
    with pytest.raises(rf.SingularException):
        rf.lu(A, rf.NoPivot())


@pytest.mark.parametrize("n,dtype", [(512, np.float64), (1000, np.float64), (2048, np.float64), (1000, np.float32),
                                     (2048, np.float32)])
def test_lu_medium_sizes_vs_oracle(n, dtype):
    A = rand_matrix(n, n, seed=12, dtype=dtype)
    F = rf.lu_(to_dev_cm(A), None, True, check=False)
    check_against_oracle(A, F)


@pytest.mark.parametrize("shape", [(400, 130), (1000, 64), (130, 400), (64, 1000), (777, 333)])
def test_lu_tall_and_fat(shape):
    A = rand_matrix(shape[0], shape[1], seed=77)
    F = rf.lu(A, True, check=False)
    check_against_oracle(A, F)


@pytest.mark.parametrize("blocksize", [64, 128, 256])
def test_blocked_lookahead_variant_gives_same_pivots(blocksize):
    # BASELINE config 3's block-size sweep: right-looking block columns + one block column of lookahead on two streams
    A = rand_matrix(1000, 1000, seed=10)
    F = rf.lu(A, True, check=False, blocksize=blocksize)
    assert rf.last_path() == "hip-lookahead"
    check_against_oracle(A, F)
    G = rf.lu(A, True, check=False, blocksize=-1)
    assert rf.last_path() == "hip-recursive"
    assert np.array_equal(G.ipiv, F.ipiv)


@pytest.mark.parametrize("shape,blocksize", [((900, 1300), 256), ((1300, 900), 256), ((2048, 2048), 512), ((700, 700), 128)])
def test_lookahead_shapes(shape, blocksize):
    A = rand_matrix(shape[0], shape[1], seed=31)
    F = rf.lu(A, True, check=False, blocksize=blocksize)
    assert rf.last_path() == "hip-lookahead"
    check_against_oracle(A, F)
    N = (rand_matrix(shape[0], shape[1], seed=32) + 10 * np.eye(*shape)).astype(np.float64, order="F")
    Fn = rf.lu(N, rf.NoPivot(), check=False, blocksize=blocksize)
    check_against_oracle(N, rf.LU(Fn.factors, np.arange(1, min(shape) + 1), Fn.info), pivot=False)


@pytest.mark.parametrize("shape,blocksize,env", [
    ((20480, 1024), 128, {"RFLU_SPLIT_SCALE": "0.02"}),     # panels of 33..40 workgroups: 64 CUs reserved, restA/restB split
    ((20480, 1024), 128, {}),                                # same panels, model-sized split (everything fits on U)
    ((40000, 640), 128, {}),                                 # panels of 78 workgroups: single-stream block columns
    ((3000, 3000), 256, {"RFLU_SPLIT_ALL": "1", "RFLU_SPLIT_SCALE": "0.1"}),   # split forced on ordinary panels
    ((1500, 2600), 256, {"RFLU_SPLIT_ALL": "1", "RFLU_SPLIT_SCALE": "0.1"}),   # ... and on a fat matrix
])
def test_lookahead_schedules_tall_panels_and_split(shape, blocksize, env, monkeypatch):
    # the lookahead driver picks the CU reservation per block column and may hand the tail of an update to the panel
    # stream (driver.cpp factor_lookahead): same operations on the same columns, so pivots must not move
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    A = rand_matrix(shape[0], shape[1], seed=41)
    F = rf.lu(A, True, check=False, blocksize=blocksize)
    assert rf.last_path() == "hip-lookahead"
    check_against_oracle(A, F)


def test_wilkinson_single_leaf_exact_growth():
    # n <= 64: one leaf, no inverse products anywhere -- the growth 2^(k-1) is exact (test/runtests.jl:130-140)
    for n in (17, 64):
        W = wilkinson(n)
        G = rf.lu(W, True, check=False, blocksize=-1)
        assert G.info == 0 and np.array_equal(G.ipiv, np.arange(1, n + 1))
        assert np.array_equal(np.asarray(G.factors)[:, -1], 2.0 ** np.arange(n))


@pytest.mark.parametrize("shape,bs", [((1000, 1000), 0), ((2048, 2048), 512), ((700, 384), 0), ((300, 900), 0)])
def test_more_geometries_against_oracle(shape, bs):
    A = rand_matrix(shape[0], shape[1], seed=77)
    F = rf.lu(A, True, check=False, blocksize=bs)
    check_against_oracle(A, F)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_wilkinson_all_ties_exact_growth(dtype):
    # all-ties pivoting (test/runtests.jl:130-140): identity pivots, exact growth 2^(k-1) (Float32 stays finite up to 2^127)
    n = 256 if dtype == np.float64 else 128
    W = wilkinson(n).astype(dtype, order="F")
    G = rf.lu(W, True, check=False, blocksize=-1)
    assert G.info == 0 and np.array_equal(G.ipiv, np.arange(1, n + 1))
    # growth 2^(k-1) in the last column; the second leaf's rows come through inv(L11) products on the MFMAs (sums of up to 63
    # powers of two: rounded, not exact, beyond the mantissa), hence a few ulps
    want = (2.0 ** np.arange(n)).astype(dtype)
    assert np.allclose(np.asarray(G.factors)[:, -1], want, rtol=8 * np.finfo(dtype).eps, atol=0)


@pytest.mark.parametrize("n,nrhs", [(129, 8), (1000, 9), (3000, 1), (3000, 20), (4100, 64), (2000, 70), (5000, 5), (1000, 33),
                                    (2100, 130), (300, 64), (4100, 400)])
def test_ldiv_cooperative_and_recursive_paths(n, nrhs):
    # up to 32 right-hand sides: one cooperative launch per triangle and pass of 8 (trsv.hip: trsv_chain_kernel); 33 .. 320: the same
    # chain in passes of 64 columns on the MFMA units (trsm_chain_kernel, round 5); beyond: recursive TRSM/GEMM.
    # Same bound as runtests.jl:126-128, on a general (pivoted) matrix and through the device entry
    A = rand_matrix(n, n, seed=900 + n)
    B = rand_matrix(n, nrhs, seed=901 + n).copy(order="F")
    dF = rf.lu_(to_dev_cm(A), None, True)
    dB = to_dev_cm(B)
    rf.ldiv_(dF, dB)
    X = dB.cpu().numpy()
    Xref = np.linalg.solve(A, B)
    scale = np.linalg.norm(A, 2) * np.linalg.norm(Xref) + np.linalg.norm(B)
    assert np.linalg.norm(A @ X - B) < 1000 * n * np.finfo(np.float64).eps * scale
    assert np.linalg.norm(X - Xref) / np.linalg.norm(Xref) < 1e-6   # cond(rand(n,n)) ~ n: far above what we need


@pytest.mark.parametrize("dtype,n,nrhs", [(np.float64, 1500, 64), (np.float64, 1500, 100), (np.float32, 1500, 64), (np.float32, 700, 200)])
def test_ldiv_block_of_right_hand_sides_matches_recursive_path(dtype, n, nrhs, monkeypatch):
    """ldiv!(F, B) with a block of right-hand sides (src/lu.jl:60-64, test/runtests.jl:116-128): the cooperative MFMA chain
    (RFLU_TRSM_CHAIN_MAX_RHS, default 320) against the recursive TRSM/GEMM splitting it replaces for 33 .. 320 columns -- same
    factors, same interchanges: solutions equal to rounding, residual inside the reference's bound."""
    eps = np.finfo(dtype).eps
    D = (rand_matrix(n, n, seed=610 + n, dtype=dtype) + dtype(10) * np.eye(n, dtype=dtype)).astype(dtype, order="F")
    B = rand_matrix(n, nrhs, seed=611 + n, dtype=dtype).copy(order="F")
    F = rf.lu_(to_dev_cm(D), None, True)
    X = to_dev_cm(B); rf.ldiv_(F, X)
    monkeypatch.setenv("RFLU_TRSM_CHAIN_MAX_RHS", "0")
    Y = to_dev_cm(B); rf.ldiv_(F, Y)
    x, y = X.cpu().numpy().astype(np.float64), Y.cpu().numpy().astype(np.float64)
    assert np.linalg.norm(D.astype(np.float64) @ x - B) < 1000 * n * eps * np.sqrt(nrhs)
    assert np.linalg.norm(x - y) <= 100 * eps * np.linalg.norm(y)


def test_row_major_device_entry():
    A = rand_matrix(700, 700, seed=21)
    d = torch.from_numpy(np.ascontiguousarray(A)).to("cuda:0")
    F = rf.lu_(d, None, True, check=False)
    check_against_oracle(A, F)


def test_wilkinson_ties_nan():
    # runtests.jl:130-140 generator: every search is an exact tie -> lowest index -> identity pivots, growth 2^(k-1)
    for n in (50, 130, 300):
        A = wilkinson(n)
        F = rf.lu(A, True, check=False)
        assert F.info == 0 and np.array_equal(F.ipiv, np.arange(1, n + 1))
        assert np.array_equal(np.asarray(F.factors)[:, -1], 2.0 ** np.arange(n))
    # small-integer matrix with many exact ties
    T = np.asfortranarray(np.floor(rand_matrix(300, 300, seed=413) * 4) - 1.5)
    F = rf.lu(T, True, check=False)
    assert np.array_equal(F.ipiv, O.lu(T)[1])
    # NaN never wins the argmax (src/lu.jl:298-304)
    N = rand_matrix(130, 130, seed=512)
    N[5, 0] = np.nan
    F = rf.lu(N, True, check=False)
    assert np.array_equal(F.ipiv, O.lu(N)[1])


GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_hip_path_reproduces_golden_fixtures(path):
    from golden.make_golden import build_input

    g = np.load(path, allow_pickle=False)
    A = build_input(g)
    pivot = bool(g["pivot"])
    F = rf.lu(A, pivot, check=False)
    assert abs(F.info) == int(g["info"])
    ip = np.asarray(F.ipiv)
    assert np.array_equal(ip, g["ipiv"])
    ok = np.isfinite(g["lu_sample"])
    got = np.asarray(F.factors).ravel(order="F")[g["sample_idx"]]
    tol = 64 * np.finfo(A.dtype).eps * max(A.shape) * max(1.0, float(np.max(np.abs(g["lu_sample"][ok]))))
    assert np.allclose(got[ok], g["lu_sample"][ok], atol=tol, rtol=0)


def test_large_properties_on_device():
    # BASELINE config sizes: size-independent properties computed on the GPU with torch (an independent checker):
    # ||PA - LU||_F / ||A||_F < 1e-12 and ipiv == LAPACK getrf (same comparator as the reference's tests)
    n = 4096
    A = rand_matrix(n, n, seed=12)
    dA = to_dev_cm(A)
    F = rf.lu_(dA, None, True, check=False)
    ip = F.ipiv.cpu().numpy()
    _, lpiv, linfo = sla.lapack.dgetrf(A)
    assert F.info == linfo == 0
    assert np.array_equal(ip, lpiv.astype(np.int64) + 1)
    LUm = F.factors
    L = torch.tril(LUm, -1) + torch.eye(n, dtype=LUm.dtype, device=LUm.device)
    U = torch.triu(LUm)
    perm = torch.from_numpy(O.oracle.perm_from_ipiv(ip, n)).to(LUm.device)
    PA = torch.from_numpy(A).to(LUm.device)[perm]
    res = (torch.linalg.norm(L @ U - PA) / torch.linalg.norm(PA)).item()
    assert res < 1e-12, res


@pytest.mark.parametrize("n,block,pivot", [(700, 128, True), (1000, 256, True), (300, 64, False)])
def test_block_column_driver_single_rank_hip_ops(n, block, pivot):
    # the multi-GPU driver with the real HIP building blocks (rflu_panel_rm with w > 64, laswp_rm, trsm_rm, gemm_rm),
    # world = 1 so no collective is needed: pivots bit-exact vs the oracle, factors to rounding
    from recursivefactorization.jl_amd import _ffi
    from recursivefactorization.jl_amd.distributed import BlockColumnLU, HipOps

    diag = 0.0 if pivot else 10.0
    h = _ffi.default_handle(0)
    h.set_stream(None)
    job = BlockColumnLU(HipOps(h, "f64"), n, torch.float64, 0, 1, torch.device("cuda:0"), block=block, pivot=pivot,
                        seed=12, diag_add=diag)
    job.regenerate()
    torch.cuda.synchronize()
    A = O.np_uniform(n, n, 12) + diag * np.eye(n)
    assert np.array_equal(job.R[:, :n].cpu().numpy(), A)
    info = job.factor()
    torch.cuda.synchronize()
    Fo, ipo, infoo = O.lu(A, pivot=pivot)
    assert info == infoo == 0
    assert np.array_equal(job.ipiv.cpu().numpy(), ipo)
    assert np.max(np.abs(job.gather_factors() - Fo)) < 50 * tol_E(A)
    assert job.matvec_residual() < 1e-12


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n", [1, 8, 64, 65, 200, 300, 1000])
def test_ldiv_solve_step(dtype, n):
    # runtests.jl:21-28 (b = ldiv!(MF, A[:, end]) ~ e_n) and :116-128 (NotIPIV ldiv!, ||Ax-b|| < 1000*n*eps on rand + 10I)
    eps = np.finfo(dtype).eps
    A = rand_matrix(n, n, seed=500 + n, dtype=dtype)
    F = rf.lu(A, True, check=False)
    b = rf.ldiv_(F, A[:, -1].copy())
    rhs = np.zeros(n); rhs[-1] = 1
    if np.all(np.isfinite(b)):
        assert np.allclose(b, rhs, atol=100 * 20 * n * eps * max(1.0, float(np.max(np.abs(F.factors)))), rtol=0)
    D = (rand_matrix(n, n, seed=600 + n, dtype=dtype) + dtype(10) * np.eye(n, dtype=dtype)).astype(dtype, order="F")
    for pivot in (rf.NoPivot(), rf.RowMaximum()):
        G = rf.lu(D, pivot)
        v = rand_matrix(n, 1, seed=700 + n, dtype=dtype)[:, 0].copy()
        x = rf.ldiv_(G, v.copy())
        assert x.dtype == dtype and np.linalg.norm(D.astype(np.float64) @ x - v) < 1000 * n * eps
        B3 = rand_matrix(n, 3, seed=800 + n, dtype=dtype).copy(order="F")
        X = rf.ldiv_(G, B3.copy(order="F"))
        assert np.linalg.norm(D.astype(np.float64) @ X - B3) < 1000 * n * eps
    # device-resident, column-major and row-major
    dF = rf.lu_(to_dev_cm(D), None, True)
    dB = to_dev_cm(B3)
    rf.ldiv_(dF, dB)
    assert np.linalg.norm(D.astype(np.float64) @ dB.cpu().numpy() - B3) < 1000 * n * eps
    rF = rf.lu_(torch.from_numpy(np.ascontiguousarray(D)).to("cuda:0"), None, True)
    rB = torch.from_numpy(np.ascontiguousarray(B3)).to("cuda:0")
    rf.ldiv_(rF, rB)
    assert np.linalg.norm(D.astype(np.float64) @ rB.cpu().numpy() - B3) < 1000 * n * eps
    S = D.copy(); S[:, n // 2] = 0
    with pytest.raises(rf.SingularException):
        rf.ldiv_(rf.lu(S, True, check=False), v.copy())
