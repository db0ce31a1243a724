"""-m gpu: every BASELINE.json configuration at its FULL size on one MI355X, through the reference-shaped API, checked by
size-independent properties (the n^3 product and the CPU oracle are not affordable here):
  * info == 0 (test/runtests.jl:15 compares info with LAPACK's, which is 0 for these full-rank inputs),
  * max over random x of ||P*A*x - L*(U*x)|| / ||A*x||  (O(n^2), plain torch ops as an independent checker) below the
    north star's Float64 bar 1e-12 (Float32 / NoPivot: the reference's own looser bounds, test/runtests.jl:19-20, 124-126),
  * ipiv identical between the outer-block widths, including blocksize = -1 (the pure Toledo recursion, the reference's own
    structure src/lu.jl:189-263): the pivot sequence must not depend on the schedule.
Configs: (1) N=4096; (2) N=16384 with the block-size sweep 64/128/256 and the default; (3) N=32768; (4) N=65536 Float64,
Float32 and NoPivot (on `rand + 10I` as the reference's NoPivot tests do, test/runtests.jl:75,94,118, and on the plain
uniform matrix, whose residual is only reported).  The 2/4/8-GPU layouts of configs 3-4 need more than one GPU: their
partition logic is covered by tests/test_distributed.py (gloo) and tests/test_gpu_multiproc.py."""
import numpy as np
import pytest
import torch

import recursivefactorization.jl_amd as rf
from gpu_util import fill_uniform_cm, matvec_residual

pytestmark = pytest.mark.gpu


def _factor(n, dtype, pivot, blocksize, diag_add=0.0, seed=12):
    A = fill_uniform_cm(n, dtype, seed, diag_add)
    W = A.clone()
    F = rf.lu_(W, None, pivot, check=False, blocksize=blocksize)
    return A, F


def _free(*ts):
    for t in ts:
        del t
    torch.cuda.empty_cache()


def _cpu_pivots_of_left_panel(n, k, seed=12):
    """The first k pivots of a factorization depend on the first k columns only: `ipiv` of the CPU path (the threaded oracle,
    src/lu.jl:298-305 restated) and of LAPACK dgetrf on the tall n x k left panel of the SAME matrix -- the generator's counter
    is j*m + i, so `fill_uniform(n, k, seed)` is exactly the first k columns of `fill_uniform(n, n, seed)`."""
    import os
    import oracle as O
    O.use_native()
    O.set_threads(min(64, os.cpu_count() or 1))
    try:
        A = O.fill_uniform(n, k, seed)
        _, ipo, info_o = O.lu(A)
    finally:
        O.set_threads(1)
    assert info_o == 0
    piv_lapack = None
    try:
        import scipy.linalg as sla
        _, piv, info_l = sla.lapack.dgetrf(A, overwrite_a=True)
        assert info_l == 0
        piv_lapack = piv.astype(np.int64) + 1
    except ImportError:
        pass
    return ipo, piv_lapack


def _assert_leading_pivots_equal_cpu_path(dA, ip_gpu, n, k=4096):
    ipo, piv_lapack = _cpu_pivots_of_left_panel(n, k)
    # same input on both sides: a strided sample of the device matrix's left panel against the host generator
    idx = torch.arange(0, n, 1021, device=dA.device)
    jdx = torch.arange(0, k, 127, device=dA.device)
    assert np.array_equal(dA[idx][:, jdx].cpu().numpy(), _sample_uniform(n, 12, np.arange(0, n, 1021), np.arange(0, k, 127)))
    ip = ip_gpu[:k].cpu().numpy()
    assert np.array_equal(ip, ipo), f"first difference from the CPU path at pivot {int(np.argmax(ip != ipo))}"
    if piv_lapack is not None:
        assert np.array_equal(ip, piv_lapack), f"first difference from dgetrf at pivot {int(np.argmax(ip != piv_lapack))}"


def _sample_uniform(m, seed, rows, cols):
    """entries (rows x cols) of the m x m generator matrix without building it (oracle.np_uniform's arithmetic)"""
    with np.errstate(over="ignore"):
        ctr = cols.astype(np.uint64)[None, :] * np.uint64(m) + rows.astype(np.uint64)[:, None]
        z = np.uint64(seed) + (ctr + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


@pytest.mark.parametrize("blocksize", [64, 128, 256, 0])
def test_config2_n16384_block_size_sweep(blocksize, record_property):
    n = 16384
    A, F = _factor(n, np.float64, True, blocksize)
    assert F.info == 0
    assert rf.last_path() == ("hip-engine" if blocksize == 0 else "hip-lookahead")   # (default width: the update engine)
    res = matvec_residual(A, F.factors, F.ipiv)
    record_property("residual", res)
    assert res < 1e-12, res
    # the pivot sequence is a property of the matrix, not of the schedule: compare with the pure recursion
    Aref, G = _factor(n, np.float64, True, -1)
    assert rf.last_path() == "hip-recursive"
    assert torch.equal(F.ipiv, G.ipiv)
    _free(A, Aref)


def test_config1_n4096(record_property):
    A, F = _factor(4096, np.float64, True, 0)
    assert F.info == 0
    res = matvec_residual(A, F.factors, F.ipiv)
    assert res < 1e-12, res
    _, G = _factor(4096, np.float64, True, -1)
    assert torch.equal(F.ipiv, G.ipiv)


def test_config3_n32768(record_property):
    n = 32768
    A, F = _factor(n, np.float64, True, 0)
    assert F.info == 0
    res = matvec_residual(A, F.factors, F.ipiv, chunk=8192)
    record_property("residual", res)
    assert res < 1e-12, res
    ip = F.ipiv.clone()
    # the first 4096 pivots against the CPU path (threaded oracle and dgetrf on the tall left panel): the 64-workgroup leaves
    _assert_leading_pivots_equal_cpu_path(A, ip, n)
    _free(A, F)
    _, G = _factor(n, np.float64, True, 512)   # the multi-GPU layout's block width
    assert torch.equal(ip, G.ipiv)


def test_config4_n65536_float64(record_property):
    n = 65536
    A, F = _factor(n, np.float64, True, 0)
    assert F.info == 0
    res = matvec_residual(A, F.factors, F.ipiv, chunk=8192, trials=1)
    record_property("residual", res)
    assert res < 1e-12, res
    ip = F.ipiv.clone()
    # the first 4096 pivots against the CPU path: 128-workgroup leaves and the single-stream branch for panels of more than
    # 32768 rows (src/lu.jl:298-305 on 65536 rows; ~1.1 TFLOP of host work)
    _assert_leading_pivots_equal_cpu_path(A, ip, n)
    _free(A, F)
    _, G = _factor(n, np.float64, True, 1024)
    assert torch.equal(ip, G.ipiv)


def test_config4_n65536_float32(record_property):
    # Float32 at this size: the pivot sequence may legitimately fork between summation orders, so only
    # info and the residual are pinned; E = 20*n*eps as in test/runtests.jl:19 (max|L| <= 1 keeps the growth in check)
    n = 65536
    A, F = _factor(n, np.float32, True, 0)
    assert F.info == 0
    res = matvec_residual(A, F.factors, F.ipiv, chunk=8192, trials=1)
    record_property("residual", res)
    assert res < 20 * n * np.finfo(np.float32).eps, res
    _free(A, F)


def test_config4_n65536_nopivot(record_property):
    n = 65536
    # the reference's own NoPivot inputs are diagonally shifted (test/runtests.jl:75,94,118): meaningful residual
    A, F = _factor(n, np.float64, rf.NoPivot(), 0, diag_add=10.0)
    assert F.info == 0 and isinstance(F.ipiv, rf.NotIPIV)
    res = matvec_residual(A, F.factors, np.arange(1, n + 1), chunk=8192, trials=1)
    record_property("residual_rand_plus_10I", res)
    assert res < 10 * np.sqrt(20 * n * np.finfo(np.float64).eps), res   # test/runtests.jl:20 (unpivoted bound)
    _free(A, F)
    # plain uniform input without pivoting: element growth is unbounded, BASELINE only asks for the residual to be REPORTED
    A, F = _factor(n, np.float64, rf.NoPivot(), 0)
    res = matvec_residual(A, F.factors, np.arange(1, n + 1), chunk=8192, trials=1)
    record_property("residual_plain_uniform", res)
    print(f"N=65536 NoPivot on plain uniform input: info={F.info}, matvec residual {res:.3e} (reported, not bounded)")
    assert isinstance(F.info, int)
    _free(A, F)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_n16384_float32_and_nopivot_variants(dtype, record_property):
    n = 16384
    A, F = _factor(n, dtype, rf.NoPivot(), 0, diag_add=10.0)
    assert F.info == 0
    res = matvec_residual(A, F.factors, np.arange(1, n + 1))
    assert res < 10 * np.sqrt(20 * n * np.finfo(dtype).eps), res
    _free(A, F)
    if dtype == np.float32:
        A, F = _factor(n, dtype, True, 0)
        assert F.info == 0
        res = matvec_residual(A, F.factors, F.ipiv)
        assert res < 20 * n * np.finfo(np.float32).eps, res
        _free(A, F)


def test_headline_n16384_ipiv_equals_cpu_path(record_property):
    """The north star's bar on the configuration the headline is quoted on: `ipiv` of the N=16384 Float64 factorization equal,
    entry by entry, to the CPU path's on the same input (src/lu.jl:298-305 picks the pivots; the oracle restates it, threaded
    as `thread = Val(true)` does, results independent of the thread count) -- and to LAPACK's dgetrf, the comparator of the
    reference's own tests (test/runtests.jl:11,52).  About a minute of host time on the GPU box."""
    import os
    import oracle as O
    n = 16384
    O.use_native()
    O.set_threads(min(64, os.cpu_count() or 1))
    try:
        A = O.fill_uniform(n, n, 12)                      # the generator the device fill mirrors bit for bit
        _, ipo, info_o = O.lu(A)
    finally:
        O.set_threads(1)
    assert info_o == 0
    dA, F = _factor(n, np.float64, True, 0, seed=12)
    assert F.info == 0 and rf.last_path() == "hip-engine"   # the shipped schedule of the headline size
    # same input on both sides: spot-check a strided sample of the device matrix against the host one
    idx = torch.arange(0, n, 257, device=dA.device)
    assert np.array_equal(dA[idx][:, idx].cpu().numpy(), A[::257, ::257])
    ip = F.ipiv.cpu().numpy()
    assert np.array_equal(ip, ipo), f"first difference at pivot {int(np.argmax(ip != ipo))}"
    res = matvec_residual(dA, F.factors, F.ipiv)
    record_property("residual", res)
    assert res < 1e-12, res
    try:
        import scipy.linalg as sla
        _, piv, info_l = sla.lapack.dgetrf(A, overwrite_a=True)
        assert info_l == 0
        assert np.array_equal(ip, piv.astype(np.int64) + 1), "ipiv differs from LAPACK dgetrf"
    except ImportError:
        pass
    _free(dA, F)


def test_headline_size_float32_info_and_residual_vs_cpu_path(record_property):
    """N=16384 Float32 against the CPU path: `info` equal and both residuals inside the reference's bound E = 20 n eps
    (test/runtests.jl:19-20).  Float32 pivot sequences may fork between two equally valid summation orders from n ~ 1500 on
    (DESIGN.md section 5), so the number of equal leading pivots is reported and only a floor (1024) is asserted."""
    import os
    import oracle as O
    n = 16384
    O.use_native()
    O.set_threads(min(64, os.cpu_count() or 1))
    try:
        A = O.fill_uniform(n, n, 12, np.float32)
        Fo, ipo, info_o = O.lu(A)
    finally:
        O.set_threads(1)
    dA, F = _factor(n, np.float32, True, 0, seed=12)
    assert F.info == info_o == 0
    E = 20 * n * np.finfo(np.float32).eps
    res = matvec_residual(dA, F.factors, F.ipiv)
    dFo = torch.from_numpy(np.ascontiguousarray(Fo.T)).to(dA.device).T
    res_cpu = matvec_residual(dA, dFo, torch.from_numpy(ipo))
    record_property("residual_gpu", res)
    record_property("residual_cpu_path", res_cpu)
    ip = F.ipiv.cpu().numpy()
    same = int(np.argmax(ip != ipo)) if (ip != ipo).any() else n
    record_property("equal_leading_pivots", same)
    print(f"N=16384 Float32: residual GPU {res:.3e}, CPU path {res_cpu:.3e}, equal leading pivots {same} of {n}")
    assert res < E and res_cpu < E, (res, res_cpu)
    assert res < 4 * res_cpu + 1e-6
    # forks are legitimate from n ~ 1500 on (two candidates closer than the rounding error of two summation orders), but a broken
    # Float32 search would fork at once: the leading pivots have to agree
    assert same >= 1024, same
    _free(dA, F, dFo)


@pytest.mark.parametrize("n,bs", [(2048, 128), (3000, 256), (4096, 512), (12288, 512)])
def test_leafwise_schedule_matches_default(n, bs, monkeypatch):
    """The leaf-wise schedule (driver.cpp: factor_leafwise, the default for panels of at most 8192 rows) applies the same
    eliminations in the same order as the block-column lookahead schedule it replaces (RFLU_LEAFWISE=0): identical pivots,
    factors equal to rounding, residual below the 1e-12 bar."""
    monkeypatch.setenv("RFLU_LEAFWISE", "0")
    A, F = _factor(n, np.float64, True, bs)
    monkeypatch.delenv("RFLU_LEAFWISE", raising=False)
    _, G = _factor(n, np.float64, True, bs)
    assert F.info == 0 and G.info == 0
    assert rf.last_path() == "hip-lookahead"
    assert torch.equal(F.ipiv, G.ipiv)
    scale = float(F.factors.abs().max())
    assert float((F.factors - G.factors).abs().max()) <= 1e-10 * scale
    assert matvec_residual(A, G.factors, G.ipiv) < 1e-12


@pytest.mark.parametrize("n,dtype", [(4096, np.float64), (3000, np.float64), (6144, np.float64), (4096, np.float32)])
def test_xcd_local_short_panels_are_bit_identical(n, dtype, monkeypatch):
    """Panels of at most 4096 rows run with all participants on one XCD and plain-store records (panel.hip: RFLU_PANEL_LOCAL_ROWS);
    taller ones, and everything with RFLU_PANEL_LOCAL_ROWS=0, with write-through records on any placement.  Same arithmetic, same
    order: identical factors and pivots."""
    monkeypatch.setenv("RFLU_PANEL_LOCAL_ROWS", "0")
    A, F = _factor(n, dtype, True, 0)
    monkeypatch.delenv("RFLU_PANEL_LOCAL_ROWS")
    for _ in range(2):
        _, G = _factor(n, dtype, True, 0)
        assert F.info == G.info == 0
        assert torch.equal(F.ipiv, G.ipiv)
        assert torch.equal(F.factors, G.factors)


import os as _os
_EXPERIMENT_BUILD = _os.environ.get("RFLU_LIB", "").endswith("librflu_exp.so")
_needs_experiments = pytest.mark.skipif(not _EXPERIMENT_BUILD, reason="the sub-panel leaf lives in the experiments build only: "
                                        "RFLU_EXPERIMENTS=1 python recursivefactorization.jl_amd/build.py; RFLU_LIB=.../librflu_exp.so")


@_needs_experiments
@pytest.mark.parametrize("n,dtype", [(4096, np.float64), (3000, np.float64), (9000, np.float64), (4096, np.float32), (1100, np.float32)])
def test_subpanel_leaf_is_bit_identical(n, dtype, monkeypatch):
    """RFLU_PANEL_BLOCKED=1 routes every full pivoted leaf to the sub-panel kernel (panel_blocked.hip: one chain wave per workgroup
    carries the pivot search of 8 columns at a time, the other waves follow through LDS counters).  Every entry receives the
    multiply-adds of the unblocked algorithm (src/lu.jl:290-338) with the same operands in the same order: identical factors and
    pivots, inside the block-column schedules (offsets, XCD-local and any-placement records, one workgroup and many)."""
    _, F = _factor(n, dtype, True, 0)
    monkeypatch.setenv("RFLU_PANEL_BLOCKED", "1")
    for _ in range(2):
        _, G = _factor(n, dtype, True, 0)
        assert F.info == G.info == 0
        assert torch.equal(F.ipiv, G.ipiv)
        assert torch.equal(F.factors, G.factors)


@_needs_experiments
@pytest.mark.parametrize("kind", ["ties", "zero_column", "nan", "singular"])
def test_subpanel_leaf_special_values(kind, monkeypatch):
    """The sub-panel kernel's general search path: exact ties (lowest position wins), an all-zero column (its first row is the
    pivot, info reports it), NaN entries (never chosen while a number is there) -- the same pivots, info and bits as the default
    leaves on the same input."""
    rng = np.random.default_rng(7)
    m = 1500
    if kind == "ties":
        A = rng.integers(-3, 4, size=(m, 128)).astype(np.float64)
    else:
        A = rng.random((m, 128))
    if kind == "zero_column":
        A[:, 70] = 0.0
    if kind == "nan":
        A[5, 3] = np.nan
        A[900, 64] = np.nan
    if kind == "singular":
        A[:, 10] = A[:, 9]
    out = []
    for blocked in ("0", "1"):
        monkeypatch.setenv("RFLU_PANEL_BLOCKED", blocked)
        W = torch.from_numpy(np.ascontiguousarray(A.T)).to("cuda:0").T
        F = rf.lu_(W, None, True, check=False)
        out.append((F.info, F.ipiv.clone(), F.factors.clone()))
    assert out[0][0] == out[1][0]
    assert torch.equal(out[0][1], out[1][1])
    a, b = out[0][2], out[1][2]
    assert torch.equal(torch.isnan(a), torch.isnan(b))
    assert torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))


@pytest.mark.parametrize("m,n,bs,dtype,pivot", [
    (3000, 2048, 128, np.float64, True),     # tall: panels of 3000 .. 952 rows
    (2048, 3000, 256, np.float64, True),     # fat: the windows of the last block column reach into the tail (src/lu.jl:148-154)
    (2500, 2500, 192, np.float64, True),     # block width not a power of two, last leaf partial (2500 = 39 * 64 + 4)
    (3072, 3072, 256, np.float32, True),
    (2048, 2048, 256, np.float64, False),    # NoPivot: gates as separate one-wave kernels (no interchange launch to ride on)
])
def test_leafwise_schedule_rectangular_and_modes(m, n, bs, dtype, pivot, monkeypatch):
    """Leaf-wise schedule against the CPU oracle on shapes that stress its window arithmetic: pivots bit-exact (Float64),
    factors within the reference's own bound (test/runtests.jl:19-20)."""
    from oracle import oracle as O
    diag = 10.0 if not pivot else 0.0
    A = O.fill_uniform(m, n, 31 + m + n, dtype)
    if diag:
        A[np.arange(min(m, n)), np.arange(min(m, n))] += diag
    monkeypatch.delenv("RFLU_LEAFWISE", raising=False)
    W = torch.from_numpy(np.ascontiguousarray(A.T)).to("cuda:0").T      # column-major device view
    F = rf.lu_(W, None, pivot, check=False, blocksize=bs)
    assert rf.last_path() == "hip-lookahead" and F.info == 0
    Fo, ipo, info_o = O.lu(A, pivot=pivot)
    assert info_o == 0
    eps = np.finfo(dtype).eps
    E = 20 * min(m, n) * eps
    ip = F.ipiv.cpu().numpy() if pivot else np.arange(1, min(m, n) + 1)     # NoPivot with ipiv = None returns NotIPIV
    if dtype == np.float64 or not pivot:
        assert np.array_equal(ip, ipo)
    got = F.factors.cpu().numpy()
    tol = (10 * np.sqrt(E) if not pivot else 50 * E) * max(1.0, float(np.abs(Fo).max()))
    if dtype == np.float64 or not pivot:
        assert float(np.abs(got - Fo).max()) <= tol
    res, rel = O.residual(A, got, ip)
    assert res <= (E if pivot else 10 * np.sqrt(E)) * max(1.0, float(np.abs(A).max())) * 4
