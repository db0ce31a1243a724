"""-m gpu: the persistent update engine (csrc/engine.hip; RFLU_ENGINE=1 here -- by default it serves pivoted matrices of more than 11264
columns and the host entry: DESIGN.md section 3.5).  Every trailing update of the block columns with tall panels is pulled by resident workgroups from per-column-block
counters instead of being enqueued on the side / update streams; the eliminations and their order per column are those of the
stream schedules (src/lu.jl:189-263, :265-284), so pivots must be identical and factors equal to rounding."""
import numpy as np
import pytest
import torch

import recursivefactorization.jl_amd as rf
from gpu_util import fill_uniform_cm, matvec_residual

pytestmark = pytest.mark.gpu


def _factor(n, dtype, pivot, blocksize, m=None, diag_add=0.0):
    A = fill_uniform_cm(n, dtype, 12, diag_add, m=m)
    W = A.clone()
    F = rf.lu_(W, None, pivot, check=False, blocksize=blocksize)
    return A, F


@pytest.mark.parametrize("m,n,bs,dtype", [
    (6144, 6144, 256, np.float64),       # default width of this size
    (8192, 8192, 512, np.float64),
    (10000, 10000, 0, np.float64),       # last leaf partial
    (10000, 6144, 512, np.float64),      # tall: every block column through the engine
    (6144, 10240, 512, np.float64),      # fat (m a multiple of the block width)
    (12288, 12288, 0, np.float64),
])
def test_engine_matches_stream_schedule(m, n, bs, dtype, monkeypatch):
    monkeypatch.setenv("RFLU_ENGINE", "0")
    A, F = _factor(n, dtype, True, bs, m=m)
    monkeypatch.setenv("RFLU_ENGINE", "1")
    for ahead in ("1", "2"):   # leaf windows over the own block column + 1 (default) / 2 block columns right of it (engine.hpp: EngGeo::ahead)
        monkeypatch.setenv("RFLU_ENGINE_AHEAD", ahead)
        _, G = _factor(n, dtype, True, bs, m=m)
        assert F.info == G.info == 0
        assert rf.last_path() == "hip-engine"
        assert torch.equal(F.ipiv, G.ipiv)
        scale = float(F.factors.abs().max())
        assert float((F.factors - G.factors).abs().max()) <= 1e-10 * scale
    if m == n:
        assert matvec_residual(A, G.factors, G.ipiv) < 1e-12


def test_engine_nopivot_and_float32(monkeypatch):
    monkeypatch.setenv("RFLU_ENGINE", "1")
    n = 8192
    A, F = _factor(n, np.float64, rf.NoPivot(), 0, diag_add=10.0)
    assert F.info == 0
    assert matvec_residual(A, F.factors, np.arange(1, n + 1)) < 10 * np.sqrt(20 * n * np.finfo(np.float64).eps)
    # Float32: the pivot sequence may fork between two summation orders: info and the residual are pinned
    A, F = _factor(n, np.float32, True, 0)
    assert F.info == 0
    assert matvec_residual(A, F.factors, F.ipiv) < 20 * n * np.finfo(np.float32).eps


@pytest.mark.parametrize("n,rows", [(6144, "1024"), (8192, "4096"), (12288, "4096")])
def test_engine_hands_over_to_the_streams(n, rows, monkeypatch):
    """RFLU_ENGINE_ROWS > 0: the streams take over at the first panel of at most that many rows (default 0: the engine to the end, its
    workgroups on the chain's XCD retiring in front of the XCD-local short panels -- what test_engine_matches_stream_schedule runs)."""
    monkeypatch.setenv("RFLU_ENGINE", "0")
    A, F = _factor(n, np.float64, True, 0)
    monkeypatch.setenv("RFLU_ENGINE", "1")
    monkeypatch.setenv("RFLU_ENGINE_ROWS", rows)
    _, G = _factor(n, np.float64, True, 0)
    assert torch.equal(F.ipiv, G.ipiv)
    assert matvec_residual(A, G.factors, G.ipiv) < 1e-12


def test_engine_without_retirement(monkeypatch):
    """RFLU_ENGINE_RETIRE=0: the engine's workgroups stay everywhere to the end, every leaf any-placement."""
    monkeypatch.setenv("RFLU_ENGINE", "0")
    A, F = _factor(8192, np.float64, True, 0)
    monkeypatch.setenv("RFLU_ENGINE", "1")
    monkeypatch.setenv("RFLU_ENGINE_RETIRE", "0")
    _, G = _factor(8192, np.float64, True, 0)
    assert torch.equal(F.ipiv, G.ipiv)
    assert matvec_residual(A, G.factors, G.ipiv) < 1e-12


@pytest.mark.parametrize("m,n", [(12288, 12288), (12000, 11500), (13000, 13000), (16384, 12800), (16000, 15000)])
def test_default_rule_sends_these_shapes_through_the_engine(m, n, monkeypatch):
    """RFLU_ENGINE unset: pivoted matrices of more than 11264 columns (default block width, at most 16384 rows, not fat) are
    factored through the engine; pivots equal to the stream schedule's (RFLU_ENGINE=0), factors equal to rounding."""
    monkeypatch.setenv("RFLU_ENGINE", "0")
    A, F = _factor(n, np.float64, True, 0, m=m)
    assert rf.last_path() == "hip-lookahead"
    monkeypatch.delenv("RFLU_ENGINE")
    _, G = _factor(n, np.float64, True, 0, m=m)
    assert rf.last_path() == "hip-engine"
    assert F.info == G.info == 0
    assert torch.equal(F.ipiv, G.ipiv)
    scale = float(F.factors.abs().max())
    d = float((F.factors - G.factors).abs().max())
    assert d <= 1e-10 * scale
    if m == n:
        assert matvec_residual(A, G.factors, G.ipiv) < 1e-12


@pytest.mark.parametrize("m,n,bs", [(6144, 6144, 256), (8192, 8192, 512), (10000, 6144, 512)])
def test_engine_against_the_cpu_oracle(m, n, bs, monkeypatch):
    """The engine against the ORACLE (not against the stream schedules of the same library): `ipiv` and `info` bit-exact, factors within
    50 E max|LU| of the CPU restatement of src/lu.jl:189-338 on the same input (E = 20 s eps, test/runtests.jl:19-20), residual below E."""
    import os
    import oracle as O
    O.use_native()
    O.set_threads(min(64, os.cpu_count() or 1))
    try:
        A = O.fill_uniform(m, n, 40 + m + n, np.float64)
        Fo, ipo, info_o = O.lu(A)
    finally:
        O.set_threads(1)
    assert info_o == 0
    monkeypatch.setenv("RFLU_ENGINE", "1")
    W = torch.from_numpy(np.ascontiguousarray(A.T)).to("cuda:0").T      # column-major device view
    F = rf.lu_(W, None, True, check=False, blocksize=bs)
    assert rf.last_path() == "hip-engine" and F.info == 0
    ip = F.ipiv.cpu().numpy()
    assert np.array_equal(ip, ipo), f"first difference from the CPU path at pivot {int(np.argmax(ip != ipo))}"
    got = F.factors.cpu().numpy()
    E = 20 * min(m, n) * np.finfo(np.float64).eps
    assert float(np.abs(got - Fo).max()) <= 50 * E * max(1.0, float(np.abs(Fo).max()))
    if m == n:
        dA = torch.from_numpy(np.ascontiguousarray(A.T)).to("cuda:0").T   # (W was factored in place)
        assert matvec_residual(dA, F.factors, F.ipiv) < 1e-12


def test_default_rule_float32_headline_size_goes_through_the_engine(monkeypatch):
    """Round 6: Float32 pivoted matrices of more than 11264 columns take the engine by default too (N=16384: 55.3 vs 58.8 ms).  Float32 is held
    to what the reference holds it to -- info, the residual bound (test/runtests.jl:19-20) -- plus a floor of leading pivots equal to the
    stream schedule's: past a near-tie two valid summation orders may choose different pivots."""
    n = 16384
    monkeypatch.setenv("RFLU_ENGINE", "0")
    A, F = _factor(n, np.float32, True, 0)
    assert rf.last_path() == "hip-lookahead" and F.info == 0
    monkeypatch.delenv("RFLU_ENGINE")
    _, G = _factor(n, np.float32, True, 0)
    assert rf.last_path() == "hip-engine" and G.info == 0
    same = (F.ipiv == G.ipiv).cpu().numpy()
    lead = int(np.argmin(same)) if not same.all() else n
    assert lead >= 1024, lead
    assert matvec_residual(A, G.factors, G.ipiv) < 20 * n * np.finfo(np.float32).eps
