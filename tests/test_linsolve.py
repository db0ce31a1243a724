"""LinearSolve.jl's RFLUFactorization cache protocol (the operator surface README.md:36-37 of the reference
names, call shape src/lu.jl:97-130 with check = false), executed.

-m gpu: the protocol of recursivefactorization.jl_amd/linsolve.py over the HIP path, checked against the CPU oracle:
  * fresh cache -> lu! INTO the cached ipiv (pivots bit-exact vs the oracle), ReturnCode.Success, x within the reference's own
    solve bounds (test/runtests.jl:21-28);
  * a second solve with a new b REUSES the factors: the library's per-kernel-class launch counters show no new panel launch;
  * assigning A marks the cache fresh -> exactly one more factorization;
  * a singular A -> ReturnCode.Failure (no exception: check = false + issuccess), info = the zero column, u untouched;
  * NoPivot algorithm on rand + 10I: the cached ipiv is filled with 1:n (src/lu.jl:111-113).
not gpu: the same state machine with the oracle standing in for lu_ / ldiv_ (host logic only)."""
import numpy as np
import pytest

import oracle as O
from helpers import rand_matrix


def _oracle_solve(A, b):
    F, ip, info = O.lu(A)
    assert info == 0
    L, U = O.unpack_lu(F)
    pb = b[O.perm_from_ipiv(ip, A.shape[0])]
    y = np.linalg.solve(L, pb)
    return np.linalg.solve(U, y), ip


@pytest.mark.gpu
@pytest.mark.parametrize("where", ["device", "host"])
def test_rflu_factorization_cache_protocol(where):
    import torch

    import recursivefactorization.jl_amd as rf
    from recursivefactorization.jl_amd import _ffi
    from recursivefactorization.jl_amd import linsolve as LS

    n = 1500   # > 1024 columns: the block-column schedule, like a LinearSolve user at GPU sizes
    A0 = rand_matrix(n, n, seed=31)
    b1 = rand_matrix(n, 1, seed=32)[:, 0].copy()
    b2 = rand_matrix(n, 1, seed=33)[:, 0].copy()
    x1_ref, ip_ref = _oracle_solve(A0, b1)
    x2_ref, _ = _oracle_solve(A0, b2)

    def put(M):
        if where == "host":
            return np.array(M, order="F", copy=True)
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(M).T)).to("cuda:0")
        return t.T if M.ndim == 2 else t

    def get(t):
        return t.cpu().numpy() if hasattr(t, "cpu") else np.asarray(t)

    h = _ffi.default_handle(0)
    h.profile_enable(True)   # per-kernel-class launch counters
    try:
        cache = LS.init(put(A0), put(b1), LS.RFLUFactorization(pivot=rf.Val(True), thread=rf.Val(False)))
        fact0, ipiv0 = cache.cacheval
        assert cache.isfresh and fact0.factors.shape == (0, 0)
        base = h.profile()["panel"]["launches"]
        sol = LS.solve_(cache)
        assert sol.retcode is LS.ReturnCode.Success and not cache.isfresh and cache.nfactor == 1
        assert cache.cacheval[1] is ipiv0, "lu! must write into the ipiv the cache allocated"
        assert np.array_equal(get(cache.cacheval[1]), ip_ref), "ipiv differs from the CPU oracle"
        p1 = h.profile()["panel"]["launches"]
        assert p1 > base
        x1 = get(sol.u)
        E = 20 * n * np.finfo(np.float64).eps
        assert np.max(np.abs(x1 - x1_ref)) < 100 * E * max(1.0, np.max(np.abs(x1_ref)) * np.linalg.cond(A0, 1) ** 0.5)
        assert np.linalg.norm(A0 @ x1 - b1) / (np.linalg.norm(A0) * np.linalg.norm(x1)) < 1e-13
        # new right-hand side, same A: reuse (no panel kernel may run)
        cache.b = put(b2)
        sol = LS.solve_(cache)
        assert sol.retcode is LS.ReturnCode.Success and cache.nfactor == 1
        assert h.profile()["panel"]["launches"] == p1, "a cached factorization must not be recomputed"
        x2 = get(sol.u)
        assert np.linalg.norm(A0 @ x2 - b2) / (np.linalg.norm(A0) * np.linalg.norm(x2)) < 1e-13
        assert np.allclose(x2, x2_ref, rtol=1e-6, atol=1e-8)
        # new A: fresh again -> exactly one more factorization, into the same ipiv
        A1 = rand_matrix(n, n, seed=34)
        cache.A = put(A1)
        assert cache.isfresh
        sol = LS.solve_(cache)
        assert sol.retcode is LS.ReturnCode.Success and cache.nfactor == 2 and cache.cacheval[1] is ipiv0
        assert h.profile()["panel"]["launches"] > p1
        assert np.array_equal(get(cache.cacheval[1]), O.lu(A1)[1])
        x3 = get(sol.u)
        assert np.linalg.norm(A1 @ x3 - b2) / (np.linalg.norm(A1) * np.linalg.norm(x3)) < 1e-13
        # singular A: Failure as a return code (check = false), info like LAPACK's, nothing raised, u left alone
        S = rand_matrix(n, n, seed=35)
        S[:, 777] = 0.0
        cache.A = put(S)
        u_before = get(cache.u).copy()
        sol = LS.solve_(cache)
        assert sol.retcode is LS.ReturnCode.Failure
        assert cache.cacheval[0].info == 778 == O.lu(S)[2] and not cache.cacheval[0].issuccess()
        assert cache.isfresh, "a failed factorization leaves the cache fresh (LinearSolve returns before clearing the flag)"
        assert np.array_equal(get(cache.u), u_before)
    finally:
        h.profile_enable(False)


@pytest.mark.gpu
def test_rflu_factorization_nopivot_and_one_shot_solve():
    import torch

    import recursivefactorization.jl_amd as rf
    from recursivefactorization.jl_amd import linsolve as LS

    n = 700
    D = rand_matrix(n, n, seed=41) + 10 * np.eye(n)          # test/runtests.jl:75
    b = rand_matrix(n, 1, seed=42)[:, 0].copy()
    dA = torch.from_numpy(np.ascontiguousarray(D.T)).to("cuda:0").T
    db = torch.from_numpy(b).to("cuda:0")
    cache = LS.init(dA.clone(), db, LS.RFLUFactorization(pivot=rf.NoPivot()))
    cache.cacheval[1].fill_(-7)                               # poison, as test/runtests.jl:70-84 does
    sol = LS.solve_(cache)
    assert sol.retcode is LS.ReturnCode.Success
    assert np.array_equal(cache.cacheval[1].cpu().numpy(), np.arange(1, n + 1)), "NoPivot fills the caller's ipiv with 1:n"
    x = sol.u.cpu().numpy()
    assert np.linalg.norm(D @ x - b) / (np.linalg.norm(D) * np.linalg.norm(x)) < 1e-12
    # out-of-place one-shot: A and b untouched
    keep = dA.clone()
    sol = LS.solve(dA, db)
    assert sol.retcode is LS.ReturnCode.Success and torch.equal(dA, keep)
    x = sol.u.cpu().numpy()
    assert np.linalg.norm(D @ x - b) / (np.linalg.norm(D) * np.linalg.norm(x)) < 1e-13


def test_cache_state_machine_with_oracle_standins(monkeypatch):
    """No GPU: the protocol's control flow (fresh / reuse / refactor / Failure) with the CPU oracle behind lu_ and ldiv_."""
    from recursivefactorization.jl_amd import linsolve as LS
    from recursivefactorization.jl_amd.lu import LU

    calls = {"lu": 0, "ldiv": 0}

    def fake_lu_(A, ipiv, pivot, thread, *, check, blocksize):
        assert check is False, "LinearSolve calls lu! with check = false"
        calls["lu"] += 1
        F, ip, info = O.lu(np.asarray(A), pivot=bool(LS.normalize_pivot(pivot)))
        A[...] = F
        ipiv[...] = ip
        return LU(A, ipiv, info)

    def fake_ldiv_(F, B):
        calls["ldiv"] += 1
        L, U = O.unpack_lu(np.asarray(F.factors))
        B[...] = np.linalg.solve(U, np.linalg.solve(L, B[O.perm_from_ipiv(np.asarray(F.ipiv), B.shape[0])]))
        return B

    monkeypatch.setattr(LS, "lu_", fake_lu_)
    monkeypatch.setattr(LS, "ldiv_", fake_ldiv_)
    n = 60
    A = rand_matrix(n, n, seed=1)
    b = rand_matrix(n, 1, seed=2)[:, 0].copy()
    cache = LS.init(np.array(A, order="F"), b)
    assert cache.isfresh and len(cache.cacheval[1]) == n
    s = LS.solve_(cache)
    assert s.retcode is LS.ReturnCode.Success and calls == {"lu": 1, "ldiv": 1} and not cache.isfresh
    assert np.allclose(A @ s.u, b)
    cache.b = 2 * b
    s = LS.solve_(cache)
    assert calls == {"lu": 1, "ldiv": 2} and np.allclose(A @ s.u, 2 * b)
    cache.A = np.array(A.T, order="F")
    s = LS.solve_(cache)
    assert calls == {"lu": 2, "ldiv": 3} and np.allclose(A.T @ s.u, 2 * b)
    S = A.copy()
    S[:, 5] = 0
    cache.A = np.array(S, order="F")
    s = LS.solve_(cache)
    assert s.retcode is LS.ReturnCode.Failure and calls == {"lu": 3, "ldiv": 3} and cache.cacheval[0].info == 6 and cache.isfresh
    # a cache whose ipiv has the wrong length gets a new one (LinearSolve resizes)
    cache2 = LS.init(np.array(A[:, :40], order="F"), b)
    assert len(cache2.cacheval[1]) == 40
