"""LinearSolve.jl's ``RFLUFactorization`` cache protocol, executed -- the operator surface README.md:36-37 of the reference
names, on the host side of this package's C ABI.

LinearSolve's ``solve!(cache, ::RFLUFactorization{P,T})`` [external to /root/reference; the call shape it uses is
src/lu.jl:97-130 with ``check = false``] does, in order:

    fact, ipiv = cacheval
    if cache.isfresh:                                   # A was (re)assigned since the last factorization
        resize ipiv to min(size(A)...) if needed
        fact = RecursiveFactorization.lu!(A, ipiv, Val(P), Val(T), check = false)
        cache.cacheval = (fact, ipiv)
        !issuccess(fact) -> return ReturnCode.Failure   # a zero pivot is a return code here, never an exception
        cache.isfresh = false
    y = ldiv!(cache.u, fact, cache.b)                   # u <- A \ b with the cached factors
    return ReturnCode.Success

``julia/RFLUAMD/ext/RFLUAMDLinearSolveExt.jl`` is that protocol in Julia (not executable in this image); this module is the
same state machine in the language the tests here can run, over the same entry points (``lu_`` -> ``rflu_getrf_*``,
``ldiv_`` -> ``rflu_getrs_*``): fresh -> factor INTO the cached ``ipiv``, not fresh -> reuse the factors without touching
the GPU factorization again, singular -> ``ReturnCode.Failure``.
"""
from __future__ import annotations

import enum
from dataclasses import dataclass, field

import numpy as np

from .lu import LU, NotIPIV, _is_torch, ldiv_, lu_, normalize_pivot


class ReturnCode(enum.Enum):
    """SciMLBase.ReturnCode, the two values the LU algorithms return."""

    Success = 1
    Failure = 2


@dataclass
class RFLUFactorization:
    """``RFLUFactorization{P,T}(; pivot = Val(true), thread = Val(true))``: P = pivoting, T = threading (ignored on the GPU)."""

    pivot: object = True
    thread: object = True
    blocksize: int = 0

    @property
    def P(self) -> bool:
        return normalize_pivot(self.pivot)


@dataclass
class LinearSolution:
    u: object
    retcode: ReturnCode
    alg: RFLUFactorization


def _new_ipiv(A, n):
    if _is_torch(A):
        import torch

        return torch.empty(n, dtype=torch.int64, device=A.device)
    return np.empty(n, dtype=np.int64)


def _length(ipiv) -> int:
    return int(ipiv.numel()) if _is_torch(ipiv) else int(len(ipiv))


@dataclass
class LinearCache:
    """The fields of ``LinearSolve.LinearCache`` the LU algorithms use.  Assigning ``A`` marks the cache fresh (LinearSolve's
    ``setproperty!``); assigning ``b`` does not."""

    _A: object
    b: object
    u: object
    alg: RFLUFactorization
    cacheval: tuple = None
    isfresh: bool = True
    nfactor: int = field(default=0)   # factorizations performed (tests: reuse must not increase it)

    @property
    def A(self):
        return self._A

    @A.setter
    def A(self, value):
        self._A = value
        self.isfresh = True


def init_cacheval(alg: RFLUFactorization, A, b, u):
    """``init_cacheval(::RFLUFactorization, A, b, u, ...)`` -> ``(fact, ipiv)``: a placeholder factorization of the right type and
    the pivot vector that every later ``lu!`` writes into (LinearSolve allocates it once: ``Vector{BlasInt}(undef, min(size(A)...))``)."""
    ipiv = _new_ipiv(A, min(int(A.shape[0]), int(A.shape[1])))
    return (LU(A[:0, :0], ipiv[:0], 0), ipiv)


def init(A, b, alg: RFLUFactorization = None, u=None) -> LinearCache:
    """``init(LinearProblem(A, b), alg)``: ``A`` is used IN PLACE by the factorization (LinearSolve's ``alias_A``), ``u`` receives
    the solution."""
    alg = alg or RFLUFactorization()
    if u is None:
        u = b.clone() if _is_torch(b) else np.array(b, copy=True)
    cache = LinearCache(A, b, u, alg)
    cache.cacheval = init_cacheval(alg, A, b, u)
    return cache


def solve_(cache: LinearCache) -> LinearSolution:
    """``solve!(cache)`` for ``RFLUFactorization`` (see the module docstring)."""
    alg = cache.alg
    A = cache.A
    fact, ipiv = cache.cacheval
    if cache.isfresh:
        mn = min(int(A.shape[0]), int(A.shape[1]))
        if _length(ipiv) != mn:
            ipiv = _new_ipiv(A, mn)
        # lu!(A, ipiv, Val(P), Val(T), check = false): with NoPivot the reference fills the caller's ipiv with 1:n (src/lu.jl:111-113)
        fact = lu_(A, ipiv, alg.pivot, alg.thread, check=False, blocksize=alg.blocksize or None)
        cache.nfactor += 1
        cache.cacheval = (fact, ipiv)
        if not fact.issuccess():
            return LinearSolution(cache.u, ReturnCode.Failure, alg)
        cache.isfresh = False
    fact = cache.cacheval[0]
    if _is_torch(cache.u):
        cache.u.copy_(cache.b)
    else:
        np.copyto(cache.u, cache.b)
    y = ldiv_(fact, cache.u)
    return LinearSolution(y, ReturnCode.Success, alg)


def solve(A, b, alg: RFLUFactorization = None) -> LinearSolution:
    """``solve(LinearProblem(A, b), alg)``: out of place (A and b are copied, as LinearSolve does without aliasing)."""
    if _is_torch(A):
        A2 = A.clone()
        if A2.stride(0) != 1 and A2.stride(1) != 1:
            A2 = A.contiguous()
    else:
        A2 = np.array(A, order="F", copy=True)
    return solve_(init(A2, b, alg))


__all__ = ["RFLUFactorization", "LinearCache", "LinearSolution", "ReturnCode", "init", "init_cacheval", "solve", "solve_"]
