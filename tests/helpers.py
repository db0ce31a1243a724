"""Shared input builders for the parity tests (mirrors of the matrices /root/reference/test/runtests.jl draws)."""
import numpy as np

import oracle as O


def rand_matrix(m, n, seed, dtype=np.float64):
    """`rand(T, m, n)` stand-in (test/runtests.jl:45) from the repo's own counter-based generator."""
    return O.np_uniform(m, n, seed, dtype)


def wilkinson(n, dtype=np.float64):
    """test/runtests.jl:130-140: unit diagonal, -1 strictly below, last column all ones (all-ties pivoting)."""
    A = np.zeros((n, n), dtype=dtype, order="F")
    A[np.arange(n), np.arange(n)] = 1
    A[:, -1] = 1
    A += np.tril(-np.ones((n, n), dtype=dtype), -1)
    return np.asfortranarray(A)


def nopivot_lu_numpy(A):
    """Textbook unpivoted right-looking LU in float64 (comparator for NoPivot info/residual)."""
    F = np.array(A, dtype=np.float64, order="F")
    m, n = F.shape
    info = 0
    for k in range(min(m, n)):
        if F[k, k] != 0:
            F[k + 1:, k] *= 1.0 / F[k, k]
        elif info == 0:
            info = k + 1
        if k + 1 < n:
            F[k + 1:, k + 1:] -= np.outer(F[k + 1:, k], F[k, k + 1:])
    return F, info


REF_SIZES = list(range(1, 11)) + [50, 130, 300]  # test/runtests.jl:39  [1:10; 50:80:200; 300]
