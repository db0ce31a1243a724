"""ctypes front-end of ``librflu_oracle.so`` (the C restatement in ``rflu_oracle.c``) plus numpy helpers.

TEST INFRASTRUCTURE ONLY -- never imported by the product package.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librflu_oracle.so")
_lib = None

_i64 = ctypes.c_int64
_p = ctypes.c_void_p


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (``make -C oracle``).  Returns the path of the shared library."""
    srcs = [os.path.join(_HERE, f) for f in ("rflu_oracle.c", "rflu_oracle_body.inc", "Makefile")]
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B" if force else "all"])
    return _SO


def use_native() -> bool:
    """Switch to a -march=native build made on THIS machine (oracle/_native/, not shipped): for the CPU-baseline timing on
    the GPU box's host cores.  Returns False (and keeps the portable build) if the compiler is unavailable."""
    global _lib, _SO
    try:
        # always rebuilt (-B): a copy compiled for another machine's instruction set must never be picked up
        subprocess.check_call(["make", "-s", "-B", "-C", _HERE, "native"])
    except (OSError, subprocess.CalledProcessError):
        return False
    _SO = os.path.join(_HERE, "_native", "librflu_oracle.so")
    _lib = None
    return True


def set_threads(n: int) -> None:
    """1 = the reference's thread = Val(false); n > 1 = Val(true) on n cores (results do not depend on n)."""
    lib().rfo_set_threads(int(n))


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not _SO.endswith(os.path.join("_native", "librflu_oracle.so")):
            build()
        L = ctypes.CDLL(_SO)
        L.rfo_set_threads.restype = None
        L.rfo_set_threads.argtypes = [ctypes.c_int]
        for sfx in ("f64", "f32"):
            getattr(L, f"rfo_lu_{sfx}").restype = _i64
            getattr(L, f"rfo_lu_{sfx}").argtypes = [_p, _i64, _i64, _i64, _p, ctypes.c_int, _i64, _i64]
            getattr(L, f"rfo_generic_lufact_{sfx}").restype = _i64
            getattr(L, f"rfo_generic_lufact_{sfx}").argtypes = [_p, _i64, _i64, _i64, _i64, ctypes.c_int, _p, _i64]
            getattr(L, f"rfo_nsplit_{sfx}").restype = _i64
            getattr(L, f"rfo_nsplit_{sfx}").argtypes = [_i64]
            getattr(L, f"rfo_fill_uniform_{sfx}").restype = None
            getattr(L, f"rfo_fill_uniform_{sfx}").argtypes = [_p, _i64, _i64, _i64, ctypes.c_uint64]
        _lib = L
    return _lib


def _sfx(dtype) -> str:
    dtype = np.dtype(dtype)
    if dtype == np.float64:
        return "f64"
    if dtype == np.float32:
        return "f32"
    raise TypeError(f"oracle handles float64/float32 only, got {dtype}")


def nsplit(dtype, n: int) -> int:
    return int(getattr(lib(), f"rfo_nsplit_{_sfx(dtype)}")(n))


def lu(A: np.ndarray, pivot: bool = True, blocksize: int = 0, threshold: int = -1, poison_ipiv=None):
    """Factor a copy of ``A`` (any layout; copied to column-major).  Returns (factors F-order, ipiv int64 1-based, info).

    Mirrors ``RecursiveFactorization.lu!(A, ipiv, Val(pivot), Val(false); check=false, blocksize, threshold)``
    (/root/reference/src/lu.jl:97-130); ``info`` uses the positive LAPACK convention.
    """
    F = np.array(A, order="F", copy=True)
    m, n = F.shape
    ipiv = np.empty(min(m, n), dtype=np.int64)
    if poison_ipiv is not None:
        ipiv[:] = poison_ipiv
    info = getattr(lib(), f"rfo_lu_{_sfx(F.dtype)}")(
        F.ctypes.data, m, n, max(m, 1), ipiv.ctypes.data, int(bool(pivot)), int(blocksize), int(threshold)
    )
    return F, ipiv, int(info)


def generic_lufact(A: np.ndarray, pivot: bool = True):
    """The unblocked panel (/root/reference/src/lu.jl:290-338) on a copy of ``A``."""
    F = np.array(A, order="F", copy=True)
    m, n = F.shape
    mn = min(m, n)
    ipiv = np.arange(1, mn + 1, dtype=np.int64)
    info = getattr(lib(), f"rfo_generic_lufact_{_sfx(F.dtype)}")(
        F.ctypes.data, max(m, 1), m, n, mn, int(bool(pivot)), ipiv.ctypes.data, 0
    )
    return F, ipiv, int(info)


def fill_uniform(m: int, n: int, seed: int, dtype=np.float64) -> np.ndarray:
    """Synthetic dense uniform [0,1) input from the C generator (column-major result)."""
    A = np.empty((m, n), dtype=dtype, order="F")
    getattr(lib(), f"rfo_fill_uniform_{_sfx(dtype)}")(A.ctypes.data, m, n, max(m, 1), ctypes.c_uint64(seed))
    return A


# ---- numpy mirror of the generator (bit-for-bit equal to rfo_uniform01; checked in tests/test_oracle.py) ----
_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def np_uniform(m: int, n: int, seed: int, dtype=np.float64) -> np.ndarray:
    with np.errstate(over="ignore"):
        ctr = (np.arange(n, dtype=np.uint64)[None, :] * np.uint64(m) + np.arange(m, dtype=np.uint64)[:, None])
        z = np.uint64(seed) + (ctr + np.uint64(1)) * _GOLD
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return np.asfortranarray(u.astype(dtype))


def unpack_lu(F: np.ndarray):
    """Split packed factors into (L with unit diagonal, U) like LinearAlgebra.LU's .L / .U."""
    m, n = F.shape
    k = min(m, n)
    L = np.tril(F[:, :k], -1) + np.eye(m, k, dtype=F.dtype)
    U = np.triu(F[:k, :])
    return L, U


def perm_from_ipiv(ipiv: np.ndarray, m: int) -> np.ndarray:
    """Row permutation p such that (P*A) = A[p, :], from LAPACK-style 1-based sequential interchanges."""
    p = np.arange(m)
    for i, ip in enumerate(np.asarray(ipiv)):
        j = int(ip) - 1
        if j != i:
            p[i], p[j] = p[j], p[i]
    return p


def residual(A: np.ndarray, F: np.ndarray, ipiv: np.ndarray):
    """Returns (max|L*U - A[p,:]|  -- the reference test's norm(.,Inf) on a matrix, test/runtests.jl:20 --,
    Frobenius ||PA-LU||/||A||), both computed in float64."""
    A64 = np.asarray(A, dtype=np.float64)
    L, U = unpack_lu(np.asarray(F, dtype=np.float64))
    p = perm_from_ipiv(ipiv, A.shape[0])
    R = L @ U - A64[p, :]
    nrm = np.linalg.norm(A64)
    return float(np.max(np.abs(R))) if R.size else 0.0, float(np.linalg.norm(R) / nrm) if nrm > 0 else 0.0


# ---- butterfly pre-transform: NumPy restatement of /root/reference/src/butterflylu.jl (test infrastructure) -----------------
def butterfly_mul_level(A: np.ndarray, u: np.ndarray, v: np.ndarray) -> None:
    """🦋mul_level! (src/butterflylu.jl:59-88), in place on the view A; same expressions in the same order."""
    M, N = A.shape
    mh, nh = M >> 1, N >> 1
    A11, A21, A12, A22 = A[:mh, :nh].copy(), A[mh:, :nh].copy(), A[:mh, nh:].copy(), A[mh:, nh:].copy()
    T1, T2, T3, T4 = A11 + A12, A21 + A22, A11 - A12, A21 - A22
    C11, C21, C12, C22 = T1 + T2, T1 - T2, T3 + T4, T3 - T4
    u1, u2 = u[:mh, None], u[mh:, None]
    v1, v2 = v[None, :nh], v[None, nh:]
    A[:mh, :nh] = u1 * C11 * v1
    A[mh:, :nh] = u2 * C21 * v1
    A[:mh, nh:] = u1 * C12 * v2
    A[mh:, nh:] = u2 * C22 * v2


def butterfly_mul(A: np.ndarray, uv: np.ndarray) -> np.ndarray:
    """🦋mul! (src/butterflylu.jl:90-113): level 2 on the four quadrants, then level 1; in place, returns A."""
    M = A.shape[0]
    h = M >> 1
    U1, V1, U2, V2 = uv[:h], uv[h:M], uv[M:M + h], uv[M + h:2 * M]
    butterfly_mul_level(A[:h, :h], U1, V1)
    butterfly_mul_level(A[h:, :h], U2, V1)
    butterfly_mul_level(A[:h, h:], U1, V2)
    butterfly_mul_level(A[h:, h:], U2, V2)
    butterfly_mul_level(A, uv[2 * M:3 * M], uv[3 * M:4 * M])
    return A


def _bfly_matrix(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """🦋!(C, A::Diagonal, B::Diagonal) (src/butterflylu.jl:133-146): [Da Db; Da -Db]."""
    n = a.size
    C = np.zeros((2 * n, 2 * n), dtype=a.dtype)
    i = np.arange(n)
    C[i, i] = a
    C[i + n, i] = a
    C[i, i + n] = b
    C[i + n, i + n] = -b
    return C


def butterfly_materialize_uv(uv: np.ndarray, M: int):
    """materializeUV (src/butterflylu.jl:149-178): dense U = Bu2*Bu1, V = Bv2*Bv1."""
    h = M >> 1
    q = h >> 1
    def halves(x):
        k = x.size >> 1
        return x[:k], x[k:]
    U1, U2 = uv[:h], uv[2 * h:M + h]
    V1, V2 = uv[h:2 * h], uv[3 * h:2 * h + M]
    Uf, Vf = uv[2 * M:3 * M], uv[3 * M:4 * M]
    Bu2 = np.zeros((M, M), dtype=uv.dtype)
    Bu2[:h, :h] = _bfly_matrix(*halves(U1))
    Bu2[h:, h:] = _bfly_matrix(*halves(U2))
    Bv2 = np.zeros((M, M), dtype=uv.dtype)
    Bv2[:h, :h] = _bfly_matrix(*halves(V1))
    Bv2[h:, h:] = _bfly_matrix(*halves(V2))
    return Bu2 @ _bfly_matrix(*halves(Uf)), Bv2 @ _bfly_matrix(*halves(Vf))
