"""Host-side mirror of RecursiveFactorization.jl's public surface for the LU hot path, served by librflu.so.

Reference interface mirrored (citations into /root/reference/src/lu.jl):
    lu(A, pivot = Val(true), thread = Val(false); kwargs...)                       :19-21   -> ``lu``
    lu!(A, pivot = Val(true), thread = Val(false); check, kwargs...)               :67-83   -> ``lu_``  (ipiv=None)
    lu!(A, ipiv, pivot, thread; check = Val(true), blocksize, threshold)           :97-130  -> ``lu_``
    normalize_pivot: Val(true)/RowMaximum(), Val(false)/NoPivot()                  :10-17   -> ``pivot`` accepts both
    NotIPIV (lazy identity pivots for NoPivot)                                     :27-40   -> ``NotIPIV``
    LU(A, ipiv, info), checknonsingular(info) -> SingularException                 :128-129 -> ``LU``, ``SingularException``
    ldiv!(F, B) (stdlib for pivoted LU; the package's own for NotIPIV)             :60-64   -> ``ldiv_``
    NoPivot failures carry a NEGATIVE info on Julia >= 1.11                        :25,250,324 -> ``NOPIVOT_NEGATIVE_INFO``
    Adjoint/Transpose wrappers                                                     :85-87   -> ``Adjoint`` / ``lu(A.T ...)``

Same names, argument meaning and error behaviour; Julia's ``!`` is spelled ``_``.  What differs, deliberately:
  * every Float64/Float32 matrix goes to the HIP path, whatever its size -- ``threshold`` (the reference's
    recursive/unblocked crossover, :90,114) is accepted and ignored, and there is NO CPU fallback: other element types
    raise ``TypeError`` (the Julia glue in INTEGRATION.md keeps the reference's own CPU code for those);
  * ``thread`` is accepted and ignored (the GPU path has no thread flag);
  * ``blocksize``: ``None``/0 = library default (pure Toledo recursion below 1024 columns; above, right-looking block
    columns of 256 ... 2048 by matrix size, block-column lookahead for the tall panels and the leaf-wise schedule for
    panels of at most 8192 rows, see include/rflu.h); negative = pure
    recursion; 64/128/256... = width of the outer right-looking block column.

Inputs: a NumPy array (host; staged through HBM by ``rflu_getrf_*``) or a ``torch`` tensor on the GPU
(column-major view, i.e. ``stride(0) == 1``, -> ``rflu_getrf_*_dev``; C-contiguous -> ``rflu_getrf_rm_*_dev``).
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass

import numpy as np

from . import _ffi

NOPIVOT_NEGATIVE_INFO = True  # the convention of Julia >= 1.11 (src/lu.jl:25)


class SingularException(ArithmeticError):
    """LinearAlgebra.SingularException(info): raised by ``check`` when a pivot is exactly zero (src/lu.jl:128)."""

    def __init__(self, info: int):
        super().__init__(f"matrix is singular to working precision (info = {info})")
        self.info = info


class RowMaximum:
    """LinearAlgebra.RowMaximum(): partial pivoting (== Val(true), src/lu.jl:13)."""


class NoPivot:
    """LinearAlgebra.NoPivot(): no pivoting (== Val(false), src/lu.jl:14)."""


class Val:
    """Julia's Val{x}: ``Val(True)`` / ``Val(False)`` are accepted wherever the reference takes ``Val``."""

    def __init__(self, x):
        self.x = x


def normalize_pivot(pivot) -> bool:
    """src/lu.jl:10-17."""
    if isinstance(pivot, Val):
        pivot = pivot.x
    if isinstance(pivot, RowMaximum) or pivot is RowMaximum:
        return True
    if isinstance(pivot, NoPivot) or pivot is NoPivot:
        return False
    if isinstance(pivot, (bool, np.bool_)):
        return bool(pivot)
    raise TypeError(f"pivot must be Val(true/false), RowMaximum() or NoPivot(), got {pivot!r}")


def _as_bool(flag) -> bool:
    return bool(flag.x) if isinstance(flag, Val) else bool(flag)


class NotIPIV:
    """Zero-storage identity pivot vector (src/lu.jl:27-40): ``getindex(::NotIPIV, i) = i``."""

    def __init__(self, length: int):
        self.len = int(length)

    def __len__(self):
        return self.len

    def __getitem__(self, i):
        if isinstance(i, slice):
            return NotIPIV(len(range(*i.indices(self.len))))
        if not 0 <= i < self.len:
            raise IndexError(i)
        return i + 1  # 1-based pivot values, like every ipiv here

    def __array__(self, dtype=None, copy=None):
        return np.arange(1, self.len + 1, dtype=dtype or np.int64)


@dataclass
class LU:
    """LinearAlgebra.LU{T}: packed factors (aliasing the caller's matrix for ``lu_``), 1-based ipiv, info."""

    factors: object
    ipiv: object
    info: int

    def issuccess(self) -> bool:
        return self.info == 0

    def _host(self):
        f = self.factors
        if hasattr(f, "detach"):
            f = f.detach().cpu().numpy()
        p = self.ipiv
        if hasattr(p, "detach"):
            p = p.detach().cpu().numpy()
        return np.asarray(f), np.asarray(p)

    @property
    def L(self):
        f, _ = self._host()
        m, n = f.shape
        k = min(m, n)
        return np.tril(f[:, :k], -1) + np.eye(m, k, dtype=f.dtype)

    @property
    def U(self):
        f, _ = self._host()
        return np.triu(f[: min(f.shape), :])

    @property
    def p(self):
        """Row permutation (0-based) such that L*U == A[p, :]."""
        f, ip = self._host()
        perm = np.arange(f.shape[0])
        for i, t in enumerate(ip):
            j = int(t) - 1
            if j != i:
                perm[i], perm[j] = perm[j], perm[i]
        return perm


class Adjoint:
    """``A'`` / ``transpose(A)`` wrapper for real matrices: lu(A') = adjoint(lu(parent(A))) (src/lu.jl:85-87)."""

    def __init__(self, parent):
        self.parent = parent


Transpose = Adjoint


def _checknonsingular(info: int):
    if info != 0:
        raise SingularException(abs(info))


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def _sfx(dtype) -> str:
    name = str(dtype).replace("torch.", "")
    if name == "float64":
        return "f64"
    if name == "float32":
        return "f32"
    raise TypeError(
        f"the MI355X LU path serves Float64/Float32 only (got {dtype}); the reference routes other element types "
        "through its generic CPU code (src/lu.jl:122-123), which this package does not carry"
    )


def lu_(A, ipiv=None, pivot=True, thread=False, *, check=True, blocksize=None, threshold=None, handle=None) -> LU:
    """``lu!``: factor ``A`` in place.  ``ipiv`` None -> allocate (``NotIPIV`` for NoPivot), like src/lu.jl:67-83."""
    del thread, threshold  # accepted for signature parity; the HIP path has neither knob
    if isinstance(A, Adjoint):
        return Adjoint(lu_(A.parent, ipiv, pivot, check=check, blocksize=blocksize, handle=handle))
    piv = normalize_pivot(pivot)
    bs = int(blocksize or 0)
    info = ctypes.c_int64(0)
    if getattr(A, "ndim", None) != 2:
        raise ValueError("lu! needs a matrix")
    m, n = int(A.shape[0]), int(A.shape[1])
    mn = min(m, n)

    if _is_torch(A):
        import torch

        if not A.is_cuda:
            raise _ffi.RfluError("torch input must live on the MI355X (device='cuda'); host data goes in as NumPy")
        sfx = _sfx(A.dtype)
        h = handle or _ffi.default_handle(A.device.index or 0)
        h.set_stream(torch.cuda.current_stream(A.device).cuda_stream)
        if ipiv is None:
            ipiv_t = torch.empty(mn, dtype=torch.int64, device=A.device) if piv else None
        else:
            ipiv_t = ipiv
            if not (_is_torch(ipiv_t) and ipiv_t.is_cuda and ipiv_t.dtype == torch.int64 and ipiv_t.is_contiguous()):
                raise TypeError("ipiv for a GPU matrix must be a contiguous int64 CUDA tensor")
            if ipiv_t.numel() < mn:
                raise ValueError("ipiv is shorter than min(m, n)")
        ip_ptr = ctypes.c_void_p(ipiv_t.data_ptr() if ipiv_t is not None else 0)
        if m > 0 and n > 0:
            if A.stride(0) == 1 and A.stride(1) >= max(m, 1):  # column-major view (Julia layout)
                h.call(f"rflu_getrf_{sfx}_dev", m, n, ctypes.c_void_p(A.data_ptr()), A.stride(1), ip_ptr, int(piv), bs,
                       ctypes.byref(info))
            elif A.stride(1) == 1 and A.stride(0) >= max(n, 1):  # row-major: the library's internal layout
                h.call(f"rflu_getrf_rm_{sfx}_dev", m, n, ctypes.c_void_p(A.data_ptr()), A.stride(0), ip_ptr, int(piv), bs,
                       ctypes.byref(info))
            else:
                raise ValueError("matrix must be dense column-major or row-major (unit stride in one dimension)")
        out_ipiv = ipiv_t if ipiv_t is not None else NotIPIV(mn)
    else:
        if not isinstance(A, np.ndarray):
            raise TypeError("A must be a numpy.ndarray or a CUDA torch.Tensor")
        sfx = _sfx(A.dtype)
        if not A.flags.f_contiguous:
            raise ValueError("lu! works in place on column-major (Fortran-ordered) arrays; use lu() to copy")
        h = handle or _ffi.default_handle(0)
        h.set_stream(None)
        if ipiv is None:
            ipiv_a = np.empty(mn, dtype=np.int64) if piv else None
        else:
            ipiv_a = ipiv
            if not (isinstance(ipiv_a, np.ndarray) and ipiv_a.dtype == np.int64 and ipiv_a.flags.c_contiguous):
                raise TypeError("ipiv must be a contiguous int64 numpy array (Julia BlasInt)")
            if ipiv_a.size < mn:
                raise ValueError("ipiv is shorter than min(m, n)")
        ip_ptr = ctypes.c_void_p(ipiv_a.ctypes.data if ipiv_a is not None else 0)
        if m > 0 and n > 0:
            h.call(f"rflu_getrf_{sfx}", m, n, ctypes.c_void_p(A.ctypes.data), max(m, 1), ip_ptr, int(piv), bs,
                   ctypes.byref(info))
        out_ipiv = ipiv_a if ipiv_a is not None else NotIPIV(mn)

    inf = int(info.value)
    if not piv and NOPIVOT_NEGATIVE_INFO:
        inf = -inf
    if _as_bool(check):
        _checknonsingular(inf)
    return LU(A, out_ipiv, inf)


def lu(A, pivot=True, thread=False, **kwargs) -> LU:
    """``lu``: out of place, ``lu!(copy(A), ...)`` (src/lu.jl:19-21)."""
    if isinstance(A, Adjoint):
        return Adjoint(lu(A.parent, pivot, thread, **kwargs))
    if _is_torch(A):
        C = A.clone()
        if C.stride(0) != 1 and C.stride(1) != 1:
            C = A.contiguous()
    else:
        C = np.array(A, order="F", copy=True)
    return lu_(C, None, pivot, thread, **kwargs)


def ldiv_(F: LU, B, *, handle=None):
    """``ldiv!(F, B)``: overwrite ``B`` (a vector or an n x k matrix) with ``A \\ B`` using the factorization ``F``.

    Mirrors stdlib ``ldiv!(::LU, B)`` on the object ``lu!`` returns -- what LinearSolve's ``solve!`` calls right after the
    factorization -- and the package's own ``ldiv!`` for ``NotIPIV`` factors (/root/reference/src/lu.jl:60-64; tested at
    test/runtests.jl:21-28, 116-128).  Served by ``rflu_getrs_*`` (interchanges, fused unit-lower TRSM, upper solve).
    Raises ``SingularException`` when ``F.info != 0`` (the solve would divide by an exactly zero pivot).
    """
    if isinstance(F, Adjoint):
        raise NotImplementedError("solve with the adjoint factorization is not part of the MI355X path")
    if F.info != 0:
        raise SingularException(abs(F.info))
    A = F.factors
    n = int(A.shape[0])
    if A.shape[0] != A.shape[1]:
        raise ValueError("ldiv! needs a square factorization")
    if B.shape[0] != n:
        raise ValueError("right-hand side has the wrong number of rows")
    nrhs = 1 if B.ndim == 1 else int(B.shape[1])
    nopiv = isinstance(F.ipiv, NotIPIV)
    if _is_torch(A):
        import torch

        if not (_is_torch(B) and B.is_cuda and B.dtype == A.dtype):
            raise TypeError("B must be a CUDA tensor of the factorization's dtype")
        sfx = _sfx(A.dtype)
        h = handle or _ffi.default_handle(A.device.index or 0)
        h.set_stream(torch.cuda.current_stream(A.device).cuda_stream)
        ip = ctypes.c_void_p(0 if nopiv else F.ipiv.data_ptr())
        if B.ndim == 1 and n > 1 and B.stride(0) != 1:
            # a strided vector view (e.g. a column of a row-major tensor) would be read and written at the wrong addresses
            raise ValueError("a vector right-hand side must be contiguous (stride 1); copy the view first")
        if A.stride(0) == 1:  # column-major factors -> column-major right-hand sides
            if B.ndim == 2 and not (B.stride(0) == 1 and B.stride(1) >= n):
                raise ValueError("B must be column-major like the factors")
            ldb = n if B.ndim == 1 else B.stride(1)
            h.call(f"rflu_getrs_{sfx}_dev", n, nrhs, ctypes.c_void_p(A.data_ptr()), A.stride(1), ip,
                   ctypes.c_void_p(B.data_ptr()), ldb)
        else:                 # row-major factors (rflu_getrf_rm) -> row-major right-hand sides
            if B.ndim == 2 and not (B.stride(1) == 1 and B.stride(0) >= nrhs):
                raise ValueError("B must be row-major like the factors (unit column stride, row stride >= nrhs)")
            ldb = 1 if B.ndim == 1 else B.stride(0)
            h.call(f"rflu_getrs_rm_{sfx}_dev", n, nrhs, ctypes.c_void_p(A.data_ptr()), A.stride(0), ip,
                   ctypes.c_void_p(B.data_ptr()), ldb)
        return B
    if not (isinstance(B, np.ndarray) and B.dtype == A.dtype and (B.ndim == 1 and B.flags.c_contiguous or B.flags.f_contiguous)):
        raise TypeError("B must be a column-major numpy array of the factorization's dtype")
    sfx = _sfx(A.dtype)
    h = handle or _ffi.default_handle(0)
    h.set_stream(None)
    ipiv = None if nopiv else np.ascontiguousarray(F.ipiv, dtype=np.int64)
    h.call(f"rflu_getrs_{sfx}", n, nrhs, ctypes.c_void_p(A.ctypes.data), max(n, 1),
           ctypes.c_void_p(0 if ipiv is None else ipiv.ctypes.data), ctypes.c_void_p(B.ctypes.data), max(n, 1))
    return B


def last_path(device: int = 0) -> str:
    """Which implementation served the last factorization on ``device`` (``enum rflu_path`` of include/rflu.h: "hip-recursive" /
    "hip-blocked" / "hip-lookahead" / "hip-engine" / "none")."""
    return {0: "none", 1: "hip-recursive", 2: "hip-blocked", 3: "hip-lookahead", 4: "hip-engine"}[_ffi.default_handle(device).last_path()]
