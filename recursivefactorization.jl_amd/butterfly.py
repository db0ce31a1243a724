"""Host-side mirror of the reference's randomized-butterfly solver (/root/reference/src/butterflylu.jl), served by librflu.so.

    🦋workspace(A, b, Val(SEED))   :20-43   -> ``ButterflyWorkspace(A, b, seed)``   (alias ``butterfly_workspace``, :57)
    🦋solve!(ws, thread)           :45-55   -> ``butterfly_solve_(ws, thread)``
    🦋mul!(A, uv)                  :90-113  -> ``butterfly_mul_(A, uv)``            (rflu_butterfly_mul_*_dev, one pass)
    🦋generate_random!(A)          :16-19   -> ``generate_random(n, dtype, seed)``   0.5*exp(x), x uniform in [-0.05, 0.05)
    pad!(A)                        :180-197 -> ``pad(A)``

The transform A <- U' A V makes an unpivoted factorization safe, so the GPU path is: one streaming pass over A
(butterfly.hip), ``lu!(A, Val(false))`` on the MI355X (no pivot search, hence no per-column latency chain), then
x = V * ((U'AV) \\ (U' b)) with the two outer products applied as O(n) butterflies (``rflu_butterfly_vec_*``) instead of
the reference's dense U, V (materializeUV, :149-178 -- kept only in the test oracle).  The random stream is NOT the
reference's: VectorizedRNG's output depends on the CPU's SIMD width (test/runtests.jl:143-150), so only the distribution
and the layout of ``uv`` are mirrored.  No CPU fallback: the device is required."""
from __future__ import annotations

import ctypes

import numpy as np

from . import _ffi
from .lu import NoPivot, _sfx, ldiv_, lu_


def generate_random(n: int, dtype=np.float64, seed: int = 888) -> np.ndarray:
    """4n butterfly entries exphalf(x) = 0.5*exp(x), x ~ U[-0.05, 0.05)  (src/butterflylu.jl:7-19)."""
    rng = np.random.Generator(np.random.Philox(seed))
    x = rng.random(4 * n) * 0.1 - 0.05
    return (0.5 * np.exp(x)).astype(dtype)


def pad(A: np.ndarray) -> np.ndarray:
    """pad!(A): grow to the next multiple of 4 with an identity block (src/butterflylu.jl:180-197)."""
    m, n = A.shape
    xn = 4 - m % 4
    out = np.zeros((m + xn, n + xn), dtype=A.dtype, order="F")
    out[:m, :n] = A
    out[np.arange(m, m + xn), np.arange(n, n + xn)] = 1
    return out


def butterfly_mul_(A, uv, *, handle=None):
    """🦋mul!(A, uv): A <- U' A V in place.  ``A``: column-major CUDA tensor (n x n, n % 4 == 0), ``uv``: CUDA vector (4n)."""
    import torch

    n = int(A.shape[0])
    if A.shape[0] != A.shape[1] or n % 4:
        raise ValueError("butterfly needs a square matrix whose size is a multiple of 4 (pad first)")
    if not (A.is_cuda and uv.is_cuda and A.dtype == uv.dtype and A.stride(0) == 1 and uv.is_contiguous() and uv.numel() == 4 * n):
        raise ValueError("A must be a column-major CUDA matrix and uv a contiguous CUDA vector of 4n entries of the same dtype")
    h = handle or _ffi.default_handle(A.device.index or 0)
    h.set_stream(torch.cuda.current_stream(A.device).cuda_stream)
    h.call(f"rflu_butterfly_mul_{_sfx(A.dtype)}_dev", n, ctypes.c_void_p(A.data_ptr()), A.stride(1), ctypes.c_void_p(uv.data_ptr()))
    return A


def _vec(x, uv, transpose_u: bool, handle=None):
    import torch

    n = int(x.shape[0])
    h = handle or _ffi.default_handle(x.device.index or 0)
    h.set_stream(torch.cuda.current_stream(x.device).cuda_stream)
    h.call(f"rflu_butterfly_vec_{_sfx(x.dtype)}_dev", n, 1, ctypes.c_void_p(x.data_ptr()), n, ctypes.c_void_p(uv.data_ptr()),
           int(transpose_u))
    return x


class ButterflyWorkspace:
    """🦋workspace (src/butterflylu.jl:20-43): padded copy of the system on the device + the random butterfly entries.
    ``A`` (n x n) and ``b`` (n) are host NumPy arrays, as the reference takes host arrays."""

    def __init__(self, A, b, seed: int = 888, device: int = 0):
        import torch

        A = np.asarray(A)
        b = np.asarray(b)
        _sfx(A.dtype)
        self.n = int(A.shape[0])
        if A.shape[0] != A.shape[1] or b.shape != (self.n,):
            raise ValueError("butterfly workspace needs a square A and a matching vector b")
        if self.n % 4:
            A = pad(A)
            xn = 4 - self.n % 4
            b = np.concatenate([b, np.random.Generator(np.random.Philox(seed + 1)).random(xn).astype(b.dtype)])  # :33
        dev = torch.device("cuda", device)
        self.A = torch.from_numpy(np.ascontiguousarray(np.asarray(A).T)).to(dev).T      # column-major on the device
        self.b = torch.from_numpy(np.ascontiguousarray(b)).to(dev)
        self.ws = torch.from_numpy(generate_random(self.A.shape[0], A.dtype, seed)).to(dev)
        self.out = None
        self.F = None


butterfly_workspace = ButterflyWorkspace


def butterfly_solve_(ws: ButterflyWorkspace, thread=False, *, blocksize=None):
    """🦋solve!(ws, thread) (src/butterflylu.jl:45-55): returns x (host array, length n) with A x = b."""
    del thread  # accepted for signature parity
    butterfly_mul_(ws.A, ws.ws)                                       # 🦋mul!(A, ws)
    ws.F = lu_(ws.A, None, NoPivot(), check=False, blocksize=blocksize)   # lu!(A, Val(false), thread)
    tmp = _vec(ws.b.clone(), ws.ws, True)                             # mul!(tmp, U', b)
    ldiv_(ws.F, tmp)                                                  # ldiv!(F, tmp, thread)
    _vec(tmp, ws.ws, False)                                           # mul!(b, V, tmp)
    ws.b = tmp
    ws.out = tmp[: ws.n].cpu().numpy()                                # out .= @view b[1:n]
    return ws.out
