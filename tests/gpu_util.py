"""Helpers for the -m gpu tests: device buffers via torch (plumbing only), every compute call goes through the C ABI."""
import ctypes

import numpy as np
import torch

from recursivefactorization.jl_amd import _ffi


def handle():
    h = _ffi.default_handle(0)
    h.set_stream(None)
    return h


def ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def to_dev_rm(A):
    """numpy (any layout) -> row-major device tensor with the same logical shape."""
    return torch.from_numpy(np.ascontiguousarray(A)).to("cuda:0")


def to_dev_cm(A):
    """numpy -> column-major device view (stride(0) == 1), like a Julia Matrix."""
    return torch.from_numpy(np.ascontiguousarray(np.asarray(A).T)).to("cuda:0").T


def sfx(dtype):
    return "f64" if np.dtype(dtype) == np.float64 else "f32"


def tdtype(dtype):
    return torch.float64 if np.dtype(dtype) == np.float64 else torch.float32


def fill_uniform_cm(n, dtype=np.float64, seed=12, diag_add=0.0, m=None):
    """Column-major m x n device matrix of the repo's counter-based uniform[0,1) generator (what `rand(n, n)` stands for
    in BASELINE.json), produced on the device by the library's fill kernel -- no 32 GiB host transfers at n = 65536."""
    m = n if m is None else m
    t = torch.empty((n, m), dtype=tdtype(dtype), device="cuda:0").T   # shape (m, n), stride(0) == 1
    handle().call(f"rflu_fill_uniform_{sfx(dtype)}_dev", ptr(t), m, n, m, 0, seed, m, 0, 0, float(diag_add))
    return t


def matvec_residual(A, LU, ipiv, chunk=4096, trials=2):
    """max over random x of ||P*A*x - L*(U*x)|| / ||A*x|| in float64 with plain torch ops (an independent checker, O(n^2)):
    the size-independent stand-in for ||PA - LU|| / ||A|| where the n^3 product is not affordable (BASELINE configs 2-4)."""
    n = A.shape[0]
    dev = A.device
    ip = ipiv.cpu().numpy() if hasattr(ipiv, "cpu") else np.asarray(ipiv)
    perm = np.arange(n)
    for i, t in enumerate(ip):
        j = int(t) - 1
        if j != i:
            perm[i], perm[j] = perm[j], perm[i]
    perm = torch.from_numpy(perm).to(dev)
    rows = torch.arange(n, device=dev)[:, None]
    gen = torch.Generator(device="cpu").manual_seed(1234)
    worst = 0.0
    for _ in range(trials):
        x = torch.rand(n, dtype=torch.float64, generator=gen).to(dev)
        ax = torch.zeros(n, dtype=torch.float64, device=dev)
        ux = torch.zeros(n, dtype=torch.float64, device=dev)
        for c0 in range(0, n, chunk):
            c1 = min(c0 + chunk, n)
            cols = torch.arange(c0, c1, device=dev)[None, :]
            xb = x[c0:c1]
            ax += A[:, c0:c1].to(torch.float64) @ xb
            blk = LU[:, c0:c1].to(torch.float64)
            ux += torch.where(rows <= cols, blk, torch.zeros((), dtype=torch.float64, device=dev)) @ xb
            del blk
        lz = ux.clone()   # unit diagonal of L
        for c0 in range(0, n, chunk):
            c1 = min(c0 + chunk, n)
            cols = torch.arange(c0, c1, device=dev)[None, :]
            blk = LU[:, c0:c1].to(torch.float64)
            lz += torch.where(rows > cols, blk, torch.zeros((), dtype=torch.float64, device=dev)) @ ux[c0:c1]
            del blk
        r = torch.linalg.norm(ax[perm] - lz) / torch.linalg.norm(ax)
        worst = max(worst, float(r.item()))
    return worst


def matvec_residual_slabs(n, slabs, layout, ipiv, seed=12, diag_add=0.0, trials=1):
    """The same O(n^2) check for a factorization held as 1-D block-column slabs (row-major n x local columns per logical
    device, layout = [(j0, w, owner, lc)]): ||P*A*x - L*(U*x)|| / ||A*x|| with A regenerated block column by block column from
    the library's counter-based generator (nothing of size n x n is allocated besides the slabs themselves)."""
    dev = slabs[0].device
    dtype = np.float64 if slabs[0].dtype == torch.float64 else np.float32
    ip = np.asarray(ipiv)
    perm = np.arange(n)
    for i, t in enumerate(ip):
        j = int(t) - 1
        if j != i:
            perm[i], perm[j] = perm[j], perm[i]
    perm = torch.from_numpy(perm).to(dev)
    rows = torch.arange(n, device=dev)[:, None]
    gen = torch.Generator(device="cpu").manual_seed(4321)
    zero = torch.zeros((), dtype=torch.float64, device=dev)
    worst = 0.0
    for _ in range(trials):
        x = torch.rand(n, dtype=torch.float64, generator=gen).to(dev)
        ax = torch.zeros(n, dtype=torch.float64, device=dev)
        ux = torch.zeros(n, dtype=torch.float64, device=dev)
        for (j0, w, owner, lc) in layout:
            blkA = torch.empty((w, n), dtype=tdtype(dtype), device=dev).T   # n x w, column-major
            handle().call(f"rflu_fill_uniform_{sfx(dtype)}_dev", ptr(blkA), n, w, n, 0, seed, n, 0, j0, float(diag_add))
            xb = x[j0:j0 + w]
            ax += blkA.to(torch.float64) @ xb
            cols = torch.arange(j0, j0 + w, device=dev)[None, :]
            blk = slabs[owner][:, lc:lc + w].to(dev).to(torch.float64)
            ux += torch.where(rows <= cols, blk, zero) @ xb
            del blk, blkA
        lz = ux.clone()   # unit diagonal of L
        for (j0, w, owner, lc) in layout:
            cols = torch.arange(j0, j0 + w, device=dev)[None, :]
            blk = slabs[owner][:, lc:lc + w].to(dev).to(torch.float64)
            lz += torch.where(rows > cols, blk, zero) @ ux[j0:j0 + w]
            del blk
        r = torch.linalg.norm(ax[perm] - lz) / torch.linalg.norm(ax)
        worst = max(worst, float(r.item()))
    return worst
