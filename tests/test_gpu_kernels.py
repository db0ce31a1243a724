"""-m gpu: each hand-written HIP kernel of the path against a numpy/oracle restatement of the same step, through the
C ABI building blocks of include/rflu.h (row-major R layout)."""
import ctypes

import numpy as np
import pytest
import torch

import oracle as O
from gpu_util import handle, ptr, sfx, tdtype, to_dev_rm

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", [(128, 128, 16), (256, 384, 64), (130, 70, 33), (1, 1, 1), (64, 300, 200),
                                   (513, 257, 129), (1024, 1024, 512),
                                   # K = 64 / 128 with N <= 2K: the latency-optimised kernel of the panel recursion
                                   (64, 64, 64), (200, 64, 64), (1000, 100, 64), (513, 128, 128), (70, 256, 128),
                                   (3000, 70, 128), (2048, 128, 64)])
def test_gemm_sub(dtype, shape):
    # schur_complement! (src/lu.jl:265-284): C <- C - A*B ; asymmetric operands catch transposed fragments
    M, N, K = shape
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    A = rng.uniform(-1, 1, (M, K)).astype(dtype)
    B = rng.uniform(-1, 1, (K, N)).astype(dtype)
    C = rng.uniform(-1, 1, (M, N)).astype(dtype)
    dA, dB, dC = to_dev_rm(A), to_dev_rm(B), to_dev_rm(C)
    handle().call(f"rflu_gemm_rm_{sfx(dtype)}_dev", M, N, K, ptr(dA), K, ptr(dB), N, ptr(dC), N)
    torch.cuda.synchronize()
    ref = C.astype(np.float64) - A.astype(np.float64) @ B.astype(np.float64)
    tol = (K + 4) * np.finfo(dtype).eps * 4
    assert np.max(np.abs(dC.cpu().numpy() - ref)) < tol


def test_gemm_sub_strided_views_unaligned():
    # operands as sub-blocks of one buffer with odd offsets (the scalar guarded load path)
    rng = np.random.default_rng(5)
    ld = 531
    buf = rng.uniform(-1, 1, (400, ld))
    d = to_dev_rm(buf)
    M, N, K = 190, 133, 77
    a0, b0, c0 = (200 * ld + 3), (5 * ld + 301), (201 * ld + 301)
    es = 8
    handle().call("rflu_gemm_rm_f64_dev", M, N, K, ctypes.c_void_p(d.data_ptr() + a0 * es), ld,
                  ctypes.c_void_p(d.data_ptr() + b0 * es), ld, ctypes.c_void_p(d.data_ptr() + c0 * es), ld)
    torch.cuda.synchronize()
    ref = buf.copy()
    ref[201:201 + M, 301:301 + N] -= buf[200:200 + M, 3:3 + K] @ buf[5:5 + K, 301:301 + N]
    assert np.max(np.abs(d.cpu().numpy() - ref)) < 1e-12


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,nrhs", [(1, 5), (17, 40), (64, 64), (64, 1000), (100, 33), (256, 300), (300, 129), (1024, 512)])
def test_trsm_unit_lower(dtype, n, nrhs):
    # ldiv!(UnitLowerTriangular(A11), A12) (src/lu.jl:235): strict lower read, unit diagonal implied
    rng = np.random.default_rng(n + nrhs)
    L = rng.uniform(-1, 1, (n, n)).astype(dtype) * dtype(0.5)  # diagonal/upper garbage must be ignored
    B = rng.uniform(-1, 1, (n, nrhs)).astype(dtype)
    dL, dB = to_dev_rm(L), to_dev_rm(B)
    handle().call(f"rflu_trsm_rm_{sfx(dtype)}_dev", n, nrhs, ptr(dL), n, ptr(dB), nrhs)
    torch.cuda.synchronize()
    Lu = np.tril(L.astype(np.float64), -1) + np.eye(n)
    X = dB.cpu().numpy().astype(np.float64)
    resid = np.max(np.abs(Lu @ X - B.astype(np.float64)))
    assert resid < 50 * n * np.finfo(dtype).eps * max(1.0, np.max(np.abs(X)))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_laswp_matches_sequential_interchanges(dtype):
    # apply_permutation! (src/lu.jl:177-188) incl. repeated targets, chains and identity entries; bit-exact
    rng = np.random.default_rng(3)
    m, ncols, c0 = 700, 333, 17
    k0, k1 = 64, 64 + 150  # spans three chunks, last one partial
    A = rng.uniform(-1, 1, (m, 400)).astype(dtype)
    ipiv = np.arange(1, m + 1, dtype=np.int64)
    for k in range(k0, k1):
        r = rng.integers(0, 10)
        ipiv[k] = (k if r == 0 else (k0 + 5 if (r == 1 and k < k0 + 5) else rng.integers(k, m))) + 1
    ipiv[k0 + 7] = ipiv[k0 + 3]  # repeated target
    ref = A.copy()
    for k in range(k0, k1):
        p = ipiv[k] - 1
        if p != k:
            ref[[k, p], c0:c0 + ncols] = ref[[p, k], c0:c0 + ncols]
    dA = to_dev_rm(A)
    dP = torch.from_numpy(ipiv).to("cuda:0")
    handle().call(f"rflu_laswp_rm_{sfx(dtype)}_dev", ptr(dA), 400, m, c0, ncols, ptr(dP), k0, k1)
    torch.cuda.synchronize()
    assert np.array_equal(dA.cpu().numpy(), ref)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", [(1, 1), (5, 130), (64, 64), (257, 100), (1000, 1030)])
def test_layout_change_round_trip(dtype, shape):
    m, n = shape
    rng = np.random.default_rng(m + n)
    A = np.asfortranarray(rng.uniform(-1, 1, (m, n)).astype(dtype))
    lda = m + 3
    buf = np.zeros((lda, n), dtype=dtype, order="F")
    buf[:m] = A
    dcm = torch.from_numpy(np.ascontiguousarray(buf.T)).to("cuda:0")  # memory == column-major buf
    ldr = n + 5
    drm = torch.zeros((m, ldr), dtype=tdtype(dtype), device="cuda:0")
    h = handle()
    h.call(f"rflu_cm_to_rm_{sfx(dtype)}_dev", m, n, ptr(dcm), lda, ptr(drm), ldr)
    torch.cuda.synchronize()
    assert np.array_equal(drm.cpu().numpy()[:, :n], A)
    dback = torch.zeros_like(dcm)
    h.call(f"rflu_rm_to_cm_{sfx(dtype)}_dev", m, n, ptr(drm), ldr, ptr(dback), lda)
    torch.cuda.synchronize()
    assert np.array_equal(dback.cpu().numpy().T[:m], A)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_device_generator_is_bit_identical_to_oracle(dtype):
    m, n = 300, 170
    h = handle()
    d = torch.zeros((n, m), dtype=tdtype(dtype), device="cuda:0")  # column-major m x n
    h.call(f"rflu_fill_uniform_{sfx(dtype)}_dev", ptr(d), m, n, m, 0, 12, m, 0, 0, 0.0)
    torch.cuda.synchronize()
    assert np.array_equal(d.cpu().numpy().T, O.np_uniform(m, n, 12, dtype))
    # a row-major sub-block of a bigger global matrix, with a diagonal shift (rand + 10I, runtests.jl:75)
    M, i0, j0 = 1000, 100, 90
    d2 = torch.zeros((50, 64), dtype=tdtype(dtype), device="cuda:0")
    h.call(f"rflu_fill_uniform_{sfx(dtype)}_dev", ptr(d2), 50, 60, 64, 1, 99, M, i0, j0, 10.0)
    torch.cuda.synchronize()
    full = O.np_uniform(M, 200, 99, dtype) + dtype(10) * np.eye(M, 200, dtype=dtype)
    assert np.array_equal(d2.cpu().numpy()[:, :60], full[i0:i0 + 50, j0:j0 + 60])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("geom", [(64, 0, 0, 64), (300, 0, 0, 64), (1000, 64, 64, 40), (5000, 128, 3, 64), (70, 0, 5, 7)])
def test_leaf_panel_matches_oracle_panel(dtype, geom):
    # _generic_lufact! (src/lu.jl:290-338) on a tall m x w block; same pivots (bit-exact), same factors to rounding
    m, r0, c0, w = geom
    ld = c0 + w + 9
    A = O.np_uniform(m, ld, 1234 + m, dtype)
    dA = to_dev_rm(A)
    dP = torch.zeros(m, dtype=torch.int64, device="cuda:0")
    info = ctypes.c_int64(-1)
    handle().call(f"rflu_panel_rm_{sfx(dtype)}_dev", m, r0, c0, w, ptr(dA), ld, ptr(dP), 1, ctypes.byref(info))
    torch.cuda.synchronize()
    F, ipiv, oinfo = O.generic_lufact(A[r0:, c0:c0 + w])
    got = dA.cpu().numpy()
    assert info.value == oinfo == 0
    assert np.array_equal(dP.cpu().numpy()[r0:r0 + w], ipiv + r0)
    scale = max(1.0, float(np.max(np.abs(F))))
    assert np.max(np.abs(got[r0:, c0:c0 + w] - F)) < 200 * np.finfo(dtype).eps * scale
    # nothing outside the panel's columns / above r0 is touched
    mask = np.ones_like(A, dtype=bool)
    mask[r0:, c0:c0 + w] = False
    assert np.array_equal(got[mask], A[mask])
