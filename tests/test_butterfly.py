"""The randomized butterfly pre-transform (/root/reference/src/butterflylu.jl), row f3 of SURVEY.md section 8.
CPU part: the NumPy restatement in oracle/ is pinned on the reference's own algebra (🦋mul! == U' A V with U, V from
materializeUV).  -m gpu part: butterfly.hip bit-exact against that restatement, and 🦋solve! on the reference's own test
(Wilkinson matrices n = 790..810, ||A x - b|| <= 1e-8 ||b||, test/runtests.jl:130-159)."""
import numpy as np
import pytest

import oracle as O
from helpers import rand_matrix, wilkinson


def _uv(n, dtype=np.float64, seed=888):
    from recursivefactorization.jl_amd.butterfly import generate_random

    return generate_random(n, dtype, seed)


@pytest.mark.parametrize("n", [4, 8, 20, 64])
def test_oracle_butterfly_mul_is_Ut_A_V(n):
    A = rand_matrix(n, n, seed=n)
    uv = _uv(n)
    assert np.all(uv > 0.47) and np.all(uv < 0.53)            # 0.5*exp(x), |x| <= 0.05
    U, V = O.butterfly_materialize_uv(uv, n)
    B = O.butterfly_mul(A.copy(), uv)
    assert np.allclose(B, U.T @ A @ V, rtol=1e-13, atol=1e-14)
    assert np.linalg.cond(U) < 50 and np.linalg.cond(V) < 50


def test_pad_matches_reference_shape():
    from recursivefactorization.jl_amd.butterfly import pad

    A = rand_matrix(7, 7, seed=3)
    P = pad(A)
    assert P.shape == (8, 8) and np.array_equal(P[:7, :7], A) and P[7, 7] == 1 and not P[7, :7].any() and not P[:7, 7].any()
    Q = pad(rand_matrix(9, 9, seed=3))
    assert Q.shape == (12, 12) and np.array_equal(Q[9:, 9:], np.eye(3))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n", [4, 8, 64, 260, 1000])
def test_gpu_butterfly_mul_bit_exact(n, dtype):
    import recursivefactorization.jl_amd as rf
    from gpu_util import to_dev_cm
    import torch

    A = rand_matrix(n, n, seed=10 + n, dtype=dtype)
    uv = _uv(n, dtype)
    want = O.butterfly_mul(A.copy(), uv)
    dA = to_dev_cm(A)
    rf.butterfly_mul_(dA, torch.from_numpy(uv).to("cuda:0"))
    assert np.array_equal(dA.cpu().numpy(), want), "same expressions in the same order: bit-exact"


@pytest.mark.gpu
def test_gpu_butterfly_vector_transforms():
    import torch
    from recursivefactorization.jl_amd.butterfly import _vec

    n = 256
    uv = _uv(n)
    U, V = O.butterfly_materialize_uv(uv, n)
    x = rand_matrix(n, 1, seed=77)[:, 0].copy()
    d_uv = torch.from_numpy(uv).to("cuda:0")
    got_u = _vec(torch.from_numpy(x.copy()).to("cuda:0"), d_uv, True).cpu().numpy()
    got_v = _vec(torch.from_numpy(x.copy()).to("cuda:0"), d_uv, False).cpu().numpy()
    assert np.allclose(got_u, U.T @ x, rtol=1e-13, atol=1e-14)
    assert np.allclose(got_v, V @ x, rtol=1e-13, atol=1e-14)


@pytest.mark.gpu
@pytest.mark.parametrize("n", list(range(790, 811, 4)) + [791, 801, 810])
def test_gpu_butterfly_solve_wilkinson(n):
    # test/runtests.jl:142-159: pivot-free LU of a Wilkinson matrix is hopeless (growth 2^n) unless the butterflies mix it
    import recursivefactorization.jl_amd as rf

    A = wilkinson(n)
    b = np.random.Generator(np.random.Philox(1234 + n)).random(n)
    ws = rf.ButterflyWorkspace(A.copy(), b.copy())
    out = rf.butterfly_solve_(ws, rf.Val(True))
    assert out.shape == (n,)
    assert rf.last_path() in ("hip-recursive", "hip-lookahead")
    assert isinstance(ws.F.ipiv, rf.NotIPIV) and ws.F.info == 0
    assert np.linalg.norm(A @ out - b) <= 1e-8 * np.linalg.norm(b)


@pytest.mark.gpu
def test_gpu_butterfly_solve_random_large():
    import recursivefactorization.jl_amd as rf

    n = 4098   # not a multiple of 4: exercises pad!
    A = rand_matrix(n, n, seed=5)
    b = rand_matrix(n, 1, seed=6)[:, 0].copy()
    out = rf.butterfly_solve_(rf.ButterflyWorkspace(A.copy(), b.copy()))
    # uniform[0,1) matrices are ill-conditioned (cond ~ 5e5, ||x|| >> ||b||): the meaningful bound is the normwise backward
    # error ||A x - b|| / (||A|| ||x|| + ||b||); measured 7e-11 (two butterfly levels + unpivoted LU)
    assert np.linalg.norm(A @ out - b) <= 1e-9 * (np.linalg.norm(A) * np.linalg.norm(out) + np.linalg.norm(b))
