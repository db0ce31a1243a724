"""-m gpu: the multi-GPU C entry (rflu_getrf_*_mgpu) in "fake multi-GPU" mode -- k logical devices on the one physical GPU,
the panel broadcast replaced by a device-to-device copy: ownership, packing, message contents,
update order and lookahead of the 1-D block-column layout are exactly those of a real k-GPU run.  The reference has no
distributed path; parity is defined against the 1-GPU result and the CPU oracle: ipiv and info bit-exact, factors to
rounding (the trailing updates use the same kernels with different K splits)."""
import numpy as np
import pytest
import torch

import oracle as O
import recursivefactorization.jl_amd as rf
from helpers import rand_matrix
from recursivefactorization.jl_amd.multigpu import MultiGPU

pytestmark = pytest.mark.gpu


def tol_E(A):
    return 20 * A.shape[0] * np.finfo(A.dtype).eps


@pytest.mark.parametrize("k", [2, 4, 8])
@pytest.mark.parametrize("n,block,run", [(1000, 128, 1), (1536, 256, 1), (2048, 128, 2), (700, 64, 3)])
def test_fake_k_gpus_match_one_gpu(k, n, block, run):
    A = rand_matrix(n, n, seed=100 + n)
    mg = MultiGPU([0] * k)
    assert mg.fake and mg.ndev == k
    slabs, lds, layout = mg.alloc(n, torch.float64, block, run)
    mg.scatter(A, slabs, layout)
    ipiv, info = mg.getrf(n, slabs, lds, block, run, pivot=True)
    F1 = rf.lu(A, True, check=False)                 # the 1-GPU path
    Fo, ipo, infoo = O.lu(A)                         # the CPU oracle
    assert info == F1.info == infoo == 0
    assert np.array_equal(ipiv, np.asarray(F1.ipiv)), "k-GPU pivots must equal the 1-GPU pivots"
    assert np.array_equal(ipiv, ipo)
    LU = mg.gather(slabs, layout, n)
    scale = max(1.0, float(np.max(np.abs(Fo))))
    assert np.max(np.abs(LU - Fo)) < 50 * tol_E(A) * scale
    mx, fro = O.residual(A, LU, ipiv)
    assert fro < 1e-12 and mx < tol_E(A)
    mg.close()


@pytest.mark.parametrize("k", [2, 4])
def test_fake_k_gpus_device_fill_float32_nopivot_singular(k):
    n, block = 1200, 128
    mg = MultiGPU([0] * k)
    # device-side synthetic input = the same matrix as on one GPU (same counter-based generator)
    slabs, lds, layout = mg.alloc(n, torch.float64, block, 1)
    mg.fill_uniform(n, slabs, lds, block, 1, seed=12)
    A = O.np_uniform(n, n, 12)
    assert np.array_equal(mg.gather(slabs, layout, n), A)
    ipiv, info = mg.getrf(n, slabs, lds, block, 1, pivot=True)
    assert info == 0 and np.array_equal(ipiv, O.lu(A)[1])
    # Float32
    A32 = rand_matrix(n, n, seed=7, dtype=np.float32)
    s32, l32, lay32 = mg.alloc(n, torch.float32, block, 1)
    mg.scatter(A32, s32, lay32)
    ip32, info32 = mg.getrf(n, s32, l32, block, 1, pivot=True)
    assert info32 == 0 and np.array_equal(ip32, O.lu(A32)[1])
    mx, fro = O.residual(A32, mg.gather(s32, lay32, n), ip32)
    assert mx < tol_E(A32)
    # NoPivot on rand + 10I (test/runtests.jl:75): identity pivots, unpivoted bound
    D = (rand_matrix(n, n, seed=9) + 10 * np.eye(n)).astype(np.float64)
    sd, ld, layd = mg.alloc(n, torch.float64, block, 1)
    mg.scatter(D, sd, layd)
    ipd, infod = mg.getrf(n, sd, ld, block, 1, pivot=False)
    assert infod == 0 and np.array_equal(ipd, np.arange(1, n + 1))
    mx, fro = O.residual(D, mg.gather(sd, layd, n), ipd)
    assert mx < 10 * np.sqrt(tol_E(D))
    # singular: a zeroed column -> info = its index, like LAPACK (test/runtests.jl:59-64)
    S = rand_matrix(n, n, seed=11)
    S[:, 333] = 0
    ss, ls, lays = mg.alloc(n, torch.float64, block, 1)
    mg.scatter(S, ss, lays)
    ips, infos = mg.getrf(n, ss, ls, block, 1, pivot=True)
    assert infos == 334 == O.lu(S)[2]
    assert np.array_equal(ips, O.lu(S)[1])
    mg.close()


def test_one_logical_device_and_layout_query():
    mg = MultiGPU([0])
    assert not mg.fake
    n, block = 900, 128
    A = rand_matrix(n, n, seed=5)
    slabs, lds, layout = mg.alloc(n, torch.float64, block, 1)
    mg.scatter(A, slabs, layout)
    ipiv, info = mg.getrf(n, slabs, lds, block, 1)
    assert info == 0 and np.array_equal(ipiv, O.lu(A)[1])
    lib = mg.lib
    assert lib.rflu_mgpu_local_cols(1000, 128, 3, 1, 0) == 128 * 3 and lib.rflu_mgpu_local_cols(1000, 128, 3, 1, 2) == 256
    assert lib.rflu_mgpu_local_cols(1000, 100, 3, 1, 0) == 400   # block widths need not be multiples of 64 for the query
    assert lib.rflu_mgpu_local_cols(1000, 128, 3, 1, 5) == -1
    mg.close()


# ---- BASELINE configs 3 and 4 in THEIR layout at FULL size (fake multi-GPU: k logical devices on the one GPU of the box) --------
# "N=32768 Float64, 1-D block-column over 2 and 4 GPUs" and "N=65536 Float64 over 8": block 512, runs of 4 block columns per owner
# (what bench.py --gpus N uses).  Checked like tests/test_gpu_configs.py: info, the O(n^2) mat-vec residual below the north
# star's 1e-12, and ipiv equal to the 1-GPU factorization of the same generated matrix.
@pytest.mark.parametrize("n,k", [(32768, 2), (32768, 4), (65536, 8)])
def test_baseline_configs_3_4_in_their_layout_full_size(n, k, record_property):
    from gpu_util import fill_uniform_cm, matvec_residual_slabs

    block, run = 512, 4
    mg = MultiGPU([0] * k)
    slabs, lds, layout = mg.alloc(n, torch.float64, block, run)
    mg.fill_uniform(n, slabs, lds, block, run, seed=12)
    ipiv, info = mg.getrf(n, slabs, lds, block, run, pivot=True)
    assert info == 0
    res = matvec_residual_slabs(n, slabs, layout, ipiv, seed=12)
    record_property("residual", res)
    assert res < 1e-12, res
    del slabs
    mg.close()
    torch.cuda.empty_cache()
    A = fill_uniform_cm(n, np.float64, 12)
    F = rf.lu_(A, None, True, check=False)
    assert F.info == 0
    assert np.array_equal(ipiv, F.ipiv.cpu().numpy()), "k-GPU pivots must equal the 1-GPU pivots"
    del A, F
    torch.cuda.empty_cache()


def test_rccl_code_path_executes_with_a_one_rank_communicator():
    """A box of this pool has ONE GPU, so the collective of the multi-GPU driver (dlopen of librccl.so, ncclCommInitAll, the
    grouped ncclBroadcast pair of {panel, pivots} per block column on the library's panel stream, ordered against the CU-masked
    update streams) never runs in fake mode.  RFLU_MGPU_FORCE_RCCL=1 makes a one-device object build a one-rank communicator
    and broadcast every block column to itself: the same calls, on the same streams, as with k ranks.  In a subprocess: RCCL's
    runtime state should not leak into the other tests of this process."""
    import subprocess
    import sys
    code = r"""
import os, sys
os.environ["RFLU_MGPU_FORCE_RCCL"] = "1"
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import oracle as O
from recursivefactorization.jl_amd.multigpu import MultiGPU
n, block = 1536, 256
mg = MultiGPU([0])
assert not mg.fake and mg.ndev == 1
slabs, lds, layout = mg.alloc(n, torch.float64, block, 1)
mg.fill_uniform(n, slabs, lds, block, 1, seed=12)
A = O.np_uniform(n, n, 12)
ipiv, info = mg.getrf(n, slabs, lds, block, 1, pivot=True)
nblk = (n + block - 1) // block
assert mg.collectives == 2 * nblk, (mg.collectives, nblk)
Fo, ipo, info_o = O.lu(A)
assert info == info_o == 0 and np.array_equal(ipiv, ipo)
LU = mg.gather(slabs, layout, n)
mx, fro = O.residual(A, LU, ipiv)
assert fro < 1e-12, fro
# a second factorization on the same communicator, Float32
s32, l32, lay32 = mg.alloc(n, torch.float32, block, 1)
mg.fill_uniform(n, s32, l32, block, 1, seed=5)
ip32, info32 = mg.getrf(n, s32, l32, block, 1, pivot=True)
assert info32 == 0 and mg.collectives == 4 * nblk
mg.close()
print("RCCL-ONE-RANK-OK", 4 * nblk)
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL-ONE-RANK-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
