"""CPU (gloo, world_size 2 and 3) test of the multi-GPU block-column orchestration in
recursivefactorization.jl_amd/distributed.py: ownership, the one broadcast per block column (panel + pivots + info),
interchanges on left/right local columns, TRSM + GEMM on the trailing slab.  The kernels are replaced by a NumPy
stand-in for ``ops`` (test infrastructure, built on the oracle); the product path plugs ``HipOps`` into the same code.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle as O


def _view(t, off, rows, cols, ld):
    flat = t.numpy().reshape(-1)
    return np.lib.stride_tricks.as_strided(flat[off:], shape=(rows, cols), strides=(ld * flat.itemsize, flat.itemsize))


class NumpyOps:
    """Restatement of the four kernels on host slabs (row-major), for the orchestration test only."""

    def panel(self, R, ld, m, r0, c0, w, ipiv, pivot):
        A = _view(R, r0 * ld + c0, m - r0, w, ld)
        F, ip, info = O.generic_lufact(np.array(A), pivot)
        A[:] = F
        if pivot:
            ipiv.numpy()[r0:r0 + w] = ip + r0
        return info + r0 if info else 0

    def laswp(self, R, ld, m, c0, ncols, ipiv, k0, k1):
        if ncols <= 0:
            return
        A = _view(R, c0, m, ncols, ld)
        p = ipiv.numpy()
        for k in range(k0, k1):
            t = int(p[k]) - 1
            if t != k:
                A[[k, t]] = A[[t, k]]

    def trsm(self, n, nrhs, L, l_off, ldl, B, b_off, ldb):
        Lm = np.tril(np.array(_view(L, l_off, n, n, ldl), dtype=np.float64), -1) + np.eye(n)
        Bv = _view(B, b_off, n, nrhs, ldb)
        Bv[:] = np.linalg.solve(Lm, Bv.astype(np.float64)).astype(Bv.dtype)

    def gemm(self, M, N, K, A, a_off, lda, B, b_off, ldb, C, c_off, ldc):
        Cv = _view(C, c_off, M, N, ldc)
        Cv -= _view(A, a_off, M, K, lda) @ _view(B, b_off, K, N, ldb)

    def fill(self, R, ld, m, w, c0, seed, n_global, j0, diag_add=0.0):
        full = O.np_uniform(n_global, n_global, seed, R.numpy().dtype)
        blk = np.array(full[:m, j0:j0 + w])
        for jj in range(w):
            if j0 + jj < m:
                blk[j0 + jj, jj] += diag_add
        _view(R, c0, m, w, ld)[:] = blk


def _worker(rank, world, port, n, block, pivot, diag_add, q, sync=False, tall_rows=None, run=1):
    if sync:
        os.environ["RFLU_DIST_SYNC"] = "1"
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from recursivefactorization.jl_amd.distributed import BlockColumnLU

        job = BlockColumnLU(NumpyOps(), n, torch.float64, rank, world, torch.device("cpu"), block=block, pivot=pivot,
                            seed=12, diag_add=diag_add, run=run)
        if tall_rows is not None:
            job.tall_rows = tall_rows
        job.regenerate()
        info = job.factor()
        F = job.gather_factors()
        res = job.matvec_residual()
        if rank == 0:
            q.put((F, job.ipiv.numpy().copy(), info, res))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,n,block,pivot,sync,tall_rows,run", [
    (2, 300, 64, True, False, None, 1), (3, 257, 64, True, False, None, 1), (2, 200, 128, False, False, None, 1),
    (2, 300, 64, True, True, None, 1),
    (2, 300, 64, True, False, 150, 1),   # block columns with more than 150 rows take the tall-panel order, the rest overlap
    (3, 321, 64, True, False, 0, 1),     # every block column tall
    (2, 450, 64, True, False, None, 2),  # runs of two consecutive block columns per owner (broadcasts off the chain)
    (3, 500, 64, True, False, 200, 3),   # runs of three, tall and ordinary block columns
    (2, 450, 64, True, True, None, 2),   # the synchronous schedule on the same layout
])
def test_block_column_lu_matches_single_process(world, n, block, pivot, sync, tall_rows, run):
    # sync=False: the lookahead schedule (panel b+1 factored and broadcast while update b is still queued; a tall panel is
    # factored by its owner before that owner's bulk update);  sync=True : one block column at a time.
    # Same collectives in the same order on every rank in all of them.
    diag_add = 0.0 if pivot else 10.0
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, block, pivot, diag_add, q, sync, tall_rows, run))
             for port in [_free_port()] for r in range(world)]
    for p in procs:
        p.start()
    F, ipiv, info, res = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    A = O.np_uniform(n, n, 12) + diag_add * np.eye(n)
    Fo, ipo, infoo = O.lu(A, pivot=pivot)
    assert info == infoo == 0
    assert np.array_equal(ipiv, ipo)          # k ranks vs 1 process: pivots bit-exact
    assert np.max(np.abs(F - Fo)) < 1e-10
    assert O.residual(A, F, ipiv)[1] < 1e-13
    assert res < 1e-12


def test_block_layout_is_cyclic_and_complete():
    from recursivefactorization.jl_amd.distributed import block_layout

    layout, local = block_layout(1000, 128, 3)
    assert [o for (_, _, o, _) in layout] == [0, 1, 2, 0, 1, 2, 0, 1]
    assert sum(w for (_, w, _, _) in layout) == 1000 and sum(local) == 1000
    assert layout[3] == (384, 128, 0, 128) and layout[7] == (896, 104, 1, 256)
    # runs of two consecutive block columns per owner
    layout2, local2 = block_layout(1000, 128, 3, run=2)
    assert [o for (_, _, o, _) in layout2] == [0, 0, 1, 1, 2, 2, 0, 0]
    assert sum(local2) == 1000 and layout2[1] == (128, 128, 0, 128) and layout2[6] == (768, 128, 0, 256)
