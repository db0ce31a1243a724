"""Worker of tests/test_gpu_multiproc.py: the multi-GPU block-column driver with the REAL HIP building blocks and streams,
two processes sharing one MI355X, collectives over gloo (RCCL refuses two ranks on one device).  Launched by
`python -m torch.distributed.run --nproc-per-node 2 tests/mp_gpu_worker.py`; rank 0 prints MP_GPU_OK when every case holds."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import oracle as O  # noqa: E402  (the checker)
from recursivefactorization.jl_amd import _ffi  # noqa: E402
from recursivefactorization.jl_amd.distributed import BlockColumnLU, HipOps  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    h = _ffi.Handle(0)
    h.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    # (n, block, pivot, tall_rows, sync, run)
    cases = [(1536, 256, True, None, False, 1), (1536, 256, True, 700, False, 1), (1000, 128, True, 0, False, 1),
             (1200, 256, False, None, False, 1), (1536, 256, True, None, True, 1),
             (2048, 128, True, None, False, 2), (2048, 128, True, 900, False, 4)]
    for (n, block, pivot, tall_rows, sync, run) in cases:
        os.environ["RFLU_DIST_SYNC"] = "1" if sync else "0"
        diag = 0.0 if pivot else 10.0
        job = BlockColumnLU(HipOps(h, "f64"), n, torch.float64, rank, world, dev, block=block, pivot=pivot, seed=12,
                            diag_add=diag, run=run)
        if tall_rows is not None:
            job.tall_rows = tall_rows
        for rep in range(2):  # twice: buffers, events and streams are reused
            job.regenerate()
            torch.cuda.synchronize(dev)
            dist.barrier()
            info = job.factor()
            torch.cuda.synchronize(dev)
        F = job.gather_factors()
        res = job.matvec_residual()
        A = O.np_uniform(n, n, 12) + diag * np.eye(n)
        Fo, ipo, infoo = O.lu(A, pivot=pivot)
        assert info == infoo == 0, (info, infoo)
        assert np.array_equal(job.ipiv.cpu().numpy(), ipo), "pivots must be bit-exact on every rank"
        tol = 50 * 20 * n * np.finfo(np.float64).eps if pivot else 10 * np.sqrt(20 * n * np.finfo(np.float64).eps)
        assert np.max(np.abs(F - Fo)) < tol * max(1.0, float(np.max(np.abs(Fo)))), np.max(np.abs(F - Fo))
        assert res < 1e-12, res
    dist.barrier()
    if rank == 0:
        print("MP_GPU_OK", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
