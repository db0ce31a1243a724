"""ldiv!(F, B) on the device (row-major factors from lu_ on a contiguous tensor, row-major B): time per solve."""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import recursivefactorization.jl_amd as rf
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
A = torch.rand((n, n), dtype=torch.float64, device="cuda")          # a contiguous tensor is taken as the row-major matrix
A0 = A.clone()
F = rf.lu_(A, None, True, check=False)
for nrhs in (1, 64, 1024):
    X = torch.rand((n, nrhs), dtype=torch.float64, device="cuda")
    B0 = X.clone()
    rf.ldiv_(F, X); torch.cuda.synchronize()
    R = A0 @ X - B0
    res = (R.norm() / B0.norm()).item()
    ts = []
    for _ in range(5):
        X.copy_(B0); torch.cuda.synchronize(); t0 = time.perf_counter(); rf.ldiv_(F, X); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    t = sorted(ts)[2]
    print(f"n={n} nrhs={nrhs:5d}: {t*1e3:8.2f} ms  ({2.0*n*n*nrhs/t/1e9:9.1f} GFLOP/s)  relative residual {res:.2e}", flush=True)
