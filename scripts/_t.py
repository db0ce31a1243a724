import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import recursivefactorization.jl_amd as rf
from recursivefactorization.jl_amd import _ffi
n = int(os.environ.get("TN", "16384"))
A = torch.rand((n, n), dtype=torch.float64, device="cuda"); A0 = A.clone()
F = rf.lu_(A, None, True, check=False)
for nrhs in (33, 64, 128):
    B0 = torch.rand((n, nrhs), dtype=torch.float64, device="cuda")
    for mode in ("0", "1", "2"):
        os.environ["RFLU_TRSM_CHAIN_SPLIT"] = mode; _ffi.reload_tuning()
        X = B0.clone(); rf.ldiv_(F, X); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            X.copy_(B0); torch.cuda.synchronize(); t0 = time.perf_counter(); rf.ldiv_(F, X); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        res = ((A0 @ X - B0).norm() / B0.norm()).item()
        print(f"n={n} nrhs={nrhs} mode={mode}: {sorted(ts)[2]*1e3:.2f} ms (min {min(ts)*1e3:.2f}) residual {res:.2e}", flush=True)
