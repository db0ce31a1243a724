"""ldiv!(F, B) for blocks of right-hand sides: the cooperative MFMA chain (trsv.hip: trsm_chain_kernel, 33 .. 512 columns) against the
recursive splitting it replaces (RFLU_TRSM_CHAIN_MAX_RHS=0): residuals, agreement, time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import recursivefactorization.jl_amd as rf
from recursivefactorization.jl_amd import _ffi

def solve(F, B0, chain):
    os.environ["RFLU_TRSM_CHAIN_MAX_RHS"] = "320" if chain else "0"
    _ffi.reload_tuning()
    X = B0.clone()
    rf.ldiv_(F, X); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        X.copy_(B0); torch.cuda.synchronize(); t0 = time.perf_counter(); rf.ldiv_(F, X); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return X, sorted(ts)[1] * 1e3

ok = True
for dt in (torch.float64, torch.float32):
    for n in (1000, 4096, 5000, 16384):
        if dt == torch.float32 and n > 5000: continue
        A = torch.rand((n, n), dtype=dt, device="cuda")
        A0 = A.clone()
        F = rf.lu_(A, None, True, check=False)
        for nrhs in (33, 64, 100, 256):
            B0 = torch.rand((n, nrhs), dtype=dt, device="cuda")
            Xc, tc = solve(F, B0, True)
            Xr, tr = solve(F, B0, False)
            res_c = ((A0 @ Xc - B0).norm() / B0.norm()).item()
            res_r = ((A0 @ Xr - B0).norm() / B0.norm()).item()
            dx = ((Xc - Xr).norm() / Xr.norm()).item()
            good = res_c < 20 * max(res_r, 1e-15) and np.isfinite(res_c)
            ok &= good
            print(f"{'OK ' if good else 'BAD'} {str(dt)[6:]} n={n} nrhs={nrhs}: chain {tc:7.2f} ms residual {res_c:.2e} | recursive {tr:7.2f} ms residual {res_r:.2e} | dx {dx:.1e}", flush=True)
print("ALL OK" if ok else "FAILURES")
