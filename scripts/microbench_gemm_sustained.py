"""Sustained large-K GEMM (about 10 s) so that the shader clock / power can be sampled next to it (scripts/clock_under_load.sh)."""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recursivefactorization.jl_amd import _ffi
h = _ffi.Handle(0); h.set_stream(None)
P = lambda t: ctypes.c_void_p(t.data_ptr())
M = N = K = 8192
A = torch.rand((M, K), dtype=torch.float64, device="cuda") - 0.5
B = (torch.rand((K, N), dtype=torch.float64, device="cuda") - 0.5) * 1e-3
C = torch.rand((M, N), dtype=torch.float64, device="cuda")
h.call("rflu_gemm_rm_f64_dev", M, N, K, P(A), K, P(B), N, P(C), N); h.synchronize()
print("start", flush=True)
t_all = time.perf_counter()
for rep in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        h.call("rflu_gemm_rm_f64_dev", M, N, K, P(A), K, P(B), N, P(C), N)
    h.synchronize(); t = (time.perf_counter() - t0) / 50
    print(f"t={time.perf_counter()-t_all:5.1f}s gemm 8192^3: {t*1e3:8.3f} ms {2*M*N*K/t/1e12:6.2f} TFLOP/s", flush=True)
