"""Latency of the panel-recursion GEMMs (small N and K) -- standalone, back-to-back launches on one stream."""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recursivefactorization.jl_amd import _ffi
h = _ffi.Handle(0); h.set_stream(None)
P = lambda t: ctypes.c_void_p(t.data_ptr())
ld = 16384
R = torch.rand((16384 + 512, ld), dtype=torch.float64, device="cuda")
def bench(M, N, K, reps=200):
    a = R.data_ptr() + (512 * ld + 0) * 8      # A = R[512:512+M, 0:K]
    b = R.data_ptr() + (0 * ld + 512) * 8      # B = R[0:K, 512:512+N]
    c = R.data_ptr() + (512 * ld + 512) * 8    # C = R[512:512+M, 512:512+N]
    f = lambda: h.call("rflu_gemm_rm_f64_dev", M, N, K, ctypes.c_void_p(a), ld, ctypes.c_void_p(b), ld, ctypes.c_void_p(c), ld)
    for _ in range(5): f()
    h.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    h.synchronize(); t = (time.perf_counter() - t0) / reps
    print(f"M={M:6d} N={N:4d} K={K:4d}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:6.2f} TF", flush=True)
for M in (16384, 8192, 2048):
    for (N, K) in ((64, 64), (128, 128), (256, 256), (512, 512)):
        bench(M, N, K)
