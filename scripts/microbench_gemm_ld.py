"""GEMM shapes as issued inside the N=16384 factorization: all operands are views of ONE row-major matrix with row stride
LD (16384 doubles = 128 KiB, a power of two) versus a padded stride."""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recursivefactorization.jl_amd import _ffi
h = _ffi.Handle(0); h.set_stream(None)
K = 512
def run(n, ld, reps=30):
    R = torch.rand((n, ld), dtype=torch.float64, device="cuda") - 0.5
    base = R.data_ptr()
    A = base + (K * ld) * 8            # rows K.., cols 0..K
    B = base + K * 8                   # rows 0..K, cols K..
    C = base + (K * ld + K) * 8
    M = N = n - K
    fn = lambda: h.call("rflu_gemm_rm_f64_dev", M, N, K, ctypes.c_void_p(A), ld, ctypes.c_void_p(B), ld, ctypes.c_void_p(C), ld)
    for _ in range(20): fn()
    h.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    h.synchronize(); t = (time.perf_counter() - t0) / reps
    print(f"n={n} ld={ld}: {t*1e3:8.3f} ms  {2*M*N*K/t/1e12:6.2f} TFLOP/s", flush=True)
    del R
for ld in (16384, 16384 + 16, 16384 + 64, 16384 + 256, 16384 + 512 + 16):
    run(16384, ld)
