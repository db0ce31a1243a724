"""Back-to-back GEMM launches (no host sync in between): does the sustained rate differ from the single-launch rate?
usage: microbench_gemm_sustained.py [M=N [K [f64|f32]]]"""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recursivefactorization.jl_amd import _ffi
h = _ffi.Handle(0); h.set_stream(None)
P = lambda t: ctypes.c_void_p(t.data_ptr())
M = N = int(sys.argv[1]) if len(sys.argv) > 1 else 15872
K = int(sys.argv[2]) if len(sys.argv) > 2 else 512
sfx = sys.argv[3] if len(sys.argv) > 3 else "f64"
dt = torch.float64 if sfx == "f64" else torch.float32
C = torch.rand((M, N), dtype=dt, device="cuda")
A = torch.rand((M, K), dtype=dt, device="cuda") - 0.5
B = torch.rand((K, N), dtype=dt, device="cuda") - 0.5
fn = lambda: h.call(f"rflu_gemm_rm_{sfx}_dev", M, N, K, P(A), K, P(B), N, P(C), N)
fn(); h.synchronize()
for reps in (1, 1, 5, 20, 60, 1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    h.synchronize(); t = (time.perf_counter() - t0) / reps
    print(f"{reps:3d} back-to-back: {t*1e3:8.3f} ms each  {2*M*N*K/t/1e12:6.2f} TFLOP/s", flush=True)
