import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recursivefactorization.jl_amd import _ffi
h = _ffi.Handle(0); h.set_stream(None)
P = lambda t: ctypes.c_void_p(t.data_ptr())
def timeit(fn, reps=5):
    fn(); h.synchronize(); ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); h.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts)//2]
M = N = int(sys.argv[1]) if len(sys.argv) > 1 else 15360
C = torch.rand((M, N), dtype=torch.float64, device="cuda")
LDX = int(os.environ.get('GEMM_LD', '0'))
for K in [int(x) for x in os.environ.get('GEMM_KS', '256,512,1024,2048').split(',')]:
    lda = LDX or K   # GEMM_LD: leading dimension of A as inside the factorization (a block column of the big matrix)
    A = torch.rand((M, lda), dtype=torch.float64, device="cuda") - 0.5
    B = torch.rand((K, N), dtype=torch.float64, device="cuda") - 0.5
    t = timeit(lambda: h.call("rflu_gemm_rm_f64_dev", M, N, K, P(A), lda, P(B), N, P(C), N))
    print(f"gemm {M}x{N}x{K}: {t*1e3:9.3f} ms  {2*M*N*K/t/1e12:6.2f} TFLOP/s   per-K {t*1e6/K:7.3f} us", flush=True)
