import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recursivefactorization.jl_amd import _ffi
h = _ffi.Handle(0); h.set_stream(None)
P = lambda t: ctypes.c_void_p(t.data_ptr())
def per_launch(fn, reps=50):
    fn(); h.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    h.synchronize()
    return (time.perf_counter() - t0) / reps
for (M, N, K) in ((128,128,16),(128,128,64),(128,128,256),(128,128,1024),(128,128,4096),(64,8192,64),(64,8192,256),(4096,128,128),(16384,64,64),(1024,1024,1024),(2048,2048,2048)):
    A = torch.rand((M, K), dtype=torch.float64, device="cuda") - 0.5
    B = torch.rand((K, N), dtype=torch.float64, device="cuda") - 0.5
    C = torch.rand((M, N), dtype=torch.float64, device="cuda")
    t = per_launch(lambda: h.call("rflu_gemm_rm_f64_dev", M, N, K, P(A), K, P(B), N, P(C), N))
    print(f"gemm {M}x{N}x{K}: {t*1e6:9.1f} us/launch  {2*M*N*K/t/1e12:6.2f} TF", flush=True)
for (n, nrhs) in ((64, 128), (64, 8192), (64,16384)):
    L = torch.rand((n, n), dtype=torch.float64, device="cuda") * 0.1
    B = torch.rand((n, nrhs), dtype=torch.float64, device="cuda")
    t = per_launch(lambda: h.call("rflu_trsm_rm_f64_dev", n, nrhs, P(L), n, P(B), nrhs))
    print(f"trsm base n={n} nrhs={nrhs}: {t*1e6:9.1f} us/launch", flush=True)
m = 16384
for ncols in (16384, 512):
    A = torch.rand((m, ncols), dtype=torch.float64, device="cuda")
    ip = (torch.arange(m, device="cuda") + 1); ip[:64] = torch.randint(64, m, (64,), device="cuda") + 1
    t = per_launch(lambda: h.call("rflu_laswp_rm_f64_dev", P(A), ncols, m, 0, ncols, P(ip), 0, 64))
    print(f"laswp(+perm_build) 64 pivots x {ncols} cols: {t*1e6:9.1f} us/launch-pair", flush=True)
x = torch.zeros(1024, device="cuda")
t = per_launch(lambda: x.add_(1.0))
print(f"torch tiny kernel: {t*1e6:.1f} us/launch")
