"""Micro-benchmarks of the individual kernels through the C ABI (run on the GPU box)."""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recursivefactorization.jl_amd import _ffi

h = _ffi.Handle(0)
h.set_stream(None)
P = lambda t: ctypes.c_void_p(t.data_ptr())

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); h.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts)//2]

which = sys.argv[1:] or ["panel", "gemm", "trsm", "laswp"]
if "panel" in which:
    for m in (256, 512, 1024, 4096, 16384, 65536):
        ld = 64
        A0 = torch.rand((m, ld), dtype=torch.float64, device="cuda")
        A = A0.clone(); ip = torch.zeros(m, dtype=torch.int64, device="cuda"); info = ctypes.c_int64(0)
        def run():
            A.copy_(A0)
            h.call("rflu_panel_rm_f64_dev", m, 0, 0, 64, P(A), ld, P(ip), 1, ctypes.byref(info))
        def base():
            A.copy_(A0); 
        t = timeit(run) - timeit(base)
        print(f"panel m={m:6d} G={max(1,(m+255)//256):4d}: {t*1e6:9.1f} us  -> {t*1e6/64:6.2f} us/column", flush=True)
if "gemm" in which:
    for (M, N, K) in ((8192, 8192, 8192), (4096, 4096, 4096), (16384, 16384, 256), (16384, 16384, 64), (12288, 4096, 4096), (2048, 2048, 2048), (1024, 1024, 1024), (64, 8192, 64), (1024, 1024, 64)):
        A = torch.rand((M, K), dtype=torch.float64, device="cuda") - 0.5
        B = torch.rand((K, N), dtype=torch.float64, device="cuda") - 0.5
        C = torch.rand((M, N), dtype=torch.float64, device="cuda")
        t = timeit(lambda: h.call("rflu_gemm_rm_f64_dev", M, N, K, P(A), K, P(B), N, P(C), N))
        print(f"gemm {M}x{N}x{K}: {t*1e3:8.3f} ms  {2*M*N*K/t/1e12:6.2f} TFLOP/s", flush=True)
if "trsm" in which:
    for (n, nrhs) in ((64, 8192), (64, 1024), (64, 128), (256, 8192), (1024, 8192), (8192, 8192)):
        L = torch.rand((n, n), dtype=torch.float64, device="cuda") * 0.1
        B = torch.rand((n, nrhs), dtype=torch.float64, device="cuda")
        t = timeit(lambda: h.call("rflu_trsm_rm_f64_dev", n, nrhs, P(L), n, P(B), nrhs))
        print(f"trsm n={n} nrhs={nrhs}: {t*1e3:8.3f} ms  {n*n*nrhs/t/1e12:6.2f} TFLOP/s", flush=True)
if "laswp" in which:
    m = 16384
    for ncols in (16384, 4096, 512):
        A = torch.rand((m, ncols), dtype=torch.float64, device="cuda")
        ip = (torch.arange(m, device="cuda") + 1)
        ip[:64] = torch.randint(64, m, (64,), device="cuda") + 1
        t = timeit(lambda: h.call("rflu_laswp_rm_f64_dev", P(A), ncols, m, 0, ncols, P(ip), 0, 64))
        print(f"laswp 64 pivots x {ncols} cols: {t*1e6:8.1f} us  {32*64*ncols/t/1e9:8.1f} GB/s", flush=True)
