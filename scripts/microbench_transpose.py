import ctypes, sys, time
sys.path.insert(0, "/root/repo")
import torch
from recursivefactorization.jl_amd import _ffi
h = _ffi.Handle(0); h.set_stream(None)
for n in (16384, 16400, 8192):
    A = torch.rand((n, n), dtype=torch.float64, device="cuda"); R = torch.empty_like(A)
    f = lambda: h.call("rflu_cm_to_rm_f64_dev", n, n, ctypes.c_void_p(A.data_ptr()), n, ctypes.c_void_p(R.data_ptr()), n)
    for _ in range(3): f()
    h.synchronize(); t0 = time.perf_counter()
    for _ in range(20): f()
    h.synchronize(); t = (time.perf_counter() - t0) / 20
    ok = torch.equal(R, A.t().contiguous()) or torch.equal(R.t().contiguous(), A)
    print(f"n={n}: {t*1e3:.3f} ms  {2*8*n*n/t/1e12:.2f} TB/s  correct={ok}")
