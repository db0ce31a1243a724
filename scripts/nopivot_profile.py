import ctypes, os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from recursivefactorization.jl_amd import _ffi
n = 16384
h = _ffi.Handle(0); h.set_stream(None)
A0 = torch.rand((n, n), dtype=torch.float64, device="cuda") + 10 * torch.eye(n, dtype=torch.float64, device="cuda")
info = ctypes.c_int64(0)
def run():
    A = A0.clone(); torch.cuda.synchronize(); t0 = time.perf_counter()
    h.call("rflu_getrf_rm_f64_dev", n, n, ctypes.c_void_p(A.data_ptr()), n, None, 0, 0, ctypes.byref(info))
    return (time.perf_counter() - t0) * 1e3
run(); print("nopivot plain: %.2f ms" % min(run() for _ in range(3)))
h.profile_enable(1); t = run(); pr = h.profile(); h.profile_enable(0)
print("single-stream profiled: %.2f ms" % t)
for k, v in pr.items():
    if v["launches"]: print(f"   {k:12s} {v['ms']:8.2f} ms  {v['launches']:5d} launches  avg {v['ms']*1e3/v['launches']:7.1f} us")
