"""Per-kernel-class time of one factorization measured INSIDE the multi-stream schedule (rflu_profile_enable(2): event pairs
around every launch on whatever stream it runs).  usage: python scripts/inschedule_profile.py [n]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recursivefactorization.jl_amd import _ffi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
h = _ffi.Handle(0); h.set_stream(None)
A0 = torch.rand((n, n), dtype=torch.float64, device="cuda")
ip = torch.zeros(n, dtype=torch.int64, device="cuda"); info = ctypes.c_int64(0)
def run():
    A = A0.clone(); torch.cuda.synchronize(); t0 = time.perf_counter()
    h.call("rflu_getrf_rm_f64_dev", n, n, ctypes.c_void_p(A.data_ptr()), n, ctypes.c_void_p(ip.data_ptr()), 1, 0, ctypes.byref(info))
    return (time.perf_counter() - t0) * 1e3
run(); print("plain      : %.2f ms" % min(run() for _ in range(3)))
h.profile_enable(2)
t = run()
pr = h.profile(); h.profile_enable(0)
print("with events: %.2f ms" % t)
for k, v in pr.items():
    if v["launches"]: print(f"   {k:12s} {v['ms']:8.2f} ms  {v['launches']:5d} launches  avg {v['ms']*1e3/v['launches']:7.1f} us")
