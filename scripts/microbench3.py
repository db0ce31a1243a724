import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recursivefactorization.jl_amd import _ffi
h = _ffi.Handle(0); h.set_stream(None)
P = lambda t: ctypes.c_void_p(t.data_ptr())
def med(fn, reps=7):
    fn(); h.synchronize(); ts=[]
    for _ in range(reps):
        torch.cuda.synchronize(); t0=time.perf_counter(); fn(); h.synchronize(); ts.append(time.perf_counter()-t0)
    return sorted(ts)[len(ts)//2]
for m in (256, 512, 2048, 16384):
    for w in (1, 8, 16, 32, 64):
        ld = 64
        A0 = torch.rand((m, ld), dtype=torch.float64, device="cuda"); A = A0.clone()
        ip = torch.zeros(m, dtype=torch.int64, device="cuda"); info = ctypes.c_int64(0)
        def run():
            A.copy_(A0); h.call("rflu_panel_rm_f64_dev", m, 0, 0, w, P(A), ld, P(ip), 1, ctypes.byref(info))
        t = med(run) - med(lambda: A.copy_(A0))
        print(f"panel m={m:6d} w={w:3d}: {t*1e6:8.1f} us", flush=True)
