"""Leaf cadence of the leaf-wise schedule without a profiler attached: wall-clock stamps written by the gate signals of the
critical path (P) and the two side streams.  usage: RFLU_GATE_TRACE=1 python scripts/gate_trace.py [n] [first leaf] [count]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RFLU_GATE_TRACE"] = "1"; os.environ.setdefault("RFLU_LEAFWISE", "1")
import numpy as np, torch
from recursivefactorization.jl_amd import _ffi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
first = int(sys.argv[2]) if len(sys.argv) > 2 else 160
count = int(sys.argv[3]) if len(sys.argv) > 3 else 24
h = _ffi.Handle(0); h.set_stream(None)
A0 = torch.rand((n, n), dtype=torch.float64, device="cuda")
ip = torch.zeros(n, dtype=torch.int64, device="cuda"); info = ctypes.c_int64(0)
for it in range(3):
    A = A0.clone(); torch.cuda.synchronize(); t0 = time.perf_counter()
    h.call("rflu_getrf_rm_f64_dev", n, n, ctypes.c_void_p(A.data_ptr()), n, ctypes.c_void_p(ip.data_ptr()), 1, 0, ctypes.byref(info))
    ms = (time.perf_counter() - t0) * 1e3
st = np.zeros((3, 4096), dtype=np.int64)
h.call("rflu_debug_gate_stamps", ctypes.c_void_p(st.ctypes.data))
nl = n // 64
t = (st[:, :nl] - st[0, 0]) / 100.0   # us
print(f"n={n}: {ms:.2f} ms; P passes its last leaf at {t[0, nl-1]/1e3:.2f} ms")
print(" leaf   P signal (us)   dP      S1 lag   S2 lag")
for g in range(first, min(first + count, nl)):
    print(f"{g:5d} {t[0, g]:12.1f} {t[0, g] - t[0, g-1]:8.1f} {t[1, g] - t[0, g]:9.1f} {t[2, g] - t[0, g]:9.1f}")
d = np.diff(t[0])
for a, b in ((0, nl // 4), (nl // 4, nl // 2), (nl // 2, 3 * nl // 4), (3 * nl // 4, nl - 1)):
    print(f"leaves {a:4d}..{b:4d}: mean dP {d[a:b].mean():7.1f} us, median {np.median(d[a:b]):7.1f}, max {d[a:b].max():8.1f}")
