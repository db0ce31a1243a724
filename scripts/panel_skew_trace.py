"""Skew between the workgroups of the pipelined panel kernel: wall-clock stamps (10 ns) of thread 0 of every workgroup
(needs librflu_trace_all.so: scripts/build_trace_all_lib.sh).  Per step: when did each workgroup enter the poll, see the last
header, pass the three barriers, finish its update; when was each workgroup's next header published."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from recursivefactorization.jl_amd import _ffi
_ffi.LIB_PATH = os.path.join(os.path.dirname(_ffi.LIB_PATH), "librflu_trace_all.so")
lib = _ffi.load()
lib.rflu_debug_panel_trace_all.restype = ctypes.c_int
lib.rflu_debug_panel_trace_all.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong]
h = _ffi.Handle(0); h.set_stream(None)
P = lambda t: ctypes.c_void_p(t.data_ptr())
m = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
G = (m + 511) // 512
A0 = torch.rand((m, 64), dtype=torch.float64, device="cuda"); ip = torch.zeros(m, dtype=torch.int64, device="cuda"); info = ctypes.c_int64(0)
for _ in range(3):
    A = A0.clone()
    h.call("rflu_panel_rm_f64_dev", m, 0, 0, 64, P(A), 64, P(ip), 1, ctypes.byref(info))
buf = np.zeros(32 * 65 * 8, dtype=np.int64)
n = lib.rflu_debug_panel_trace_all(h.ptr, buf.ctypes.data, buf.size)
assert n == buf.size, n
st = buf.reshape(32, 65, 8)[:G, :64, :].astype(np.float64) * 10.0   # ns
names = ["enter poll", "headers seen", "barrier 1", "search done", "barrier 3", "update done", "next header out"]
ks = range(8, 56)
print(f"m={m}, {G} workgroups; times in ns relative to the step's earliest poll entry; mean over steps 8..55")
for i, nme in enumerate(names):
    rel = np.array([st[:, k, i] - st[:, k, 0].min() for k in ks])       # steps x G
    print(f"  {nme:16s} earliest {rel.min(axis=1).mean():7.0f}  median {np.median(rel, axis=1).mean():7.0f}  latest {rel.max(axis=1).mean():7.0f}   (which workgroup is latest most often: {np.bincount(rel.argmax(axis=1), minlength=G).argmax()})")
step = np.array([st[:, k + 1, 0].min() - st[:, k, 0].min() for k in ks])
print(f"  step period {step.mean():.0f} ns")
# per step: latest header publication (slot 6 of step k is the header of column k+1) vs the moment everybody has seen the headers of k+1
lat = np.array([st[:, k + 1, 1].max() - st[:, k, 6].max() for k in ks])
print(f"  last header out -> last workgroup has seen all headers: {lat.mean():.0f} ns")
first = np.array([st[:, k + 1, 1].min() - st[:, k, 6].max() for k in ks])
print(f"  last header out -> first workgroup has seen all headers: {first.mean():.0f} ns")
spread = np.array([st[:, k, 6].max() - st[:, k, 6].min() for k in ks])
print(f"  spread of the header publications of one step: {spread.mean():.0f} ns")
