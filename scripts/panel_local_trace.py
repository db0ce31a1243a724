"""Per-phase breakdown of the XCD-local panel kernel (thread 0 of workgroup 0; needs librflu_trace.so: scripts/build_trace_lib.sh)."""
import ctypes, sys, os
os.environ.setdefault("RFLU_PANEL_LOCAL", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from recursivefactorization.jl_amd import _ffi
_ffi.LIB_PATH = os.path.join(os.path.dirname(_ffi.LIB_PATH), "librflu_trace.so")
lib = _ffi.load()
lib.rflu_debug_panel_trace.restype = ctypes.c_int
lib.rflu_debug_panel_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
h = _ffi.Handle(0); h.set_stream(None)
P = lambda t: ctypes.c_void_p(t.data_ptr())
names = ["poll + reduce + LDS hand-over -> barrier A", "bookkeeping + division + wave search", "row arrival -> barrier B", "combine (end of mid)", "update / publish (caller)"]
for m in [int(x) for x in (sys.argv[1:] or ["1024", "4096", "16384"])]:
    A0 = torch.rand((m, 64), dtype=torch.float64, device="cuda"); ip = torch.zeros(m, dtype=torch.int64, device="cuda"); info = ctypes.c_int64(0)
    for _ in range(3):
        A = A0.clone()
        h.call("rflu_panel_rm_f64_dev", m, 0, 0, 64, P(A), 64, P(ip), 1, ctypes.byref(info))
    buf = np.zeros(528, dtype=np.int64)
    lib.rflu_debug_panel_trace(h.ptr, buf.ctypes.data)
    st = buf[:512].reshape(64, 8)[:, :6].astype(np.float64)
    d = np.diff(st, axis=1)
    gap = st[1:, 0] - st[:-1, 5]
    print(f"m={m}: avg clock64 ticks per step (thread 0 of workgroup 0):")
    for i, n in enumerate(names):
        print(f"   {n:48s} {d[1:62, i].mean():8.0f}   (k=1: {d[1, i]:6.0f}, k=32: {d[32, i]:6.0f}, k=61: {d[61, i]:6.0f})")
    print(f"   step total {(st[1:62, 5] - st[1:62, 0]).mean():8.0f}; gap to next step {gap[1:61].mean():6.0f}; steps 0..63: {(st[63, 5] - st[0, 0]):.0f} ticks")
