"""Per-phase breakdown of the XCD-local panel kernel (workgroup 0: thread 0 = a row thread, lane 0 of the communication
wave; needs librflu_trace.so: scripts/build_trace_lib.sh)."""
import ctypes, sys, os
os.environ.setdefault("RFLU_PANEL_LOCAL", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from recursivefactorization.jl_amd import _ffi
_ffi.LIB_PATH = os.environ.get("RFLU_TRACE_LIB") or os.path.join(_ffi.HERE, "librflu_trace.so")
lib = _ffi.load()
lib.rflu_debug_panel_trace.restype = ctypes.c_int
lib.rflu_debug_panel_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
h = _ffi.Handle(0); h.set_stream(None)
P = lambda t: ctypes.c_void_p(t.data_ptr())
names = ["row thread: A -> bookkeeping done", "row thread: wave argmax + record + barrier B", "row thread: (comm publish call skipped)", "row thread: staging + deferred update loop", "row thread: wait for barrier A"]
for m in [int(x) for x in (sys.argv[1:] or ["1024", "4096", "16384"])]:
    A0 = torch.rand((m, 64), dtype=torch.float64, device="cuda"); ip = torch.zeros(m, dtype=torch.int64, device="cuda"); info = ctypes.c_int64(0)
    for _ in range(3):
        A = A0.clone()
        h.call("rflu_panel_rm_f64_dev", m, 0, 0, 64, P(A), 64, P(ip), 1, ctypes.byref(info))
    buf = np.zeros(528, dtype=np.int64)
    lib.rflu_debug_panel_trace(h.ptr, buf.ctypes.data)
    st = buf[:512].reshape(64, 8).astype(np.float64)
    d = np.diff(st[:, :6], axis=1)
    ks = slice(2, 60)
    print(f"m={m}: avg clock64 ticks per step (steps 2..59):")
    for i, n in enumerate(names):
        print(f"   {n:48s} {d[ks, i].mean():8.0f}")
    print(f"   row thread step (stamp 0 -> next stamp 0)        {np.diff(st[:, 0])[ks].mean():8.0f}")
    print(f"   comm wave: barrier B (row stamp 2) -> H(c+1) published   {(st[ks, 6] - st[ks, 2]).mean():8.0f}")
    print(f"   comm wave: H(c+1) published -> hand-over written         {(st[ks, 7] - st[ks, 6]).mean():8.0f}")
    print(f"   hand-over written -> row thread past barrier A (next 0)  {(st[3:61, 0] - st[2:60, 7]).mean():8.0f}")
