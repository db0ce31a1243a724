"""Per-phase breakdown of the chain wave of the sub-panel leaf (panel_blocked.hip), workgroup 0, lane 0; needs a
librflu_trace.so whose panel_blocked.o was compiled with -DRFLU_PANEL_TRACE (scripts/build_trace_lib.sh)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from recursivefactorization.jl_amd import _ffi
_ffi.LIB_PATH = os.environ.get("RFLU_TRACE_LIB") or os.path.join(_ffi.HERE, "librflu_trace.so")
lib = _ffi.load()
lib.rflu_debug_panel_trace.restype = ctypes.c_int
lib.rflu_debug_panel_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
h = _ffi.Handle(0); h.set_stream(None)
P = lambda t: ctypes.c_void_p(t.data_ptr())
names = ["decide (poll, reduce, divide)", "eliminate (all row slots)", "retire bookkeeping", "search + header", "deferred rest", "pivot row save", "row record"]
for m in [int(x) for x in (sys.argv[1:] or ["448", "4096", "14336"])]:
    A0 = torch.rand((m, 64), dtype=torch.float64, device="cuda"); ip = torch.zeros(m, dtype=torch.int64, device="cuda"); info = ctypes.c_int64(0)
    for _ in range(3):
        A = A0.clone()
        h.call("rflu_panel_rm_f64_dev", m, 0, 0, 64, P(A), 64, P(ip), 1, ctypes.byref(info))
    buf = np.zeros(528, dtype=np.int64)
    lib.rflu_debug_panel_trace(h.ptr, buf.ctypes.data)
    st = buf[:512].reshape(64, 8).astype(np.float64)
    d = np.diff(st, axis=1)            # within a step
    nxt = st[1:, 0] - st[:-1, 7]       # last stamp -> next step's first (sub-panel boundary every 8th)
    inner = [c for c in range(63) if c % 8 != 7]
    bnd = [c for c in range(63) if c % 8 == 7]
    print(f"m={m}: avg clock64 ticks, chain wave of workgroup 0 (steps inside sub-panels):")
    for i, n in enumerate(names):
        per_i = " ".join(f"{d[[c for c in range(64) if c % 8 == ii], i].mean():6.0f}" for ii in range(8))
        print(f"   {n:34s} {d[:, i].mean():7.0f}   by local column: {per_i}")
    print(f"   step end -> next step start        {nxt[inner].mean():7.0f}")
    print(f"   whole step (inside a sub-panel)    {np.diff(st[:, 0])[inner].mean():7.0f}")
    print(f"   sub-panel boundary (step 7 end -> next step 0 start) {nxt[bnd].mean():7.0f}")
    print(f"   leaf: first stamp -> last stamp    {st[63, 7] - st[0, 0]:9.0f}")
