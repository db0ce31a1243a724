"""Per-phase breakdown of the one-workgroup leaf (panel_single.hip), thread 0; needs a librflu_trace.so whose panel_single.o was
compiled with -DRFLU_PANEL_TRACE."""
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
names = ["barrier -> records read, winner known", "division, own row, elimination C on a[C+1]", "wave argmax + record write", "perm bookkeeping", "deferred elimination loop", "owner row write"]
for m in [int(x) for x in (sys.argv[1:] or ["64", "256", "512"])]:
    A0 = torch.rand((m, 64), dtype=torch.float64, device="cuda"); ip = torch.zeros(m, dtype=torch.int64, device="cuda"); info = ctypes.c_int64(0)
    for _ in range(3):
        A = A0.clone()
        h.call("rflu_panel_rm_f64_dev", m, 0, 0, 64, P(A), 64, P(ip), 1, ctypes.byref(info))
    buf = np.zeros(528, dtype=np.int64)
    lib.rflu_debug_panel_trace(h.ptr, buf.ctypes.data)
    st = buf[:512].reshape(64, 8).astype(np.float64)
    d = np.diff(st[:, :7], axis=1)
    for ks, lab in [(slice(2, 20), "steps 2..19"), (slice(40, 60), "steps 40..59")]:
        print(f"m={m}: avg clock64 ticks per step ({lab}), thread 0:")
        for i, n in enumerate(names):
            print(f"   {n:48s} {d[ks, i].mean():8.0f}")
        print(f"   last stamp -> next step's first (barrier)        {(st[1:, 0] - st[:-1, 6])[ks].mean():8.0f}")
        print(f"   whole step                                       {np.diff(st[:, 0])[ks].mean():8.0f}")
