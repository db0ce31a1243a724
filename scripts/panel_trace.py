"""Per-phase cycle breakdown of the pivoted panel kernel (needs librflu_trace.so built with -DRFLU_PANEL_TRACE)."""
import ctypes, sys, os
os.environ["RFLU_PIPE"] = "0"   # this script reads the stamps of the two-trip kernel (panel_pipe_trace.py: the pipelined one)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from recursivefactorization.jl_amd import _ffi
_ffi.LIB_PATH = os.path.join(os.path.dirname(_ffi.LIB_PATH), "librflu_trace.so")
lib = _ffi.load()
lib.rflu_debug_panel_trace.restype = ctypes.c_int
lib.rflu_debug_panel_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
h = _ffi.Handle(0); h.set_stream(None)
P = lambda t: ctypes.c_void_p(t.data_ptr())
names = ["a:front+bar", "a:combine+hdr", "rowpub+poll+reduce", "rowfetch+bar2", "b:tail", "update", "step total"]
for m in [int(x) for x in (sys.argv[1:] or ["256", "2048", "16384"])]:
    A0 = torch.rand((m, 64), dtype=torch.float64, device="cuda"); ip = torch.zeros(m, dtype=torch.int64, device="cuda"); info = ctypes.c_int64(0)
    for _ in range(3):
        A = A0.clone()
        h.call("rflu_panel_rm_f64_dev", m, 0, 0, 64, P(A), 64, P(ip), 1, ctypes.byref(info))
    buf = np.zeros(528, dtype=np.int64)
    lib.rflu_debug_panel_trace(h.ptr, buf.ctypes.data)
    ex = buf[512:516].astype(np.float64)
    st = buf[:512].reshape(64, 8)[:, :7].astype(np.float64)
    d = np.diff(st, axis=1)
    tot = st[:, 6] - st[:, 0]
    gap = st[1:, 0] - st[:-1, 6]
    print(f"m={m}: avg cycles per step (thread 0 of WG 0):")
    for i, n in enumerate(names[:6]):
        print(f"   {n:10s} {d[:, i].mean():8.0f}   (k=0: {d[0, i]:6.0f}, k=32: {d[32, i]:6.0f}, k=63: {d[63, i]:6.0f})")
    print(f"   kernel entry->rows loaded {ex[1]-ex[0]:.0f} cyc; loaded->step0 {st[0,0]-ex[1]:.0f}; last step->store start {ex[2]-st[63,6]:.0f}; store {ex[3]-ex[2]:.0f}; entry->end {(ex[3]-ex[0])/2.4e3:.1f} us")
    print(f"   {'total':10s} {tot.mean():8.0f}  inter-step gap {gap.mean():6.0f};  whole kernel steps {(st[63,6]-st[0,0]):.0f} cycles = {(st[63,6]-st[0,0])/2.4e3:.1f} us")
