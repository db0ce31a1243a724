"""Phase breakdown of the pair-leaf kernel between its two leaves (needs librflu_trace.so: scripts/build_trace_lib.sh)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from recursivefactorization.jl_amd import _ffi
_ffi.LIB_PATH = os.path.join(os.path.dirname(_ffi.LIB_PATH), "librflu_trace.so")
lib = _ffi.load()
lib.rflu_debug_panel_trace.restype = ctypes.c_int
lib.rflu_debug_panel_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
h = _ffi.Handle(0); h.set_stream(None)
P = lambda t: ctypes.c_void_p(t.data_ptr())
for m in [int(x) for x in (sys.argv[1:] or ["512", "4096", "16384"])]:
    A0 = torch.rand((m, 128), dtype=torch.float64, device="cuda"); ip = torch.zeros(m, dtype=torch.int64, device="cuda"); info = ctypes.c_int64(0)
    for _ in range(3):
        A = A0.clone()
        h.call("rflu_panel_rm_f64_dev", m, 0, 0, 128, P(A), 128, P(ip), 1, ctypes.byref(info))
    buf = np.zeros(528, dtype=np.int64)
    lib.rflu_debug_panel_trace(h.ptr, buf.ctypes.data)
    ex = buf[512:528].astype(np.float64)
    st = buf[:512].reshape(64, 8).astype(np.float64)
    names = ["publish L + store A + perm + load B + publish B", "gather", "trsm", "schur"]
    print(f"m={m}: leaf B steps {(st[63,6]-st[0,0])/2.4e3:.1f} us; between the leaves (thread 0 of WG 0):")
    for i, n in enumerate(names):
        print(f"   {n:50s} {(ex[5+i]-ex[4+i])/2.4e3:7.2f} us")
    print(f"   total {(ex[8]-ex[4])/2.4e3:.2f} us")
