import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recursivefactorization.jl_amd import _ffi
m = int(sys.argv[1]) if len(sys.argv) > 1 else 256
w = int(sys.argv[2]) if len(sys.argv) > 2 else 64
h = _ffi.Handle(0); h.set_stream(None)
P = lambda t: ctypes.c_void_p(t.data_ptr())
A0 = torch.rand((m, 64), dtype=torch.float64, device="cuda"); ip = torch.zeros(m, dtype=torch.int64, device="cuda"); info = ctypes.c_int64(0)
for _ in range(5):
    A = A0.clone()
    h.call("rflu_panel_rm_f64_dev", m, 0, 0, w, P(A), 64, P(ip), 1, ctypes.byref(info))
torch.cuda.synchronize()
