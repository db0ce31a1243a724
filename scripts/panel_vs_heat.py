"""Per-column time of the cooperative leaf (m x 64) alone vs next to register-only MFMA load on the other 224 CUs
(does the clock the power management grants a lightly loaded chip limit the latency-bound panel chain?)."""
import ctypes, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recursivefactorization.jl_amd import _ffi
P = lambda t: ctypes.c_void_p(t.data_ptr())
sp = torch.cuda.Stream()
hp = _ffi.Handle(0); hp.set_stream(sp.cuda_stream)
hh = _ffi.Handle(0); hh.set_stream(torch.cuda.Stream().cuda_stream)
info = ctypes.c_int64(0)
def leaves(m, reps=40):
    A0 = torch.rand((m, 64), dtype=torch.float64, device="cuda"); A = A0.clone()
    ip = torch.zeros(m, dtype=torch.int64, device="cuda")
    for it in range(reps + 5):
        if it == 5: hp.profile_enable(True)
        with torch.cuda.stream(sp): A.copy_(A0)
        hp.call("rflu_panel_rm_f64_dev", m, 0, 0, 64, P(A), 64, P(ip), 1, ctypes.byref(info))
    pr = hp.profile()["panel"]; hp.profile_enable(False)
    return pr["ms"] * 1e3 / pr["launches"]
stop = False
def heat():
    while not stop:
        hh.call("rflu_debug_heat", 2000.0); hh.call("rflu_debug_heat", 2000.0); time.sleep(0.003)
for m in (6144, 15872):
    a = leaves(m)
    stop = False; th = threading.Thread(target=heat); th.start(); time.sleep(0.05)
    b = leaves(m)
    stop = True; th.join(); torch.cuda.synchronize()
    c = leaves(m)
    print(f"m={m}: alone {a:7.1f} us/leaf ({a/64*1e3:5.0f} ns/col)   next to MFMA load {b:7.1f} ({b/64*1e3:5.0f})   alone again {c:7.1f}", flush=True)
