"""laswp alone through the C ABI: 512 interchanges over a wide column range, with the pivot rows drawn from row windows of
different heights (how much of the rate is address translation / DRAM page locality rather than bytes) and two row strides."""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recursivefactorization.jl_amd import _ffi

h = _ffi.Handle(0)
h.set_stream(None)
P = lambda t: ctypes.c_void_p(t.data_ptr())
m, ncols, npiv = 16384, 16384, 512

def run(ld, span, reps=20):
    A = torch.rand((m, ld), dtype=torch.float64, device="cuda")
    ip = torch.arange(m, device="cuda") + 1
    ip[:npiv] = torch.randint(npiv, npiv + span, (npiv,), device="cuda") + 1
    call = lambda: h.call("rflu_laswp_rm_f64_dev", P(A), ld, m, 0, ncols, P(ip), 0, npiv)
    call(); h.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): call()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e-3
    return t

for ld in (16384, 16384 + 32):
    for span in (512, 2048, 8192, m - npiv):
        t = run(ld, span)
        print(f"ld={ld} pivot rows within {span:6d} rows: {t*1e6:8.1f} us  {32*npiv*ncols/t/1e12:6.2f} TB/s", flush=True)
