"""One leaf at an offset (r0, c0) inside a wider matrix: the sub-panel kernel against the older leaves, bit for bit."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recursivefactorization.jl_amd import _ffi
P = lambda t: ctypes.c_void_p(t.data_ptr())
res = {}
for blocked in (0, 1):
    os.environ["RFLU_PANEL_BLOCKED"] = str(blocked)
    h = _ffi.Handle(0); h.set_stream(None)
    for (m, r0, c0, n) in [(1024, 0, 0, 64), (1024, 64, 64, 192), (4096, 64, 64, 256), (4096, 128, 128, 256), (5000, 448, 448, 512), (9000, 64, 0, 64)]:
        torch.manual_seed(m + r0)
        A0 = torch.rand((m, n), dtype=torch.float64, device="cuda")
        ip = torch.zeros(m, dtype=torch.int64, device="cuda")
        info = ctypes.c_int64(0)
        A = A0.clone()
        try:
            h.call("rflu_panel_rm_f64_dev", m, r0, c0, 64, P(A), n, P(ip), 1, ctypes.byref(info))
            ok = "ok"
        except Exception as ex:
            ok = "ERR " + str(ex)[-60:]
        key = (m, r0, c0, n)
        if blocked == 0: res[key] = (A.clone(), ip.clone())
        else:
            rA, rip = res[key]
            ok += f" same A {bool(torch.equal(rA, A))} ipiv {bool(torch.equal(rip[r0:r0+64], ip[r0:r0+64]))}"
        print(f"blocked={blocked} m={m} r0={r0} c0={c0} n={n}: {ok} info={info.value}", flush=True)
    h.close()
