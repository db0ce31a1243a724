"""The trailing updates of the N=16384 factorization as a bare sequence of masked GEMMs (no panels, no solves): per-step time
vs the same shapes inside the real schedule (scripts/trace_timeline.sh).  GEMM_SEQ_REPEAT=1: each shape twice in a row."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RFLU_GEMM_MASKED", "32")
import torch
from recursivefactorization.jl_amd import _ffi
hg = _ffi.Handle(0); hg.set_stream(None)
n, K = 16384, 512
R = torch.rand((n, n), dtype=torch.float64, device="cuda") - 0.5
base = R.data_ptr()
if os.environ.get("GEMM_SEQ_FACTORED"):   # operands with the values a factorization leaves behind (L in [-1,1], U grown)
    ip = torch.zeros(n, dtype=torch.int64, device="cuda"); info = ctypes.c_int64(0)
    hf = _ffi.Handle(0); hf.set_stream(None)
    hf.call("rflu_getrf_rm_f64_dev", n, n, ctypes.c_void_p(base), n, ctypes.c_void_p(ip.data_ptr()), 1, 0, ctypes.byref(info))
    print("factored: info", info.value, " max|R|", float(R.abs().max()), flush=True)
if os.environ.get("GEMM_SEQ_SCALE"):
    R *= float(os.environ["GEMM_SEQ_SCALE"])
def gemm(b):
    je = (b + 1) * K
    M, N = n - je, n - je - 2 * K
    t0 = time.perf_counter()
    hg.call("rflu_gemm_rm_f64_dev", M, N, K, ctypes.c_void_p(base + (je * n + b * K) * 8), n, ctypes.c_void_p(base + (b * K * n + je + 2 * K) * 8), n,
            ctypes.c_void_p(base + (je * n + je + 2 * K) * 8), n)
    return time.perf_counter() - t0, 2.0 * M * N * K, (M // 128) * ((N + 127) // 128)
for rep in range(3):
    tot = 0
    line = []
    for b in range(0, 12):
        if os.environ.get("GEMM_SEQ_GAP_US"):   # an idle gap before every GEMM (power-management step response)
            t1 = time.perf_counter() + float(os.environ["GEMM_SEQ_GAP_US"]) * 1e-6
            if os.environ.get("GEMM_SEQ_HEAT"): hg.call("rflu_debug_heat", float(os.environ["GEMM_SEQ_GAP_US"]) * float(os.environ["GEMM_SEQ_HEAT"]))
            while time.perf_counter() < t1: pass
        t, fl, tiles = gemm(b)
        if os.environ.get("GEMM_SEQ_REPEAT"): t, fl, tiles = gemm(b)
        tot += t
        line.append(f"{t*1e3:.2f}({fl/t/1e12:.1f})")
    print(f"rep {rep}: ms(TF) per step:", " ".join(line), f" sum {tot*1e3:.1f} ms", flush=True)
