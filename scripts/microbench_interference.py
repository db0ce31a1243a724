"""What does the bulk update lose to the chain?  A train of bulk GEMMs (S x S x 512, the shape of a phase-A update) runs on the
CU-masked update stream (224 CUs) while the caller's stream runs (a) nothing, (b) 64-column leaves of S rows back to back (the
cooperative pivot kernel on the 32 reserved CUs, nothing else), (c) whole 512-column panels (leaves + interchanges + solves +
the recursion's merges: everything the chain does in phase A except the next-block update), (d) K=64 skinny updates only.
usage: python scripts/microbench_interference.py [S]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recursivefactorization.jl_amd import _ffi
S = int(sys.argv[1]) if len(sys.argv) > 1 else 12800
NG = int(os.environ.get("NG", "10"))
h = _ffi.Handle(0)
dev = torch.device("cuda", 0)
U = torch.cuda.ExternalStream(h.update_stream(), device=dev)
PL = torch.cuda.Stream(device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
A = torch.rand((S, 512), dtype=torch.float64, device=dev)
B = torch.rand((512, S), dtype=torch.float64, device=dev) * 1e-3
C = torch.rand((S, S), dtype=torch.float64, device=dev)
ld = 16384
R = torch.rand((S, ld), dtype=torch.float64, device=dev)
src = torch.rand((S, 512), dtype=torch.float64, device=dev)
ip = torch.zeros(S, dtype=torch.int64, device=dev)
info = ctypes.c_int64(0)

def bulk_train():
    h.set_stream(U.cuda_stream)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    h.call("rflu_gemm_rm_f64_dev", S, S, 512, P(A), 512, P(B), S, P(C), S)   # warm
    e0.record(U)
    for _ in range(NG): h.call("rflu_gemm_rm_f64_dev", S, S, 512, P(A), 512, P(B), S, P(C), S)
    e1.record(U)
    return e0, e1

def chain(kind, e1):
    h.set_stream(PL.cuda_stream)
    n = 0
    t0 = time.perf_counter()
    with torch.cuda.stream(PL):
        while not e1.query():
            if kind == "leaf":
                R[:, :64].copy_(src[:, :64])
                h.call("rflu_panel_rm_f64_dev", S, 0, 0, 64, P(R), ld, P(ip), 1, ctypes.byref(info))
            elif kind == "panel":
                R[:, :512].copy_(src)
                h.call("rflu_panel_rm_f64_dev", S, 0, 0, 512, P(R), ld, P(ip), 1, ctypes.byref(info))
            elif kind == "skinny":
                for _ in range(20):
                    h.call("rflu_gemm_rm_f64_dev", S - 64, 64, 64, ctypes.c_void_p(R.data_ptr() + 64 * ld * 8), ld,
                           ctypes.c_void_p(R.data_ptr() + 64 * 8), ld, ctypes.c_void_p(R.data_ptr() + (64 * ld + 64) * 8), ld)
                PL.synchronize()
            elif kind == "merge":   # the recursion's widest merge: S x 256 x 256
                for _ in range(5):
                    h.call("rflu_gemm_rm_f64_dev", S - 256, 256, 256, ctypes.c_void_p(R.data_ptr() + 256 * ld * 8), ld,
                           ctypes.c_void_p(R.data_ptr() + 256 * 8), ld, ctypes.c_void_p(R.data_ptr() + (256 * ld + 256) * 8), ld)
                PL.synchronize()
            else:
                time.sleep(0.0005)
            n += 1
    return n, (time.perf_counter() - t0) * 1e3

for kind in ("none", "leaf", "panel", "skinny", "merge", "none"):
    torch.cuda.synchronize()
    e0, e1 = bulk_train()
    n, ms = chain(kind, e1)
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / NG
    print(f"S={S} bulk GEMM next to {kind:7s}: {t:7.3f} ms per launch = {2.0 * S * S * 512 / t / 1e9:6.2f} TFLOP/s   ({n} chain calls in {ms:.1f} ms)", flush=True)
