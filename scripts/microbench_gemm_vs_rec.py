"""Masked update GEMM (15872 x 14848 x 512, inside a 16384^2 matrix) while a host thread keeps factoring a 15872 x 512 block
column with the panel recursion on another handle (the critical-path stream's kind of work, unconfined)."""
import ctypes, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RFLU_GEMM_MASKED"] = "32"
import torch
from recursivefactorization.jl_amd import _ffi
P = lambda t: ctypes.c_void_p(t.data_ptr())
sg, sp = torch.cuda.Stream(), torch.cuda.Stream()   # non-blocking streams: the null stream would serialize the two handles
hg = _ffi.Handle(0); hg.set_stream(sg.cuda_stream)
hp = _ffi.Handle(0); hp.set_stream(sp.cuda_stream)
n, K = 16384, 512
R = torch.rand((n, n), dtype=torch.float64, device="cuda") - 0.5
base = R.data_ptr()
M, N = n - K, n - 3 * K
gemm = lambda: hg.call("rflu_gemm_rm_f64_dev", M, N, K, ctypes.c_void_p(base + K * n * 8), n, ctypes.c_void_p(base + 3 * K * 8), n,
                       ctypes.c_void_p(base + (K * n + 3 * K) * 8), n)
mode = sys.argv[1] if len(sys.argv) > 1 else "rec"
m = 15872
W = 512 if mode == "rec" else 64
A0 = torch.rand((m, W), dtype=torch.float64, device="cuda")
A = A0.clone()
inmat = len(sys.argv) > 2   # the block column lives inside the big matrix (row stride 128 KiB) like in the factorization
Av = R[K:, K:K + W] if inmat else A
ip = torch.zeros(m, dtype=torch.int64, device="cuda")
info = ctypes.c_int64(0)
stop = False
count = [0]
def work():
    while not stop:
        with torch.cuda.stream(sp): Av.copy_(A0)
        if mode == "rec":
            hp.call("rflu_getrf_rm_f64_dev", m, W, P(Av), Av.stride(0), P(ip), 1, -1, ctypes.byref(info))
        else:
            hp.call("rflu_panel_rm_f64_dev", m, 0, 0, 64, P(A), 64, P(ip), 1, ctypes.byref(info))
        count[0] += 1
        if count[0] <= 3 or stop: print(f"   [{mode} call {count[0]} done, info={info.value}]", flush=True)
def bench(tag, reps=25):
    for _ in range(15): gemm()
    c0 = count[0]; t0 = time.perf_counter()
    for _ in range(reps): gemm()
    t = (time.perf_counter() - t0) / reps
    print(f"{tag}: masked GEMM {t*1e3:7.3f} ms  {2*M*N*K/t/1e12:6.2f} TFLOP/s   ({count[0]-c0} {mode} calls meanwhile, {t*reps*1e3/max(count[0]-c0,1):.2f} ms each)", flush=True)
bench("alone  ")
th = threading.Thread(target=work); th.start()
bench("next to")
stop = True; th.join()
bench("alone  ")
