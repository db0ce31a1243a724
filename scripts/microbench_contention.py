"""How much slower do the critical-path kernels run next to the bulk update?  A loop of big GEMMs runs on the CU-masked update
stream while small kernels are timed on (a) an unmasked stream, (b) a stream masked to exactly the 32 reserved CUs."""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recursivefactorization.jl_amd import _ffi
lib = _ffi.load()
h = _ffi.Handle(0)
dev = torch.device("cuda", 0)
U = torch.cuda.ExternalStream(h.update_stream(), device=dev)
lib.rflu_debug_masked_stream.restype = ctypes.c_int
lib.rflu_debug_masked_stream.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
mask = (ctypes.c_uint * 8)(0xffffffff, 0, 0, 0, 0, 0, 0, 0)   # the 32 CUs the update stream leaves alone
sp = ctypes.c_void_p()
assert lib.rflu_debug_masked_stream(h.ptr, mask, ctypes.byref(sp)) == 0
R32 = torch.cuda.ExternalStream(sp.value, device=dev)
PL = torch.cuda.Stream(device=dev)
ld = 16384
R = torch.rand((16384 + 512, ld), dtype=torch.float64, device=dev)
Abig = torch.rand((8192, 512), dtype=torch.float64, device=dev); Bbig = torch.rand((512, 8192), dtype=torch.float64, device=dev) * 1e-3
Cbig = torch.rand((8192, 8192), dtype=torch.float64, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
def small(M, N, K):
    a = R.data_ptr() + (512 * ld) * 8; b = R.data_ptr() + 512 * 8; c = R.data_ptr() + (512 * ld + 512) * 8
    h.call("rflu_gemm_rm_f64_dev", M, N, K, ctypes.c_void_p(a), ld, ctypes.c_void_p(b), ld, ctypes.c_void_p(c), ld)
def run(stream, load, M, N, K, reps=100):
    torch.cuda.synchronize()
    if load:
        h.set_stream(U.cuda_stream)
        for _ in range(12): h.call("rflu_gemm_rm_f64_dev", 8192, 8192, 512, P(Abig), 512, P(Bbig), 8192, P(Cbig), 8192)
    h.set_stream(stream.cuda_stream)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        for _ in range(5): small(M, N, K)
        e0.record(stream)
        for _ in range(reps): small(M, N, K)
        e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (M, N, K) in ((16384, 64, 64), (16384, 128, 128), (8192, 64, 64)):
    print(f"small gemm {M}x{N}x{K}: alone/unmasked {run(PL, False, M, N, K):6.1f} us | alone/32-CU stream {run(R32, False, M, N, K):6.1f} us | "
          f"loaded/unmasked {run(PL, True, M, N, K):6.1f} us | loaded/32-CU stream {run(R32, True, M, N, K):6.1f} us", flush=True)
