"""sha1 of C after C -= A*B for a few shapes (interior / edge tiles, K a multiple of 16 or not, Float64 / Float32): two builds
of gemm.hip that claim identical arithmetic print identical lines.  usage: RFLU_LIB=... python scripts/gemm_checksum.py"""
import ctypes, hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recursivefactorization.jl_amd import _ffi
h = _ffi.Handle(0); h.set_stream(None)
P = lambda t: ctypes.c_void_p(t.data_ptr())
for sfx, dt in (("f64", torch.float64), ("f32", torch.float32)):
    for (M, N, K) in ((2048, 2048, 512), (1000, 900, 77), (640, 512, 48), (384, 256, 16), (4096, 1024, 256), (130, 4000, 1024)):
        g = torch.Generator(device="cuda"); g.manual_seed(M * 7 + N * 3 + K)
        A = torch.rand((M, K), dtype=dt, device="cuda", generator=g) - 0.5
        B = torch.rand((K, N), dtype=dt, device="cuda", generator=g) - 0.5
        C = torch.rand((M, N), dtype=dt, device="cuda", generator=g)
        ref = C.double() - A.double() @ B.double()
        h.call(f"rflu_gemm_rm_{sfx}_dev", M, N, K, P(A), K, P(B), N, P(C), N)
        h.synchronize()
        err = float((C.double() - ref).abs().max())
        print(f"{sfx} {M}x{N}x{K}: sha1 {hashlib.sha1(C.cpu().numpy().tobytes()).hexdigest()[:16]}  max |C - fp64 reference| {err:.3e}")
