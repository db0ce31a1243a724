"""sha1 of the factors and pivots of lu!(A) for a few sizes / element types: two builds or two settings of a tuning variable that
claim identical arithmetic print identical lines.  usage: [RFLU_...=..] python scripts/lu_checksum.py [n ...]"""
import ctypes, hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recursivefactorization.jl_amd import _ffi
h = _ffi.Handle(0); h.set_stream(None)
sizes = [int(x) for x in sys.argv[1:]] or [4096, 8192, 5000]
for sfx, dt in (("f64", torch.float64), ("f32", torch.float32)):
    for n in sizes:
        g = torch.Generator(device="cuda"); g.manual_seed(n)
        A = torch.rand((n, n), dtype=dt, device="cuda", generator=g)
        ip = torch.zeros(n, dtype=torch.int64, device="cuda"); info = ctypes.c_int64(0)
        h.call(f"rflu_getrf_rm_{sfx}_dev", n, n, ctypes.c_void_p(A.data_ptr()), n, ctypes.c_void_p(ip.data_ptr()), 1, 0, ctypes.byref(info))
        print(f"{sfx} n={n}: info {info.value} factors {hashlib.sha1(A.cpu().numpy().tobytes()).hexdigest()[:16]} ipiv {hashlib.sha1(ip.cpu().numpy().tobytes()).hexdigest()[:16]}")
