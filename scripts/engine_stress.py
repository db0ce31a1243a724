"""Liveness of the persistent update engine: the same factorization over and over (the engine to the end), stopping at the first call that
does not come back with info 0.  RFLU_ENGINE_DUMP=1 prints the engine's state when somebody times out.
usage: python scripts/engine_stress.py [n] [calls]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from recursivefactorization.jl_amd import _ffi
from gpu_util import fill_uniform_cm

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 100
os.environ.setdefault("RFLU_ENGINE", "1")
os.environ.setdefault("RFLU_ENGINE_ROWS", "0")
os.environ.setdefault("RFLU_ENGINE_DUMP", "1")
h = _ffi.Handle(0); h.set_stream(None)
A0 = fill_uniform_cm(n, np.float64, 12, 0.0)
ip = torch.zeros(n, dtype=torch.int64, device="cuda")
ref = None
t0 = time.time()
for i in range(calls):
    A = A0.clone()
    info = ctypes.c_int64(0)
    try:
        h.call("rflu_getrf_f64_dev", n, n, ctypes.c_void_p(A.data_ptr()), n, ctypes.c_void_p(ip.data_ptr()), 1, 0, ctypes.byref(info))
        torch.cuda.synchronize()
    except Exception as e:
        print(f"call {i}: FAILED {e}", flush=True)
        sys.exit(1)
    if ref is None:
        ref = (A.clone(), ip.clone())
    elif not (torch.equal(ip, ref[1]) and float((A - ref[0]).abs().max()) <= 1e-10 * float(ref[0].abs().max())):
        print(f"call {i}: result differs from the first call's", flush=True)
        sys.exit(2)
print(f"{calls} calls of n={n} ok in {time.time() - t0:.1f} s (path {h.last_path()})", flush=True)
