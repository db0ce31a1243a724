"""The host-pointer entry (rflu_getrf_f64 on a pageable array: through the engine by default) over and over: pivots equal to the first
call's, factors to rounding, no timeout.  usage: python scripts/host_entry_stress.py [n] [calls]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from recursivefactorization.jl_amd import _ffi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 50
rng = np.random.default_rng(12)
A0 = np.asfortranarray(rng.random((n, n)))
h = _ffi.Handle(0); h.set_stream(None)
ref = None
ts = []
reuse = os.environ.get("REUSE", "0") != "0"   # REUSE=1: one caller buffer refilled in place (what bench.py does) instead of a fresh array per call
A = A0.copy(order="F")
for i in range(calls):
    if reuse: np.copyto(A, A0)
    else: A = A0.copy(order="F")
    ip = np.zeros(n, dtype=np.int64); info = ctypes.c_int64(0)
    t0 = time.perf_counter()
    try:
        h.call("rflu_getrf_f64", n, n, ctypes.c_void_p(A.ctypes.data), n, ctypes.c_void_p(ip.ctypes.data), 1, 0, ctypes.byref(info))
    except Exception as e:
        print(f"call {i}: FAILED {e}", flush=True); sys.exit(1)
    ts.append(time.perf_counter() - t0)
    if os.environ.get("VERBOSE"): print(f"call {i}: {1e3*ts[-1]:.1f} ms", flush=True)
    if ref is None:
        ref = (A.copy(), ip.copy())
        scale = float(np.abs(ref[0][::37]).max())
    elif not (np.array_equal(ip, ref[1]) and np.abs(A[::37] - ref[0][::37]).max() <= 1e-10 * scale):   # (every 37th row: the full difference of two 0.5-2 GiB arrays is most of a call's time)
        print(f"call {i}: differs from the first call", flush=True); sys.exit(2)
print(f"{calls} host-entry calls of n={n} ok: median {1e3*sorted(ts)[len(ts)//2]:.1f} ms, min {1e3*min(ts):.1f}, max {1e3*max(ts[1:]):.1f} (path {h.last_path()})", flush=True)
