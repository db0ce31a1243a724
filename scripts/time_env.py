"""A/B timing of environment settings inside ONE process on one box: the device entry rflu_getrf_*_dev at size n, alternating between the
settings (rflu_reload_tuning between calls), best and median of `reps` calls each.
usage: python scripts/time_env.py n reps [f64|f32] [pivot 0|1] "A=1,B=2" "" "C=3" ...   ("" = the defaults)"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from recursivefactorization.jl_amd import _ffi
from gpu_util import fill_uniform_cm, matvec_residual

n = int(sys.argv[1]); reps = int(sys.argv[2])
args = sys.argv[3:]
sfx = "f64"
pivot = 1
if args and args[0] in ("f64", "f32"): sfx = args.pop(0)
if args and args[0] in ("0", "1"): pivot = int(args.pop(0))
cfgs = args or [""]
h = _ffi.Handle(0); h.set_stream(None)
dt = np.float64 if sfx == "f64" else np.float32
A0 = fill_uniform_cm(n, dt, 12, 0.0 if pivot else 10.0)
ip = torch.zeros(n, dtype=torch.int64, device="cuda")
times = {c: [] for c in cfgs}
res = {}
for r in range(reps + 1):
    for c in cfgs:
        for k in list(os.environ):
            if k.startswith("RFLU_"): del os.environ[k]
        bs = 0
        for kv in filter(None, c.split(",")):
            k, v = kv.split("=")
            if k == "BS": bs = int(v)      # (pseudo-variable: the blocksize argument of the call)
            else: os.environ[k] = v
        h.reload_tuning()
        A = A0.clone(); info = ctypes.c_int64(0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        h.call(f"rflu_getrf_{sfx}_dev", n, n, ctypes.c_void_p(A.data_ptr()), n, ctypes.c_void_p(ip.data_ptr()) if pivot else None, pivot, bs, ctypes.byref(info))
        torch.cuda.synchronize()
        if r > 0: times[c].append((time.perf_counter() - t0) * 1e3)
        elif c not in res:
            res[c] = (info.value, matvec_residual(A0, A, ip if pivot else np.arange(1, n + 1)), h.last_path())
for c in cfgs:
    t = sorted(times[c])
    print(f"n={n} {sfx} piv={pivot} [{c or 'defaults'}]: best {t[0]:.2f} ms, median {t[len(t)//2]:.2f} (info {res[c][0]}, residual {res[c][1]:.1e}, path {res[c][2]})", flush=True)
