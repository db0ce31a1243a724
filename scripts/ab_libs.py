"""A/B of two builds of librflu inside ONE process on one box (alternating calls): python scripts/ab_libs.py libA.so libB.so n reps [f64|f32] [blocksize]
(the environment, e.g. RFLU_ENGINE=1, applies to both).  Box-to-box and process-to-process differences (1-2 %) are larger than most of what a
change to the engine's scheduler moves; the same matrix, the same process and alternating calls are not."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from gpu_util import fill_uniform_cm

paths = [os.path.abspath(sys.argv[1]), os.path.abspath(sys.argv[2])]
n = int(sys.argv[3]); reps = int(sys.argv[4])
sfx = sys.argv[5] if len(sys.argv) > 5 else "f64"
bs = int(sys.argv[6]) if len(sys.argv) > 6 else 0
c_p, c_i64, c_int = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
libs, hs = [], []
for p in paths:
    L = ctypes.CDLL(p, mode=os.RTLD_LOCAL)
    L.rflu_create.argtypes = [ctypes.POINTER(c_p), c_int]
    f = getattr(L, f"rflu_getrf_{sfx}_dev")
    f.argtypes = [c_p, c_i64, c_i64, c_p, c_i64, c_p, c_int, c_i64, c_p]
    L.rflu_set_stream.argtypes = [c_p, c_p]
    L.rflu_last_path.argtypes = [c_p]
    h = c_p()
    assert L.rflu_create(ctypes.byref(h), 0) == 0
    L.rflu_set_stream(h, None)
    libs.append(L); hs.append(h)
dt = np.float64 if sfx == "f64" else np.float32
A0 = fill_uniform_cm(n, dt, 12, 0.0)
ip = torch.zeros(n, dtype=torch.int64, device="cuda")
times = [[], []]
outs = [None, None]
for r in range(reps + 1):
    for k in (0, 1):
        A = A0.clone(); info = c_i64(0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc = getattr(libs[k], f"rflu_getrf_{sfx}_dev")(hs[k], n, n, c_p(A.data_ptr()), n, c_p(ip.data_ptr()), 1, bs, ctypes.byref(info))
        torch.cuda.synchronize()
        assert rc == 0 and info.value == 0, (rc, info.value)
        if r > 0: times[k].append((time.perf_counter() - t0) * 1e3)
        else: outs[k] = (A, ip.clone(), libs[k].rflu_last_path(hs[k]))
same = torch.equal(outs[0][1], outs[1][1])
d = float((outs[0][0] - outs[1][0]).abs().max())
for k in (0, 1):
    t = sorted(times[k])
    print(f"{os.path.basename(paths[k])}: n={n} {sfx} bs={bs} path {outs[k][2]}: best {t[0]:.2f} ms, median {t[len(t)//2]:.2f}")
print(f"pivots equal: {same}, max |factor difference| {d:.3e}")
