"""The engine's right-looking block-row solve (csrc/engine.hip: eng_prep_unit) against the left-looking one it replaces: every accumulator
receives the same products in the same order, so the factors must be equal TO THE BIT.  Needs the experiments build (RFLU_ENGINE_SOLVE_RL is
read by it only):  RFLU_LIB=$PWD/recursivefactorization.jl_amd/librflu_exp.so python scripts/engine_solve_check.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from recursivefactorization.jl_amd import _ffi
from gpu_util import fill_uniform_cm

h = _ffi.Handle(0); h.set_stream(None)


def factor(m, n, sfx, bs, rl):
    for k in list(os.environ):
        if k.startswith("RFLU_") and k != "RFLU_LIB": del os.environ[k]
    os.environ["RFLU_ENGINE"] = "1"; os.environ["RFLU_ENGINE_SOLVE_RL"] = str(rl)
    h.reload_tuning()
    A = fill_uniform_cm(n, np.float64 if sfx == "f64" else np.float32, 12, 0.0, m=m).clone()
    ip = torch.zeros(min(m, n), dtype=torch.int64, device="cuda"); info = ctypes.c_int64(0)
    h.call(f"rflu_getrf_{sfx}_dev", m, n, ctypes.c_void_p(A.data_ptr()), m, ctypes.c_void_p(ip.data_ptr()), 1, bs, ctypes.byref(info))
    torch.cuda.synchronize()
    return A, ip, info.value, h.last_path()


ok = True
for (m, n, sfx, bs) in [(8192, 8192, "f64", 512), (10000, 6144, "f64", 512), (6144, 6144, "f64", 256), (16384, 16384, "f64", 0), (8192, 8192, "f32", 512), (6000, 6000, "f64", 384)]:
    A1, p1, i1, path = factor(m, n, sfx, bs, 1)
    A0, p0, i0, _ = factor(m, n, sfx, bs, 0)
    same = bool(torch.equal(A1, A0)) and bool(torch.equal(p1, p0)) and i0 == i1 == 0
    ok &= same and path == 4
    print(f"{'OK ' if same else 'BAD'} m={m} n={n} {sfx} W={bs or 'default'} path {path}: factors bit-identical {bool(torch.equal(A1, A0))}, pivots equal {bool(torch.equal(p1, p0))}", flush=True)
print("ALL OK" if ok else "FAILURES")
