"""The persistent update engine ALONE on the GPU (RFLU_ENGINE_REPLAY=1): one real factorization leaves the pivots' move lists and the
diagonal inverses behind, then the engine is launched on that image with every leaf counted as done -- no chain of leaves next to it, no
waiting for it: what the resident kernel does all-in with 224 CUs to itself, and the one form in which rocprofv3 --pmc (one kernel at
a time) can collect its counters (scripts/pmc_engine.sh).  The replayed numbers are meaningless (updates applied to factors), the
addresses, shapes and counts are those of the factorization.
usage: python scripts/engine_replay.py [n] [reps]"""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from recursivefactorization.jl_amd import _ffi
from gpu_util import fill_uniform_cm

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
os.environ.pop("RFLU_ENGINE_REPLAY", None)
h = _ffi.Handle(0); h.set_stream(None)
A0 = fill_uniform_cm(n, np.float64, 12, 0.0)
ip = torch.zeros(n, dtype=torch.int64, device="cuda")
info = ctypes.c_int64(0)


def getrf(A):
    h.call("rflu_getrf_f64_dev", n, n, ctypes.c_void_p(A.data_ptr()), n, ctypes.c_void_p(ip.data_ptr()), 1, 0, ctypes.byref(info))
    torch.cuda.synchronize()


A = A0.clone()
getrf(A)                      # the real factorization (under counter collection: the event schedule, same pivots and inverses)
print(f"factorization: info {info.value}, path {h.last_path()}", flush=True)
os.environ["RFLU_ENGINE_REPLAY"] = "1"
os.environ["RFLU_ENGINE"] = "1"
h.reload_tuning()
out = []
for r in range(reps):
    h.profile_enable(2)
    t0 = time.perf_counter()
    try:
        getrf(A)
    except _ffi.RfluError as e:   # (the replayed numbers may trip nothing, but a timeout would show here)
        print("replay failed:", e, flush=True)
        sys.exit(1)
    wall = (time.perf_counter() - t0) * 1e3
    k = h.profile()["gemm"]
    h.profile_enable(False)
    out.append({"wall_ms": round(wall, 3), "engine_ms": round(k["ms"], 3), "launches": k["launches"], "flops": k["work"], "algorithmic_bytes": k["bytes"],
                "tflops": round(k["work"] / (k["ms"] * 1e-3) / 1e12, 2) if k["ms"] > 0 else None, "path": h.last_path()})
    print(json.dumps(out[-1]), flush=True)
