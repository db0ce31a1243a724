"""What the update engine does by itself: RFLU_ENGINE_NOPANEL=1 publishes every panel at once (no panels are factored: wrong
factors, right amount of update work), so the time of the call is the engine's time for all trailing updates of an n x n matrix.
usage: python scripts/engine_alone.py [n ...]   (extra RFLU_* variables are passed through)"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recursivefactorization.jl_amd import _ffi
os.environ["RFLU_ENGINE"] = "1"; os.environ["RFLU_ENGINE_NOPANEL"] = "1"; os.environ["RFLU_LEAFWISE"] = "0"
h = _ffi.Handle(0); h.set_stream(None)
sizes = [int(x) for x in sys.argv[1:]] or [16384]
for n in sizes:
    W = 512
    A = torch.zeros((n, n), dtype=torch.float64, device="cuda")    # zeros: nothing overflows, the MFMA work is the same
    ip = torch.zeros(n, dtype=torch.int64, device="cuda"); info = ctypes.c_int64(0)
    best = 1e9
    for r in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        try:
            h.call("rflu_getrf_rm_f64_dev", n, n, ctypes.c_void_p(A.data_ptr()), n, ctypes.c_void_p(ip.data_ptr()), 1, W, ctypes.byref(info))
        except Exception as e:
            print("call failed:", e)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    nb = n // W
    if int(os.environ.get("RFLU_ENGINE_X2", "0")) > 0:
        nb = min(nb, int(os.environ["RFLU_ENGINE_X2"]))
    flops = sum(2.0 * (n - (b + 1) * W) ** 2 * W for b in range(nb))
    print(f"n={n}: engine alone {best * 1e3:.2f} ms for {flops / 1e12:.3f} TFLOP of Schur updates = {flops / best / 1e12:.1f} TFLOP/s "
          f"({os.environ.get('RFLU_ENGINE_POLICY', '0')=}, wgs={os.environ.get('RFLU_ENGINE_WGS', 'default')})", flush=True)
