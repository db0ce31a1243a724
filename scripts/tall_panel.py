"""One block column of a tall matrix by itself: what the OWNER of a block column does in the 1-D block-column multi-GPU driver before it
broadcasts (driver.cpp: mgpu_getrf factors its panel with the pure recursion, Fact::rec) -- m x 512 through rflu_getrf_rm_f64_dev with the
pure recursion (blocksize -1) and with 128- / 256-wide block columns inside the panel.  The measured time is what DESIGN.md section 6's
8-GPU model uses for the chain of panels.   usage: python scripts/tall_panel.py [w]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from recursivefactorization.jl_amd import _ffi

w = int(sys.argv[1]) if len(sys.argv) > 1 else 512
h = _ffi.Handle(0); h.set_stream(None)
for m in (16384, 32768, 49152, 65536):
    A0 = torch.rand((m, w), dtype=torch.float64, device="cuda")      # row-major m x w
    ip = torch.zeros(w, dtype=torch.int64, device="cuda")
    ref = None
    for bs in (-1, 128, 256):
        best = 1e30
        for r in range(4):
            A = A0.clone(); info = ctypes.c_int64(0)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            h.call("rflu_getrf_rm_f64_dev", m, w, ctypes.c_void_p(A.data_ptr()), w, ctypes.c_void_p(ip.data_ptr()), 1, bs, ctypes.byref(info))
            torch.cuda.synchronize()
            if r: best = min(best, (time.perf_counter() - t0) * 1e3)
        same = "" if ref is None else f", pivots equal to the recursion's: {bool(torch.equal(ref, ip))}"
        if ref is None: ref = ip.clone()
        print(f"m={m} w={w} blocksize={bs}: {best:.3f} ms = {best * 1e3 / w:.2f} us per column (info {info.value}, path {h.last_path()}{same})", flush=True)
