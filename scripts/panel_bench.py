"""Per-column time of one leaf (m x 64, Float64) for the panel kernel variants, alone on the GPU, and a bit-for-bit
comparison of their results.  Modes: 0 = the two-trip reference kernel of panel.hip (RFLU_PANEL_LOCAL=0, RFLU_PANEL_SINGLE=0),
1 = XCD-local cooperative leaf (panel_local.hip, plain-store records on one XCD), 2 = the shipped routing (one-workgroup LDS
leaf up to 512 rows, cooperative leaf with sc1 records on any placement above), 3 = mode 2 without the one-workgroup leaf, 4 = the sub-panel kernel with one chain wave per workgroup (panel_blocked.hip; modes 0-3
switch it off).
usage: python scripts/panel_bench.py [m ...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recursivefactorization.jl_amd import _ffi

sizes = [int(x) for x in sys.argv[1:]] or [1024, 2048, 4096, 8192, 15872, 16384]
P = lambda t: ctypes.c_void_p(t.data_ptr())
modes = [int(x) for x in os.environ.get("PANEL_MODES", "0,1,2").split(",")]
dtype = torch.float32 if os.environ.get("PANEL_F32") else torch.float64
sfx = "f32" if dtype == torch.float32 else "f64"
ref = {}
for mode in modes:
    os.environ["RFLU_PANEL_LOCAL"] = str({0: 0, 1: 1, 2: 2, 3: 2, 4: 2, 5: 1}[mode])
    os.environ["RFLU_PANEL_SINGLE"] = "0" if mode in (0, 3) else "1"
    os.environ["RFLU_PANEL_BLOCKED"] = "1" if mode in (4, 5) else "0"
    h = _ffi.Handle(0)
    h.set_stream(None)
    for m in sizes:
        torch.manual_seed(m)
        A0 = torch.rand((m, 64), dtype=dtype, device="cuda")
        ip = torch.zeros(m, dtype=torch.int64, device="cuda")
        info = ctypes.c_int64(0)
        reps = 12
        for it in range(reps + 3):
            if it == 3:
                h.profile_enable(True)
            A = A0.clone()
            h.call(f"rflu_panel_rm_{sfx}_dev", m, 0, 0, 64, P(A), 64, P(ip), 1, ctypes.byref(info))
        pr = h.profile()["panel"]
        h.profile_enable(False)
        us = pr["ms"] * 1e3 / pr["launches"]
        key = (m,)
        same = ""
        if key in ref:
            rA, rip = ref[key]
            same = f"  bit-identical to mode {modes[0]}: A {bool(torch.equal(rA, A))} ipiv {bool(torch.equal(rip, ip[:64]))}"
        else:
            ref[key] = (A.clone(), ip[:64].clone())
        print(f"mode {mode} m={m:6d} G={(m + 511) // 512:3d}: {us:8.1f} us per leaf = {us / 64 * 1000:7.0f} ns per column  info={info.value}{same}", flush=True)
    h.close()
