"""The caller's own streams shift which hardware queue the library's update / side streams get (the runtime hands out its few
hardware queues round-robin in creation order).  k extra streams created and used between the handle's own stream and its update / side streams (RFLU_DUMMY_QUEUES):
time per N=16384 lu! with the queue check (default) and without (RFLU_QUEUE_CHECK=0).
usage: python scripts/queue_collision.py  (spawns itself per configuration)"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    k = int(sys.argv[1])
    import ctypes
    import torch
    from recursivefactorization.jl_amd import _ffi
    torch.zeros(1, device="cuda")
    os.environ["RFLU_DUMMY_QUEUES"] = str(k)   # the library's test hook: k streams created and used right after the handle's own
    n = 16384
    h = _ffi.Handle(0)
    h.set_stream(torch.cuda.current_stream().cuda_stream)
    A = torch.empty((n, n), dtype=torch.float64, device="cuda")
    ip = torch.empty(n, dtype=torch.int64, device="cuda")
    info = ctypes.c_int64(0)
    ts = []
    for it in range(5):
        h.call("rflu_fill_uniform_f64_dev", ctypes.c_void_p(A.data_ptr()), n, n, n, 0, 12, n, 0, 0, 0.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        h.call("rflu_getrf_f64_dev", n, n, ctypes.c_void_p(A.data_ptr()), n, ctypes.c_void_p(ip.data_ptr()), 1, 0, ctypes.byref(info))
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print(f"{min(ts[1:]) * 1e3:.1f}")
    sys.exit(0)
for k in range(0, 6):
    row = []
    for chk in ("1", "0"):
        out = subprocess.run([sys.executable, __file__, str(k)], env=dict(os.environ, RFLU_QUEUE_CHECK=chk), capture_output=True, text=True)
        if chk == "1" and os.environ.get("RFLU_QUEUE_TRACE"):
            print("\n".join(l for l in out.stderr.splitlines() if "queue check" in l))
        row.append(out.stdout.strip().splitlines()[-1] if out.returncode == 0 and out.stdout.strip() else f"rc={out.returncode} {out.stderr.strip().splitlines()[-1] if out.stderr.strip() else ''}")
    print(f"{k} extra caller streams: lu! N=16384  with queue check {row[0]:>8} ms   without {row[1]:>8} ms", flush=True)
