"""The host-pointer boundary (what the Julia ccall stub binds): lu!(A, ipiv) on a caller-owned column-major host array: the whole
call (in over PCIe, factor, out over PCIe).  One buffer refilled in place, as LinearSolve reuses its A."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import recursivefactorization.jl_amd as rf
for n in (4096, 8192, 16384):
    rng = np.random.default_rng(1)
    A0 = np.asfortranarray(rng.random((n, n)))
    A = np.empty_like(A0, order="F")
    ipiv = np.empty(n, np.int64)
    ts = []
    for _ in range(4):
        A[...] = A0
        t0 = time.perf_counter(); rf.lu_(A, ipiv, rf.Val(True), rf.Val(False), check=False); ts.append(time.perf_counter() - t0)
    print(f"n={n}: host-pointer lu! {min(ts)*1e3:8.1f} ms (matrix {A0.nbytes/2**30:.2f} GiB each way)", flush=True)
