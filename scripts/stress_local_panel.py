"""Stress for the XCD-local short-panel path: factorizations next to somebody else's kernels (a torch matmul loop on another stream,
all CUs).  The leaf's participants are the workgroups of an 8*G launch that land on one XCD; if the hardware ever dealt the blocks
differently, or a participant could not become resident, the bounded spins would end the call with RFLU_ERR_TIMEOUT -- which is what
this looks for.  usage: stress_local_panel.py [n [repetitions]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import recursivefactorization.jl_amd as rf

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
A = torch.rand((n, n), dtype=torch.float64, device="cuda").T.contiguous().T
X = torch.rand((8192, 8192), dtype=torch.float32, device="cuda")
side = torch.cuda.Stream()
ref = rf.lu_(A.clone(), None, True, check=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
bad = 0
for i in range(reps):
    with torch.cuda.stream(side):
        for _ in range(3):
            Y = X @ X          # ~1.1 TFLOP each: the GPU is never idle while the factorization runs
    F = rf.lu_(A.clone(), None, True, check=False)
    if F.info != 0 or not torch.equal(F.ipiv, ref.ipiv) or not torch.equal(F.factors, ref.factors):
        bad += 1
torch.cuda.synchronize()
print(f"n={n}: {reps} factorizations next to a matmul loop, {1e3 * (time.perf_counter() - t0) / reps:.1f} ms each, "
      f"{bad} differ from the undisturbed result, last path {rf.last_path()}")
