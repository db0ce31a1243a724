import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import oracle as O
from helpers import rand_matrix
from recursivefactorization.jl_amd.multigpu import MultiGPU
n, block, run = 2048, 128, 2
A = rand_matrix(n, n, seed=100 + n)
Fo, ipo, _ = O.lu(A)
for k, env in ((8, {}), (8, {'RFLU_MGPU_SYNC': '1'}), (8, {'RFLU_MGPU_SYNC': '2'}), (8, {'RFLU_MGPU_SYNC': '3'}), (8, {'RFLU_MGPU_TALL_ROWS': '0'}), (8, {})):
    for kk in ('RFLU_MGPU_SYNC', 'RFLU_MGPU_TALL_ROWS'): os.environ.pop(kk, None)
    os.environ.update(env)
    mg = MultiGPU([0] * k)
    slabs, lds, layout = mg.alloc(n, torch.float64, block, run)
    mg.scatter(A, slabs, layout)
    ipiv, info = mg.getrf(n, slabs, lds, block, run, pivot=True)
    LU = mg.gather(slabs, layout, n)
    bad = np.nonzero(ipiv != ipo)[0]
    d = np.abs(LU - Fo)
    percol = d.max(axis=0)
    print(env, f"k={k}: info={info} first ipiv mismatch {bad[:5]}  max diff {d.max():.3e}; per block max:", " ".join(f"{percol[j:j+block].max():.1e}" for j in range(0, n, block)))
    rows = d.max(axis=1)
    print("   per row-block max:", " ".join(f"{rows[j:j+block].max():.1e}" for j in range(0, n, block)))
    mg.close()
