"""Randomised parity sweep on the GPU: random shapes / dtypes / pivot modes / block widths against the CPU oracle
(ipiv and info bit-exact, residual within the reference's bound).  Not part of the test suite; run ad hoc via gpurun."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle as O
import recursivefactorization.jl_amd as rf

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 60
maxdim = int(sys.argv[3]) if len(sys.argv) > 3 else 2600
bad = 0
for it in range(ncase):
    m = int(rng.integers(1, maxdim)); n = int(rng.integers(1, maxdim))
    if rng.random() < 0.3: n = m
    dt = np.float64 if rng.random() < 0.7 else np.float32
    pivot = rng.random() < 0.8
    bs = int(rng.choice([-1, 0, 64, 128, 192, 256, 512]))
    A = O.np_uniform(m, n, int(rng.integers(1, 1 << 30)), dt)
    kind = rng.random()
    if not pivot:
        A = A + 10 * np.eye(m, n, dtype=dt)
    elif kind < 0.15:
        A = np.floor(A * 4) - 1.5          # many exact ties
    elif kind < 0.25 and n > 3:
        A[:, n // 2] = 0                    # a zero column: info > 0, factorization continues
    A = np.asfortranarray(A.astype(dt))
    F = rf.lu(A, True if pivot else rf.NoPivot(), check=False, blocksize=bs)
    Fo, ipo, info = O.lu(A, pivot=pivot)
    ok = abs(F.info) == info
    same_piv = (not pivot) or np.array_equal(np.asarray(F.ipiv), ipo)
    # Float64: pivots must be bit-exact.  Float32: beyond n ~ 1000 two candidates can be closer than the rounding error
    # accumulated by DIFFERENT (both valid) summation orders; the sequence may then fork -- the properties below must hold
    if dt == np.float64: ok = ok and same_piv
    elif not same_piv:
        L = np.tril(np.asarray(F.factors), -1)
        ok = ok and float(np.abs(L).max()) <= 1.0 and min(m, n) > 500
    if ok and info == 0:
        E = 20 * m * np.finfo(dt).eps
        ip = np.asarray(F.ipiv) if pivot else np.arange(1, min(m, n) + 1)
        mx, _ = O.residual(A, np.asarray(F.factors), ip)
        scale = max(1.0, float(np.max(np.abs(Fo))))
        ok = mx < (E if pivot else 10 * np.sqrt(E) * scale)
    print(f"{it:3d} m={m:5d} n={n:5d} {dt.__name__} pivot={pivot} bs={bs:4d} info={info} path={rf.last_path()}: {'ok' if ok else 'MISMATCH'}{'' if same_piv else ' (f32 pivot fork)'}", flush=True)
    bad += (not ok)
print("FUZZ_BAD", bad)
