"""Generates tests/golden/*.npz -- small known-answer fixtures for the recursive-LU path.

Provenance: the reference (Julia) cannot run here and its tests hold no golden vectors, so the expected
outputs are produced by the repo's CPU oracle (oracle/rflu_oracle.c, a restatement of /root/reference/src/lu.jl) and
are only written after LAPACK getrf (scipy) agrees on ipiv/info for the pivoted cases -- the comparator the
reference's own tests use (test/runtests.jl:11,15).  Inputs are NOT stored: each fixture stores the generator spec
(kind, m, n, seed, dtype, zero_col) and `build_input` rebuilds the matrix with the repo's counter-based generator, so
fixtures stay tiny and reproducible on the GPU box.

Run:  python tests/golden/make_golden.py      (re-generates every fixture in place)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.dirname(HERE)):
    if p not in sys.path:
        sys.path.insert(0, p)

KINDS = ["uniform", "uniform_plus10I", "singular", "wilkinson", "ties", "nan"]


def build_input(g):
    """Rebuild the input matrix of a fixture (or of a spec dict with the same keys)."""
    import oracle as O
    from helpers import wilkinson

    kind = KINDS[int(g["kind"])]
    m, n, seed = int(g["m"]), int(g["n"]), int(g["seed"])
    dtype = np.float64 if int(g["dtype_bits"]) == 64 else np.float32
    if kind == "wilkinson":
        return wilkinson(n, dtype)
    A = O.np_uniform(m, n, seed, dtype)
    if kind == "uniform_plus10I":
        A = np.asfortranarray(A + dtype(10) * np.eye(m, n, dtype=dtype))
    elif kind == "singular":
        A[:, int(g["zero_col"])] = 0
    elif kind == "ties":
        # small-integer matrix: many exact ties in |a_ik| -> pins the lowest-index tie-break
        A = np.asfortranarray(np.floor(A * 4).astype(dtype) - dtype(1.5))
    elif kind == "nan":
        A[int(g["zero_col"]), 0] = np.nan  # NaN in the first column must never be chosen as pivot
    return A


def main():
    import scipy.linalg as sla

    import oracle as O

    specs = []
    for bits in (64, 32):
        for s in (8, 9, 10, 11, 41, 50, 130, 300):
            specs.append(dict(kind=0, m=s, n=s, seed=12 + s, dtype_bits=bits, pivot=1, zero_col=0))
            specs.append(dict(kind=0, m=s, n=s + 2, seed=112 + s, dtype_bits=bits, pivot=1, zero_col=0))
        specs.append(dict(kind=0, m=512, n=512, seed=12, dtype_bits=bits, pivot=1, zero_col=0))
        specs.append(dict(kind=0, m=400, n=130, seed=77, dtype_bits=bits, pivot=1, zero_col=0))
        for s, zc in ((50, 17), (130, 0), (300, 299)):
            specs.append(dict(kind=2, m=s, n=s, seed=212 + s, dtype_bits=bits, pivot=1, zero_col=zc))
        for s in (30, 130, 300):
            specs.append(dict(kind=1, m=s, n=s, seed=312 + s, dtype_bits=bits, pivot=0, zero_col=0))
        specs.append(dict(kind=5, m=130, n=130, seed=512, dtype_bits=bits, pivot=1, zero_col=5))
    specs.append(dict(kind=3, m=130, n=130, seed=0, dtype_bits=64, pivot=1, zero_col=0))
    specs.append(dict(kind=3, m=300, n=300, seed=0, dtype_bits=64, pivot=1, zero_col=0))
    specs.append(dict(kind=4, m=130, n=130, seed=412, dtype_bits=64, pivot=1, zero_col=0))
    specs.append(dict(kind=4, m=300, n=300, seed=413, dtype_bits=64, pivot=1, zero_col=0))

    for old in os.listdir(HERE):
        if old.endswith(".npz"):
            os.remove(os.path.join(HERE, old))
    rng = np.random.default_rng(0)
    for sp in specs:
        A = build_input(sp)
        F, ipiv, info = O.lu(A, pivot=bool(sp["pivot"]))
        kind = KINDS[sp["kind"]]
        if sp["pivot"] and kind != "nan":
            f = sla.lapack.dgetrf if A.dtype == np.float64 else sla.lapack.sgetrf
            _, lpiv, linfo = f(A)
            assert int(linfo) == info, (sp, linfo, info)
            assert np.array_equal(lpiv.astype(np.int64) + 1, ipiv), sp
        res_max, res_fro = (O.residual(A, F, ipiv) if info == 0 and kind != "nan" else (np.nan, np.nan))
        flat = F.ravel(order="F")
        idx = np.sort(rng.choice(flat.size, size=min(64, flat.size), replace=False)).astype(np.int64)
        name = f"{kind}_{'f64' if sp['dtype_bits'] == 64 else 'f32'}_{sp['m']}x{sp['n']}_p{sp['pivot']}.npz"
        np.savez(
            os.path.join(HERE, name),
            **{k: np.int64(v) for k, v in sp.items()},
            ipiv=ipiv, info=np.int64(info), sample_idx=idx, lu_sample=flat[idx].astype(np.float64),
            res_max=np.float64(res_max), res_fro=np.float64(res_fro),
        )
        print(f"{name:44s} info={info:4d} res_max={res_max:.2e}")


if __name__ == "__main__":
    main()
