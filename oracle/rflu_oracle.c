/*
 * rflu_oracle.c -- CPU restatement of RecursiveFactorization.jl's recursive LU hot path.
 *
 * >>> TEST INFRASTRUCTURE, NOT PRODUCT CODE. <<<
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call this file.
 * The product path (recursivefactorization.jl_amd + librflu.so) never links or falls back to it.
 *
 * What it restates (all citations are into /root/reference/):
 *   src/lu.jl:97-130    lu!(A, ipiv, pivot, thread; check, blocksize, threshold)  -> rfo_lu_{f64,f32}
 *   src/lu.jl:141-156   _recurse! incl. the fat-matrix tail                       -> rfo_lu_*
 *   src/lu.jl:158-162   nsplit                                                    -> rfo_nsplit_*
 *   src/lu.jl:177-188   apply_permutation!                                        -> rfo_apply_permutation_*
 *   src/lu.jl:189-263   reckernel! (Toledo recursion, info / pivot offsets)       -> rfo_reckernel_* (static)
 *   src/lu.jl:265-284   schur_complement!                                         -> rfo_schur_complement_*
 *   src/lu.jl:290-338   _generic_lufact! (unblocked pivoted panel)                -> rfo_generic_lufact_*
 *   src/lu.jl:153,235   TRSM call sites; arithmetic is third-party TriangularSolve.jl (compat 0.2.5,
 *                       Project.toml:23, not vendored) -> semantic restatement rfo_trsm_unit_lower_*
 *
 * Pinning status: PARITY UNPINNED against the reference's own outputs (neither golden vectors nor a runnable reference
 * exist here); pinned against the reference's TEST PROPERTIES only.  The reference is Julia and cannot run in the authoring container (no julia, 7 un-vendored
 * registry dependencies), and its own tests hold NO golden vectors or known-answer L/U: test/runtests.jl checks
 * properties against LAPACK (`baselu = LinearAlgebra.lu`, :11,52) -- info equality (:15), max|L*U - A[p,:]| < 20*s*eps
 * (:19-20), solve of A[:,end] (:21-28), singular-column info (:59-64), NoPivot ipiv == 1:n (:70-84).  This oracle is
 * pinned against exactly those properties with LAPACK getrf (scipy) as comparator, plus ipiv equality with getrf on
 * tie-free inputs (tests/test_oracle.py, tests/golden/).  Entry-wise parity of L/U with the Julia implementation itself
 * is UNPINNED (the reference's L/U bits depend on LoopVectorization's SIMD width; only ipiv, info and the residual are
 * stable observables -- SURVEY.md section 8c).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

/* ---- counter-based input generator shared (bit-for-bit) with the numpy mirror in tests/ and the HIP fill kernel ----
 * element (i,j) of an m-row matrix: ctr = j*m + i;  z = splitmix64_mix(seed + (ctr+1)*0x9E3779B97F4A7C15);
 * u = (z >> 11) * 2^-53 in [0,1).  Stands in for Julia's rand(m,n) (test/runtests.jl:45), whose stream cannot be
 * reproduced without Julia. */
static inline uint64_t rfo_mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

double rfo_uniform01(uint64_t seed, uint64_t ctr)
{
    uint64_t z = rfo_mix64(seed + (ctr + 1) * 0x9E3779B97F4A7C15ULL);
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

void rfo_fill_uniform_f64(double* A, int64_t m, int64_t n, int64_t lda, uint64_t seed)
{
    for (int64_t j = 0; j < n; ++j)
        for (int64_t i = 0; i < m; ++i) A[i + j * lda] = rfo_uniform01(seed, (uint64_t)(j * m + i));
}

void rfo_fill_uniform_f32(float* A, int64_t m, int64_t n, int64_t lda, uint64_t seed)
{
    for (int64_t j = 0; j < n; ++j)
        for (int64_t i = 0; i < m; ++i) A[i + j * lda] = (float)rfo_uniform01(seed, (uint64_t)(j * m + i));
}

/* thread = Val(true) of the reference (Polyester @batch / @tturbo in apply_permutation!, TRSM and schur_complement!; the
 * panel stays serial, src/lu.jl:164-175,268): 1 = serial (Val(false)).  The threaded loops run over independent columns, so
 * results do not depend on the thread count. */
static int rfo_threads = 1;
void rfo_set_threads(int n) { rfo_threads = n > 1 ? n : 1; }
int rfo_get_threads(void) { return rfo_threads; }

#define T double
#define SFX f64
#define TABS fabs
#include "rflu_oracle_body.inc"
#undef T
#undef SFX
#undef TABS

#define T float
#define SFX f32
#define TABS fabsf
#include "rflu_oracle_body.inc"
#undef T
#undef SFX
#undef TABS
