// trsm.hip -- base case of  B <- L^-1 B  (Left / Lower / NoTrans / Unit) for a diagonal block of at most 64 rows.
//
// Replaces the TriangularSolve.ldiv!(UnitLowerTriangular(A11), A12) call of the reference
// (/root/reference/src/lu.jl:235 and :153; the arithmetic itself is third-party TriangularSolve.jl, so only the
// semantics are reproduced: unit diagonal implied, strict lower triangle of L read, B overwritten).
// The host driver (driver.cpp: trsm_rec) splits the triangle recursively so that >= 90 % of the TRSM flops run as
// MFMA GEMMs (gemm.hip); only the 64x64 diagonal blocks land here.
//
// Kernel: one thread per right-hand-side column, the 64-entry column held in registers (static indices via template
// recursion), L staged once per workgroup in LDS and read as wave-uniform broadcasts.  Loads/stores of B are
// coalesced along the row (lanes = consecutive columns of a row-major row).
// Roofline: fp64 vector FMA issue; algorithmic work nb^2 * nrhs flops per launch (a few % of the path's total).
#include "rflu_internal.hpp"

namespace rflu {

template <typename T, int I>
struct TrsmRow {
    static __device__ __forceinline__ void run(const T* sL, T (&x)[NB])
    {
        if constexpr (I < NB) {
            // four independent partial sums: a dependent fp64 FMA chain costs ~10+ cycles per link on one wave/SIMD
            T acc[4] = {T(0), T(0), T(0), T(0)};
#pragma unroll
            for (int k = 0; k < I; ++k) acc[k & 3] += sL[I * NB + k] * x[k];
            T s = x[I] - ((acc[0] + acc[1]) + (acc[2] + acc[3]));
            // pin row I's arithmetic before the next row's LDS reads: hipcc otherwise hoists all 2016 reads above the
            // FMA chains and spills ~14 KB per lane
            asm volatile("" : "+v"(s) : : "memory");
            x[I] = s;
            TrsmRow<T, I + 1>::run(sL, x);
        }
    }
};

template <typename T>
__global__ void __launch_bounds__(128) trsm_base_kernel(int nb, int64_t nrhs, const T* __restrict__ L, int64_t ldl,
                                                        T* __restrict__ B, int64_t ldb)
{
    __shared__ T sL[NB * NB];
    const int tid = threadIdx.x;
    {
        T tmp[NB * NB / 128];
#pragma unroll
        for (int it = 0; it < NB * NB / 128; ++it) {  // all 32 loads in flight together
            const int idx = it * 128 + tid;
            const int i = idx >> 6, k = idx & 63;
            tmp[it] = (i < nb && k < i) ? L[(int64_t)i * ldl + k] : T(0);
        }
#pragma unroll
        for (int it = 0; it < NB * NB / 128; ++it) sL[it * 128 + tid] = tmp[it];
    }
    const int64_t j = (int64_t)blockIdx.x * 128 + tid;
    T x[NB];
    if (j < nrhs) {
#pragma unroll
        for (int i = 0; i < NB; ++i) x[i] = (i < nb) ? B[(int64_t)i * ldb + j] : T(0);
    }
    __syncthreads();
    if (j >= nrhs) return;
    TrsmRow<T, 1>::run(sL, x);
#pragma unroll
    for (int i = 0; i < NB; ++i)
        if (i < nb) B[(int64_t)i * ldb + j] = x[i];
}

template <typename T>
int launch_trsm_base(Handle* h, int64_t nb, int64_t nrhs, const T* L, int64_t ldl, T* B, int64_t ldb)
{
    if (nb <= 1 || nrhs <= 0) return RFLU_OK;  // a 1x1 unit block is the identity
    if (nb > NB) {
        set_error("launch_trsm_base: block of %lld rows exceeds %d", (long long)nb, NB);
        return RFLU_ERR_ARG;
    }
    ProfScope ps(h, RFLU_K_TRSM, (double)nb * (double)nb * (double)nrhs);
    const unsigned grid = (unsigned)((nrhs + 127) / 128);
    hipLaunchKernelGGL(trsm_base_kernel<T>, dim3(grid), dim3(128), 0, h->stream, (int)nb, nrhs, L, ldl, B, ldb);
    RFLU_HIP(hipGetLastError());
    return RFLU_OK;
}

template int launch_trsm_base<double>(Handle*, int64_t, int64_t, const double*, int64_t, double*, int64_t);
template int launch_trsm_base<float>(Handle*, int64_t, int64_t, const float*, int64_t, float*, int64_t);

}  // namespace rflu
