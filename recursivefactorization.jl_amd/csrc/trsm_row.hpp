// trsm_row.hpp -- register-resident forward substitution with a 64x64 unit lower triangular block held in LDS, and the
// inversion of such a block (one thread per column of the inverse).  Shared by trsm.hip and laswp.hip.
#pragma once
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


// Inverse of the unit lower triangular nb x nb block Lblk (row-major, ldl) -> dense 64x64 row-major Linv (unit diagonal
// explicit, zeros above it, identity padding outside nb).  Executed by 64 threads (j = 0..63), sL = 64*64 elements of LDS.
template <typename T>
__device__ __forceinline__ void diag_inv_block(int nb, const T* __restrict__ Lblk, int64_t ldl, T* __restrict__ Linv,
                                               T* sL, int j)
{
    {
        T tmp[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) tmp[i] = (i < nb && j < i) ? Lblk[(int64_t)i * ldl + j] : T(0);
#pragma unroll
        for (int i = 0; i < NB; ++i) sL[i * NB + j] = tmp[i];
    }
    __builtin_amdgcn_s_waitcnt(0);  // the 64 threads are one wave: LDS writes above are ordered before the reads below
    __builtin_amdgcn_wave_barrier();
    T x[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) x[i] = (i == j) ? T(1) : T(0);
    TrsmRow<T, 1>::run(sL, x);
#pragma unroll
    for (int i = 0; i < NB; ++i) Linv[i * NB + j] = x[i];
}

}  // namespace rflu
