// trsm_row.hpp -- register-resident forward substitution with a 64x64 unit lower triangular block held in LDS, and the
// inversion of such a block.  Shared by trsm.hip and laswp.hip.
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


// ---- inverse of the unit lower triangular nb x nb block Lblk (row-major, ldl) -> dense 64x64 row-major Linv (unit diagonal
// explicit, zeros above it, identity padding outside nb), by a whole workgroup (256 threads), two levels of blocking ------
// L = [A 0; B C] with 32x32 blocks:  inv(L) = [inv(A) 0; -inv(C)*B*inv(A)  inv(C)].  inv(A) and inv(C) are computed side by
// side by two waves (one lane per column, 496 multiply-adds each instead of 2016), the two 32x32x32 products by all 256
// threads.  ~6 us instead of the ~20 us of one wave solving 64 columns -- it is the long pole of every leaf's interchange
// launch.
template <typename T, int I, int N>
struct TrsmRowG {
    // forward substitution with the N x N unit lower block at sL (leading dimension ld): x <- inv(L) x, rows I..N-1
    static __device__ __forceinline__ void run(const T* sL, int ld, T (&x)[N])
    {
        if constexpr (I < N) {
            T acc[4] = {T(0), T(0), T(0), T(0)};
#pragma unroll
            for (int k = 0; k < I; ++k) acc[k & 3] += sL[I * ld + k] * x[k];
            T s = x[I] - ((acc[0] + acc[1]) + (acc[2] + acc[3]));
            asm volatile("" : "+v"(s) : : "memory");   // see TrsmRow
            x[I] = s;
            TrsmRowG<T, I + 1, N>::run(sL, ld, x);
        }
    }
};

// sL, sX: NB*NB elements of LDS each.  All 256 threads of the workgroup must call this.
// UPPER: the block is the NON-unit UPPER triangle U of the same storage and the result is inv(U): U' = L'*D with the unit lower
// L'[i][j] = U[j][i] / U[j][j], so inv(U)[i][j] = inv(L')[j][i] / U[j][j] -- the same machinery on the scaled transpose.
template <typename T, bool UPPER = false>
__device__ __forceinline__ void diag_inv_block4(int nb, const T* __restrict__ Lblk, int64_t ldl, T* __restrict__ Linv,
                                                T* sL, T* sX, int tid)
{
    constexpr int H = NB / 2;
    // strictly lower part of the block into sL (zero elsewhere, also outside nb: the padding inverts to the identity)
    {
        const int i = tid >> 2, c0 = (tid & 3) * 16;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int j = c0 + e;
            T v = T(0);
            if (i < nb && j < i) v = UPPER ? Lblk[(int64_t)j * ldl + i] / Lblk[(int64_t)j * ldl + j] : Lblk[(int64_t)i * ldl + j];
            sL[i * NB + j] = v;
            sX[i * NB + j] = T(0);
        }
    }
    __syncthreads();
    // inv(A) (wave 0) and inv(C) (wave 1): lane j < 32 owns column j
    {
        const int wave = tid >> 6, lane = tid & 63;
        if (wave < 2 && lane < H) {
            const T* blk = sL + (wave * H) * NB + wave * H;
            T x[H];
#pragma unroll
            for (int i = 0; i < H; ++i) x[i] = (i == lane) ? T(1) : T(0);
            TrsmRowG<T, 1, H>::run(blk, NB, x);
#pragma unroll
            for (int i = 0; i < H; ++i) sX[(wave * H + i) * NB + wave * H + lane] = x[i];
        }
    }
    __syncthreads();
    // T = B * inv(A), parked in the unused upper right quadrant of sL;  thread -> row i, four columns
    {
        const int i = tid >> 3, j0 = (tid & 7) * 4;
        T t[4] = {T(0), T(0), T(0), T(0)};
#pragma unroll 8
        for (int k = 0; k < H; ++k) {
            const T b = sL[(H + i) * NB + k];
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] += b * sX[k * NB + j0 + e];
        }
        __syncthreads();   // everybody has read B before the quadrant next to it is overwritten (different quadrant: cheap safety)
#pragma unroll
        for (int e = 0; e < 4; ++e) sL[i * NB + H + j0 + e] = t[e];
    }
    __syncthreads();
    // lower left block of the inverse: -inv(C) * T
    {
        const int i = tid >> 3, j0 = (tid & 7) * 4;
        T t[4] = {T(0), T(0), T(0), T(0)};
#pragma unroll 8
        for (int k = 0; k < H; ++k) {
            const T c = sX[(H + i) * NB + H + k];
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] += c * sL[k * NB + H + j0 + e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) sX[(H + i) * NB + j0 + e] = -t[e];
    }
    __syncthreads();
    {
        const int i = tid >> 2, c0 = (tid & 3) * 16;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int j = c0 + e;
            if (UPPER) Linv[i * NB + j] = sX[j * NB + i] / (j < nb ? Lblk[(int64_t)j * ldl + j] : T(1));
            else Linv[i * NB + j] = sX[i * NB + j];
        }
    }
}

}  // namespace rflu
