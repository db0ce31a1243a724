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
// LDS images: element (i, j) at i*NB + ((j + i) & 63) (columns of row i rotated by i) -- with the plain row-major image every access
// that walks down a column puts all its lanes on one bank pair (round 4: see diag_inv_block16 below).
#define RFLU_INV4_IDX(i, j) ((i) * NB + ((((j) + (i))) & (NB - 1)))
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
            sL[RFLU_INV4_IDX(i, j)] = v;
            sX[RFLU_INV4_IDX(i, j)] = T(0);
        }
    }
    __syncthreads();
    // inv(A) (wave 0) and inv(C) (wave 1): lane j < 32 owns column j
    {
        const int wave = tid >> 6, lane = tid & 63;
        if (wave < 2 && lane < H) {
            const int o = wave * H;
            T x[H];
#pragma unroll
            for (int i = 0; i < H; ++i) x[i] = (i == lane) ? T(1) : T(0);
#pragma unroll
            for (int i = 1; i < H; ++i) {
                T acc[4] = {T(0), T(0), T(0), T(0)};
#pragma unroll
                for (int k = 0; k < i; ++k) acc[k & 3] += sL[RFLU_INV4_IDX(o + i, o + k)] * x[k];
                T sv = x[i] - ((acc[0] + acc[1]) + (acc[2] + acc[3]));
                asm volatile("" : "+v"(sv) : : "memory");   // see TrsmRow
                x[i] = sv;
            }
#pragma unroll
            for (int i = 0; i < H; ++i) sX[RFLU_INV4_IDX(o + i, o + lane)] = x[i];
        }
    }
    __syncthreads();
    // T = B * inv(A), parked in the unused upper right quadrant of sL;  thread -> row i, four columns
    {
        const int i = tid >> 3, j0 = (tid & 7) * 4;
        T t[4] = {T(0), T(0), T(0), T(0)};
#pragma unroll 8
        for (int k = 0; k < H; ++k) {
            const T bv = sL[RFLU_INV4_IDX(H + i, k)];
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] += bv * sX[RFLU_INV4_IDX(k, j0 + e)];
        }
        __syncthreads();   // everybody has read B before the quadrant next to it is overwritten (different quadrant: cheap safety)
#pragma unroll
        for (int e = 0; e < 4; ++e) sL[RFLU_INV4_IDX(i, H + j0 + e)] = t[e];
    }
    __syncthreads();
    // lower left block of the inverse: -inv(C) * T
    {
        const int i = tid >> 3, j0 = (tid & 7) * 4;
        T t[4] = {T(0), T(0), T(0), T(0)};
#pragma unroll 8
        for (int k = 0; k < H; ++k) {
            const T c = sX[RFLU_INV4_IDX(H + i, H + k)];
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] += c * sL[RFLU_INV4_IDX(k, H + j0 + e)];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) sX[RFLU_INV4_IDX(H + i, j0 + e)] = -t[e];
    }
    __syncthreads();
    {
        const int i = tid >> 2, c0 = (tid & 3) * 16;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int j = c0 + e;
            if (UPPER) Linv[i * NB + j] = sX[RFLU_INV4_IDX(j, i)] / (j < nb ? Lblk[(int64_t)j * ldl + j] : T(1));
            else Linv[i * NB + j] = sX[RFLU_INV4_IDX(i, j)];
        }
    }
}
#undef RFLU_INV4_IDX

// ---- round 4: the same inverse with three levels of blocking and the products on the matrix cores --------------------------------
// The interchange launch behind every leaf ends when its inverting workgroup does, and that workgroup was the long pole: 10 us
// against 3 us for the interchanges (stamps inside the launch; leaving the inverse out -- wrong factors, right timing -- N=16384
// 77.2 -> 74.9 ms, N=8192 25.3 -> 24.5).  Here the diagonal is inverted in 16x16 blocks (four waves side by side, one lane per
// column: 120 multiply-adds instead of 496), and the two levels above it, inv([A 0; B C]) = [inv(A) 0; -inv(C) B inv(A)  inv(C)],
// are 16x16x16 and 32x32x32 products on v_mfma_*_16x16x4 with operands straight from LDS (one tile per wave), on a bank-conflict
// free image (below).  Now 6.6 us: 2.0 waiting for the block from memory, 1.8 the 16x16 substitutions, 2.0 the four product steps,
// 0.8 the store (N=16384 77.2 -> 76.9 ms, N=8192 25.3 -> 25.0, N=4096 10.35 -> 10.15).  Unit lower triangle only (the non-unit
// upper variant keeps diag_inv_block4).
template <typename T>
struct InvMfma;
template <>
struct InvMfma<double> {
    typedef double acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t run(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int crow(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <>
struct InvMfma<float> {
    typedef float acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int crow(int lane, int r) { return 4 * (lane >> 4) + r; }
};

// LDS image of a 64x64 block with the columns of row i rotated by i: element (i, j) at i*NB + ((j + i) & 63).  With the plain
// row-major image (row pitch 512 bytes = a whole number of bank cycles) every access that walks down a column -- the A operand of
// the products, the thread-per-row load and store of the block -- puts all its lanes on ONE bank pair: the load and the store of the
// block were 64-way conflicted, the operand reads 16-way (stamps: 1.1 us per 32x32x32 product, 10 us for the whole inverse).
#define RFLU_INV_IDX(i, j) ((i) * NB + ((((j) + (i))) & (NB - 1)))

// one 16x16 tile of  C = (NEG ? -1 : 1) * A(16 x K) * B(K x 16): A = rows ar.., columns ac.. of MA, B = rows br.., columns bc.. of MB,
// C = rows cr.., columns cc.. of MC (rotated images); the calling wave only
template <typename T, int K, bool NEG>
__device__ __forceinline__ void inv_tile_prod(const T* MA, int ar, int ac, const T* MB, int br, int bc, T* MC, int cr, int cc, int lane)
{
    typename InvMfma<T>::acc_t acc = {T(0), T(0), T(0), T(0)};
    const int li = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int k0 = 0; k0 < K; k0 += 4)
        acc = InvMfma<T>::run(MA[RFLU_INV_IDX(ar + li, ac + k0 + lk)], MB[RFLU_INV_IDX(br + k0 + lk, bc + li)], acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) MC[RFLU_INV_IDX(cr + InvMfma<T>::crow(lane, r), cc + li)] = NEG ? -acc[r] : acc[r];
}

// sL, sX: NB*NB elements of LDS each.  All 256 threads of the workgroup must call this.
// the block's strictly lower part, 16 entries of row tid >> 2 per thread (zero elsewhere, also outside nb: the padding inverts to the
// identity): a caller that has other memory latencies to wait for requests it first and hands it to diag_inv_block16_pre
template <typename T>
__device__ __forceinline__ void diag_inv_load16(int nb, const T* __restrict__ Lblk, int64_t ldl, int tid, T (&v)[16])
{
    const int i = tid >> 2, c0 = (tid & 3) * 16;
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = (i < nb && c0 + e < i) ? Lblk[(int64_t)i * ldl + c0 + e] : T(0);
}
template <typename T>
__device__ __forceinline__ void diag_inv_block16_pre(const T (&v)[16], T* __restrict__ Linv, T* sL, T* sX, int tid);
template <typename T>
__device__ __forceinline__ void diag_inv_block16(int nb, const T* __restrict__ Lblk, int64_t ldl, T* __restrict__ Linv, T* sL, T* sX,
                                                 int tid)
{
    T v[16];
    diag_inv_load16<T>(nb, Lblk, ldl, tid, v);
    diag_inv_block16_pre<T>(v, Linv, sL, sX, tid);
}
template <typename T>
__device__ __forceinline__ void diag_inv_block16_pre(const T (&v)[16], T* __restrict__ Linv, T* sL, T* sX, int tid)
{
    constexpr int Q = NB / 4, H = NB / 2;
    const int wave = tid >> 6, lane = tid & 63;
    {
        const int i = tid >> 2, c0 = (tid & 3) * 16;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            sL[RFLU_INV_IDX(i, c0 + e)] = v[e];
            sX[RFLU_INV_IDX(i, c0 + e)] = T(0);
        }
    }
    __syncthreads();
    if (lane < Q) {   // the four 16x16 diagonal blocks: wave w inverts block w, lane j owns column j of the inverse
        const int o = wave * Q;
        T x[Q];
#pragma unroll
        for (int i = 0; i < Q; ++i) x[i] = (i == lane) ? T(1) : T(0);
#pragma unroll
        for (int i = 1; i < Q; ++i) {
            T acc[4] = {T(0), T(0), T(0), T(0)};
#pragma unroll
            for (int k = 0; k < i; ++k) acc[k & 3] += sL[RFLU_INV_IDX(o + i, o + k)] * x[k];
            x[i] -= (acc[0] + acc[1]) + (acc[2] + acc[3]);
        }
#pragma unroll
        for (int i = 0; i < Q; ++i) sX[RFLU_INV_IDX(o + i, o + lane)] = x[i];
    }
    __syncthreads();
    // the two 32x32 diagonal blocks p = 0, 1 (waves 0, 1):  T = B inv(A), parked in the zero quadrant next to A inside sL ...
    if (wave < 2) {
        const int o = wave * H;
        inv_tile_prod<T, Q, false>(sL, o + Q, o, sX, o, o, sL, o, o + Q, lane);
    }
    __syncthreads();
    if (wave < 2) {   // ... and -inv(C) T into the inverse
        const int o = wave * H;
        inv_tile_prod<T, Q, true>(sX, o + Q, o + Q, sL, o, o + Q, sX, o + Q, o, lane);
    }
    __syncthreads();
    // the 64x64 level, one 16x16 tile per wave: T = B inv(A) (B = rows 32.., columns 0..31 of L; parked in the zero quadrant of sL)
    const int ti = (wave >> 1) * Q, tj = (wave & 1) * Q;
    inv_tile_prod<T, H, false>(sL, H + ti, 0, sX, 0, tj, sL, ti, H + tj, lane);
    __syncthreads();
    inv_tile_prod<T, H, true>(sX, H + ti, H, sL, 0, H + tj, sX, H + ti, tj, lane);   // lower left quadrant of the inverse: -inv(C) T
    __syncthreads();
    {
        const int i = tid >> 2, c0 = (tid & 3) * 16;
#pragma unroll
        for (int e = 0; e < 16; ++e) Linv[i * NB + c0 + e] = sX[RFLU_INV_IDX(i, c0 + e)];
    }
}
#undef RFLU_INV_IDX

}  // namespace rflu
