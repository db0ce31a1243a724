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
#include "trsm_row.hpp"

namespace rflu {

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


// =====================================================================================================================
// Fused strip TRSM for triangles of up to 256 rows:  B <- L^-1 B  in ONE launch.
//
// The recursive splitting in driver.cpp (trsm_rec) turns a 256-row triangle into 4 base solves + 3 GEMMs = 7 dependent
// launches of ~20 us each (all latency, no work).  Here one workgroup owns a strip of 32 right-hand-side columns and
// walks the 64-row blocks top to bottom (left-looking):
//     acc   = B_d - sum_{e<d} L_de * X_e          MFMA, A fragments straight from global/L2, X_e from LDS
//     X_d   = inv(L_dd) * acc                     MFMA with the pre-inverted 64x64 diagonal block (diag_inv_kernel)
// The inverse of a unit lower triangular block with |l_ij| <= 1 (partial pivoting) is what MAGMA/rocBLAS-style TRSMs
// use for their diagonal blocks as well; the off-diagonal 90+ % of the flops are plain GEMM.
// Geometry: 256 threads = 4 waves; wave w owns rows [16w,16w+16) of the 64-row block x 32 columns = 2 MFMA fragments.
// LDS: up to 3 solved blocks X_e as MFMA B operands (the 4th is staged in block 0's slot), [64][48] each (row stride 48 == 16 mod 32
// -> conflict-free reads).
// =====================================================================================================================
constexpr int TF_MAXN = 256;
constexpr int TF_COLS = 32;
constexpr int TF_XLD = TF_COLS + 16;

template <typename T>
struct MfmaT;
template <>
struct MfmaT<double> {
    typedef double acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t run(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int crow(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <>
struct MfmaT<float> {
    typedef float acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int crow(int lane, int r) { return 4 * (lane >> 4) + r; }
};

template <typename T>
__global__ void __launch_bounds__(256) trsm_fused_kernel(int n, int64_t nrhs, const T* __restrict__ L, int64_t ldl,
                                                         const T* __restrict__ Linv, T* __restrict__ B, int64_t ldb)
{
    typedef typename MfmaT<T>::acc_t acc_t;
    // Three slots for four blocks: the LAST block of a 256-row triangle is staged in block 0's slot -- by then nobody multiplies by
    // X_0 any more.  72 KB instead of 96 KB (Float64): TWO workgroups per CU.  The update stream's block-row solves (496 workgroups
    // on 224 CUs: 2.2 rounds of one workgroup per CU) were 82 us per launch, two launches per block column on the stream that sets
    // the pace of the update-bound part (round 4; with 24 KB of LDS ballast, i.e. the old occupancy: N=16384 77.46-77.75 vs 77.28-77.48 ms).
    constexpr int TF_SLOTS = TF_MAXN / NB - 1;
    __shared__ T Xs[TF_SLOTS * NB * TF_XLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t col0 = (int64_t)blockIdx.x * TF_COLS;
    const int nblk = (n + NB - 1) / NB;
    const int fi = lane & 15, fk = lane >> 4;      // A fragment: row fi, k-offset fk ; B fragment: k-offset fk, col fi
    const int arow = wave * 16 + fi;               // row of this lane's A fragment inside the 64-row block

    for (int d = 0; d < nblk; ++d) {
        const int rows_d = min(NB, n - d * NB);
        // ---- acc = B_d (C-fragment layout: row = wave*16 + crow(lane,r), col = t*16 + fi) ----
        acc_t acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wave * 16 + MfmaT<T>::crow(lane, r);
                const int64_t col = col0 + t * 16 + fi;
                acc[t][r] = (row < rows_d && col < nrhs) ? B[(int64_t)(d * NB + row) * ldb + col] : T(0);
            }
        // ---- acc -= L_de * X_e ----
        for (int e = 0; e < d; ++e) {
            T a[16];
            const T* Lp = L + (int64_t)(d * NB + arow) * ldl + e * NB + fk;
            const bool rok = arow < rows_d;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) a[kk] = rok ? -Lp[kk * 4] : T(0);
            const T* Xe = Xs + e * NB * TF_XLD;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const T b0 = Xe[(kk * 4 + fk) * TF_XLD + fi];
                const T b1 = Xe[(kk * 4 + fk) * TF_XLD + 16 + fi];
                acc[0] = MfmaT<T>::run(a[kk], b0, acc[0]);
                acc[1] = MfmaT<T>::run(a[kk], b1, acc[1]);
            }
        }
        // ---- stage acc as a B operand, then X_d = inv(L_dd) * acc ----
        if (d >= TF_SLOTS) __syncthreads();   // every wave is done with X_0 before its slot is reused (workgroup-uniform)
        T* Xd = Xs + (d < TF_SLOTS ? d : 0) * NB * TF_XLD;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) Xd[(wave * 16 + MfmaT<T>::crow(lane, r)) * TF_XLD + t * 16 + fi] = acc[t][r];
        T ai[16];
        {
            const T* Ip = Linv + (int64_t)d * NB * NB + arow * NB + fk;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) ai[kk] = Ip[kk * 4];
        }
        __syncthreads();
        acc_t x[2] = {acc_t{T(0), T(0), T(0), T(0)}, acc_t{T(0), T(0), T(0), T(0)}};
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const T b0 = Xd[(kk * 4 + fk) * TF_XLD + fi];
            const T b1 = Xd[(kk * 4 + fk) * TF_XLD + 16 + fi];
            x[0] = MfmaT<T>::run(ai[kk], b0, x[0]);
            x[1] = MfmaT<T>::run(ai[kk], b1, x[1]);
        }
        __syncthreads();  // everybody has read the staged acc
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wave * 16 + MfmaT<T>::crow(lane, r);
                const int64_t col = col0 + t * 16 + fi;
                Xd[row * TF_XLD + t * 16 + fi] = x[t][r];
                if (row < rows_d && col < nrhs) B[(int64_t)(d * NB + row) * ldb + col] = x[t][r];
            }
        __syncthreads();  // X_d visible to the following block rows
    }
}

// inverse of the unit lower triangular nb x nb block at Lblk (row-major, ldl) -> dense 64x64 row-major Linv
// (unit diagonal explicit, zeros above it and outside nb; identity padding so that partial blocks behave).
// Batched: block b inverts the diagonal block starting at row/column 64*b of the n x n triangle L.
template <typename T>
__global__ void __launch_bounds__(256) diag_inv_kernel(int n, const T* __restrict__ L, int64_t ldl, T* __restrict__ Linv_all)
{
    __shared__ T sL[NB * NB];
    __shared__ T sX[NB * NB];
    const int b = blockIdx.x;
    const int nb = min(NB, n - b * NB);
    diag_inv_block16<T>(nb, L + (int64_t)b * NB * ldl + b * NB, ldl, Linv_all + (size_t)b * NB * NB, sL, sX, threadIdx.x);
}

// invert every 64x64 diagonal block of the n x n unit lower triangle L into Linv[0 .. ceil(n/64))
template <typename T>
int launch_diag_inv(Handle* h, int64_t n, const T* L, int64_t ldl, T* Linv)
{
    if (n <= 0) return RFLU_OK;
    ProfScope ps(h, RFLU_K_TRSM, (double)n * NB * NB / 3.0);
    hipLaunchKernelGGL(diag_inv_kernel<T>, dim3((unsigned)((n + NB - 1) / NB)), dim3(256), 0, h->stream, (int)n, L, ldl, Linv);
    RFLU_HIP(hipGetLastError());
    return RFLU_OK;
}

template <typename T>
int launch_trsm_fused(Handle* h, int64_t n, int64_t nrhs, const T* L, int64_t ldl, const T* Linv, T* B, int64_t ldb)
{
    if (n <= 0 || nrhs <= 0) return RFLU_OK;
    if (n > TF_MAXN) { set_error("launch_trsm_fused: %lld rows exceed %d", (long long)n, TF_MAXN); return RFLU_ERR_ARG; }
    ProfScope ps(h, RFLU_K_TRSM, (double)n * (double)n * (double)nrhs);
    const unsigned grid = (unsigned)((nrhs + TF_COLS - 1) / TF_COLS);
    hipLaunchKernelGGL(trsm_fused_kernel<T>, dim3(grid), dim3(256), 0, h->stream, (int)n, nrhs, L, ldl, Linv, B, ldb);
    RFLU_HIP(hipGetLastError());
    return RFLU_OK;
}

// B <- inv(L) * B for ONE diagonal block (n <= 64 rows) whose inverse is already known: no LDS at all, so the workgroups slip
// onto CUs whose LDS the update GEMM has taken (the side streams of factor_leafwise run next to it; the 98 KB of
// trsm_fused_kernel waited for a whole CU to drain, 180 us per launch).  One workgroup = 64 right-hand-side columns; every
// wave loads the complete 64 x 64 B tile as MFMA B operands straight from global memory (16 lanes = 128 contiguous bytes of
// a row), the barrier separates the last read of B from its first overwrite, wave w produces rows 16w .. 16w+15.
template <typename T>
__global__ void __launch_bounds__(256) trsm_inv64_kernel(int n, int64_t nrhs, const T* __restrict__ Linv, T* __restrict__ B,
                                                         int64_t ldb)
{
    typedef typename MfmaT<T>::acc_t acc_t;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 15, fk = lane >> 4;
    const int64_t col0 = (int64_t)blockIdx.x * 64;
    T b[16][4];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int row = kk * 4 + fk;
            const int64_t col = col0 + t * 16 + fi;
            b[kk][t] = (row < n && col < nrhs) ? B[(int64_t)row * ldb + col] : T(0);
        }
    T ai[16];
    {
        const T* Ip = Linv + (wave * 16 + fi) * NB + fk;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) ai[kk] = Ip[kk * 4];
    }
    __syncthreads();
    acc_t x[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) x[t] = acc_t{T(0), T(0), T(0), T(0)};
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)
#pragma unroll
        for (int t = 0; t < 4; ++t) x[t] = MfmaT<T>::run(ai[kk], b[kk][t], x[t]);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = wave * 16 + MfmaT<T>::crow(lane, r);
            const int64_t col = col0 + t * 16 + fi;
            if (row < n && col < nrhs) B[(int64_t)row * ldb + col] = x[t][r];
        }
}

template <typename T>
int launch_trsm_inv64(Handle* h, int64_t n, int64_t nrhs, const T* Linv, T* B, int64_t ldb)
{
    if (n <= 1 || nrhs <= 0) return RFLU_OK;
    if (n > NB) { set_error("launch_trsm_inv64: %lld rows exceed %d", (long long)n, NB); return RFLU_ERR_ARG; }
    ProfScope ps(h, RFLU_K_TRSM, (double)n * (double)n * (double)nrhs);
    hipLaunchKernelGGL(trsm_inv64_kernel<T>, dim3((unsigned)((nrhs + 63) / 64)), dim3(256), 0, h->stream, (int)n, nrhs, Linv, B, ldb);
    RFLU_HIP(hipGetLastError());
    return RFLU_OK;
}
template int launch_trsm_inv64<double>(Handle*, int64_t, int64_t, const double*, double*, int64_t);
template int launch_trsm_inv64<float>(Handle*, int64_t, int64_t, const float*, float*, int64_t);

template int launch_diag_inv<double>(Handle*, int64_t, const double*, int64_t, double*);
template int launch_diag_inv<float>(Handle*, int64_t, const float*, int64_t, float*);
template int launch_trsm_fused<double>(Handle*, int64_t, int64_t, const double*, int64_t, const double*, double*, int64_t);
template int launch_trsm_fused<float>(Handle*, int64_t, int64_t, const float*, int64_t, const float*, float*, int64_t);


// =====================================================================================================================
// Upper-triangular (non-unit) solve  B <- U^-1 B  -- the second leg of ldiv!(F::LU, B) (SURVEY.md 8f "next" row 1:
// the reference's own ldiv! for NotIPIV, /root/reference/src/lu.jl:60-64, and stdlib getrs for pivoted factors).
// Host recursion in driver.cpp (triu_solve_rec): bottom block first, GEMM update of the rows above, then the top block;
// this kernel is the <= 64-row base: one thread per right-hand-side column, back substitution in registers, U in LDS.
// =====================================================================================================================
template <typename T, int I>
struct TriuRow {
    static __device__ __forceinline__ void run(const T* sU, T (&x)[NB])
    {
        if constexpr (I >= 0) {
            T acc[4] = {T(0), T(0), T(0), T(0)};
#pragma unroll
            for (int k = I + 1; k < NB; ++k) acc[k & 3] += sU[I * NB + k] * x[k];
            T s = (x[I] - ((acc[0] + acc[1]) + (acc[2] + acc[3]))) / sU[I * NB + I];
            asm volatile("" : "+v"(s) : : "memory");
            x[I] = s;
            TriuRow<T, I - 1>::run(sU, x);
        }
    }
};

template <typename T>
__global__ void __launch_bounds__(128) triu_base_kernel(int nb, int64_t nrhs, const T* __restrict__ U, int64_t ldu,
                                                        T* __restrict__ B, int64_t ldb)
{
    __shared__ T sU[NB * NB];
    const int tid = threadIdx.x;
    {
        T tmp[NB * NB / 128];
#pragma unroll
        for (int it = 0; it < NB * NB / 128; ++it) {
            const int idx = it * 128 + tid;
            const int i = idx >> 6, k = idx & 63;
            // rows/columns beyond nb behave like an identity block
            tmp[it] = (i < nb && k < nb) ? (k >= i ? U[(int64_t)i * ldu + k] : T(0)) : (i == k ? T(1) : T(0));
        }
#pragma unroll
        for (int it = 0; it < NB * NB / 128; ++it) sU[it * 128 + tid] = tmp[it];
    }
    const int64_t j = (int64_t)blockIdx.x * 128 + tid;
    T x[NB];
    if (j < nrhs) {
#pragma unroll
        for (int i = 0; i < NB; ++i) x[i] = (i < nb) ? B[(int64_t)i * ldb + j] : T(0);
    }
    __syncthreads();
    if (j >= nrhs) return;
    TriuRow<T, NB - 1>::run(sU, x);
#pragma unroll
    for (int i = 0; i < NB; ++i)
        if (i < nb) B[(int64_t)i * ldb + j] = x[i];
}

template <typename T>
int launch_triu_base(Handle* h, int64_t nb, int64_t nrhs, const T* U, int64_t ldu, T* B, int64_t ldb)
{
    if (nb <= 0 || nrhs <= 0) return RFLU_OK;
    if (nb > NB) { set_error("launch_triu_base: block of %lld rows exceeds %d", (long long)nb, NB); return RFLU_ERR_ARG; }
    ProfScope ps(h, RFLU_K_TRSM, (double)nb * (double)nb * (double)nrhs);
    const unsigned grid = (unsigned)((nrhs + 127) / 128);
    hipLaunchKernelGGL(triu_base_kernel<T>, dim3(grid), dim3(128), 0, h->stream, (int)nb, nrhs, U, ldu, B, ldb);
    RFLU_HIP(hipGetLastError());
    return RFLU_OK;
}
template int launch_triu_base<double>(Handle*, int64_t, int64_t, const double*, int64_t, double*, int64_t);
template int launch_triu_base<float>(Handle*, int64_t, int64_t, const float*, int64_t, float*, int64_t);

}  // namespace rflu
