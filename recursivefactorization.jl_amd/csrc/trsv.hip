// trsv.hip -- the solve step for FEW right-hand sides:  B <- U^-1 L^-1 B  after lu!  (ldiv!(F, b), what LinearSolve's solve!
// calls right after the factorization; /root/reference/src/lu.jl:60-64 and test/runtests.jl:21-28, 116-128).
//
// With one right-hand side a triangular solve is n/64 DEPENDENT block steps of almost no work; as separate launches
// (driver.cpp: trsm_rec / triu_solve_rec, right for many right-hand sides where the GEMMs carry the work) that is ~1000
// launches, 33 ms at n = 16384, while the factors are only 2 GB of HBM reads.  Here ONE cooperative launch per triangle
// runs the whole chain as a dataflow over 64-row blocks:
//   * block r of the right-hand sides belongs to workgroup r mod G (G = min(n/64, 256), all co-resident);
//   * stage d: the owner of block d finishes it  (x_d = inv(D_dd) * b_d, a 64x64 matrix-vector product with the
//     pre-inverted diagonal block), stores x_d and raises flag d; every workgroup waits for the flag, fetches x_d and
//     subtracts  T_rd * x_d  from the blocks r it owns beyond d -- nearest block first, so the next owner publishes
//     x_{d+1} before it touches the rest;
//   * no grid barrier: one flag hop plus two 64x64 products per stage on the critical path, the bulk of the traffic
//     (each 64x64 block of the triangle read exactly once) trails behind it.
// Roofline: HBM -- algorithmic bytes = sizeof(T) * n^2 / 2 per triangle; latency floor = n/64 stages x (one cross-workgroup
// hop + two block products).  Up to TV_NR right-hand sides ride along in one pass.
#include "rflu_internal.hpp"

namespace rflu {

constexpr int TV_NR = 8;            // right-hand sides per pass
constexpr int TV_THREADS = 256;     // 4 waves: lane = row of the 64x64 block, wave = quarter of its 64 columns
constexpr int TV_MAX_WGS = 256;
constexpr int TV_SPIN_LIMIT = 1 << 22;

// inverse of the 64x64 UPPER (non-unit) diagonal blocks of the factored matrix: block b -> dense row-major 64x64, identity
// padding outside the matrix.  One wave per block, lane = column of the inverse, back substitution through LDS.
template <typename T>
__global__ void __launch_bounds__(64) triu_inv_kernel(int n, const T* __restrict__ R, int64_t ld, T* __restrict__ Uinv)
{
    __shared__ T sU[NB * (NB + 1)];
    __shared__ T sX[NB * (NB + 1)];
    const int j = threadIdx.x, b = blockIdx.x;
    const int nbk = min(NB, n - b * NB);
    const T* Ub = R + (int64_t)b * NB * ld + b * NB;
    for (int i = 0; i < NB; ++i) {
        T v = (i == j) ? T(1) : T(0);
        if (i < nbk && j < nbk && j >= i) v = Ub[(int64_t)i * ld + j];
        sU[i * (NB + 1) + j] = v;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    // column j of the inverse: x_j = 1/u_jj, x_i = -(sum_{k=i+1..j} u_ik x_k) / u_ii for i < j, zero below the diagonal
    for (int i = NB - 1; i >= 0; --i) {
        T x = T(0);
        if (i == j) x = T(1) / sU[i * (NB + 1) + i];
        else if (i < j) {
            T s = T(0);
            for (int k = i + 1; k <= j; ++k) s += sU[i * (NB + 1) + k] * sX[k * (NB + 1) + j];
            x = -s / sU[i * (NB + 1) + i];
        }
        sX[i * (NB + 1) + j] = x;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    T* out = Uinv + (size_t)b * NB * NB;
    for (int i = 0; i < NB; ++i) out[i * NB + j] = sX[i * (NB + 1) + j];
}

// A 64x64 block M (row-major, leading dimension ldm) as registers: thread (i = tid & 63, q = tid >> 6) holds row i, columns
// [16q, 16q+16) -- 16 contiguous elements, four 16-byte loads.  Rows >= rows_ok / columns >= cols_ok read as zero.
template <typename T>
__device__ __forceinline__ void block_load(const T* __restrict__ M, int64_t ldm, int rows_ok, int cols_ok, int tid,
                                           T (&m)[16])
{
    constexpr int VW = 16 / (int)sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(VW)));
    const int i = tid & 63, q = tid >> 6;
    const T* Mp = M + (int64_t)i * ldm + q * 16;
    const bool vec_ok = ((reinterpret_cast<uintptr_t>(Mp) & 15) == 0);
    if (i < rows_ok && q * 16 + 16 <= cols_ok && vec_ok) {
#pragma unroll
        for (int v = 0; v < 16 / VW; ++v) {
            const vec_t x = *reinterpret_cast<const vec_t*>(Mp + v * VW);
#pragma unroll
            for (int e = 0; e < VW; ++e) m[v * VW + e] = x[e];
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) m[k] = (i < rows_ok && q * 16 + k < cols_ok) ? Mp[k] : T(0);
    }
}

// y[i][c] = sum_k M[i][k] * xs[k][c] with M in registers (block_load); the four column quarters meet in LDS `part`.
// y is valid in the threads with q == 0.
template <typename T>
__device__ __forceinline__ void block_gemv(const T (&m)[16], const T* xs, T* part, int tid, T (&y)[TV_NR])
{
    const int i = tid & 63, q = tid >> 6;
    T acc[TV_NR];
#pragma unroll
    for (int c = 0; c < TV_NR; ++c) acc[c] = T(0);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
#pragma unroll
        for (int c = 0; c < TV_NR; ++c) acc[c] += m[k] * xs[(q * 16 + k) * TV_NR + c];
    }
#pragma unroll
    for (int c = 0; c < TV_NR; ++c) part[(q * NB + i) * TV_NR + c] = acc[c];
    __syncthreads();
    if (q == 0) {
#pragma unroll
        for (int c = 0; c < TV_NR; ++c)
            y[c] = (part[i * TV_NR + c] + part[(NB + i) * TV_NR + c]) +
                   (part[(2 * NB + i) * TV_NR + c] + part[(3 * NB + i) * TV_NR + c]);
    }
    __syncthreads();  // `part` may be reused
}

constexpr int TV_KOWN = 4;   // blocks of right-hand sides a workgroup keeps in LDS (n <= 64 * 256 * 4 rows)

// x_d travels between workgroups as data-tagged 16-byte granules {value, tag} written by write-through (sc1) stores and
// polled by sc1 loads -- the same fence-free exchange as the pivot rows of the panel kernel (panel.hip).  A release fence
// per stage would cost a write-back of the L2 (measured: 23 us per stage with __threadfence + flag vs 3 us like this).
typedef unsigned tv_u4 __attribute__((ext_vector_type(4)));
constexpr int TV_AUX_SC1 = 16;
__device__ __forceinline__ void tv_store(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned tag, double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const tv_u4 x = {(unsigned)(b >> 32), tag, (unsigned)b, tag};
    __builtin_amdgcn_raw_buffer_store_b128(x, r, off, 0, TV_AUX_SC1);
}
__device__ __forceinline__ void tv_store(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned tag, float v)
{
    const tv_u4 x = {__float_as_uint(v), tag, 0u, tag};
    __builtin_amdgcn_raw_buffer_store_b128(x, r, off, 0, TV_AUX_SC1);
}
__device__ __forceinline__ bool tv_load(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned tag, double& v)
{
    const tv_u4 x = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, TV_AUX_SC1);
    v = __longlong_as_double((long long)(((unsigned long long)x[0] << 32) | (unsigned long long)x[2]));
    return x[1] == tag && x[3] == tag;
}
__device__ __forceinline__ bool tv_load(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned tag, float& v)
{
    const tv_u4 x = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, TV_AUX_SC1);
    v = __uint_as_float(x[0]);
    return x[1] == tag && x[3] == tag;
}

template <typename T, bool UPPER>
__global__ void __launch_bounds__(TV_THREADS) trsv_coop_kernel(int n, int nrhs, const T* __restrict__ R, int64_t ld,
                                                               const T* __restrict__ Dinv, T* X, int64_t ldx,
                                                               void* xchg, unsigned xchg_bytes, unsigned tag, int64_t* err)
{
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(xchg, 0, xchg_bytes, 0x00020000);
    __shared__ T xs[NB * TV_NR];
    __shared__ T part[4 * NB * TV_NR];
    __shared__ T bacc[TV_KOWN][NB * TV_NR];   // the owned blocks of B: they stay here until they become x
    __shared__ int s_dead;
    const int tid = threadIdx.x, i = tid & 63, q = tid >> 6;
    const int nb = (n + NB - 1) / NB;
    const int G = gridDim.x, w = blockIdx.x;
    if (tid == 0) s_dead = 0;
    // owned block j is block w + j*G; load them all
    for (int j = 0; j < TV_KOWN; ++j) {
        const int r = w + j * G;
        for (int e = tid; e < NB * TV_NR; e += TV_THREADS) {
            const int k = e / TV_NR, c = e % TV_NR;
            bacc[j][e] = (r < nb && r * NB + k < n && c < nrhs) ? X[(int64_t)(r * NB + k) * ldx + c] : T(0);
        }
    }
    // the diagonal inverse of the block this workgroup solves next, and the off-diagonal block of its nearest owned block
    // for the first stage: both requested before the first flag is waited for
    const int d0 = UPPER ? nb - 1 : 0;
    auto nearest = [&](int d) -> int {   // nearest owned block strictly beyond stage d (or -1)
        if (!UPPER) {
            const int r = d + 1 + ((w - (d + 1)) % G + G) % G;
            return r < nb ? r : -1;
        }
        if (d - 1 < 0) return -1;
        const int r = d - 1 - (((d - 1) - w) % G + G) % G;
        return r >= 0 ? r : -1;
    };
    auto first_own = [&]() -> int {      // the first block this workgroup will have to solve
        if (!UPPER) return w < nb ? w : -1;
        const int r = (nb - 1) - (((nb - 1) - w) % G + G) % G;
        return r >= 0 ? r : -1;
    };
    int next_solve = first_own();
    T dinv[16], mnear[16];
    if (next_solve >= 0) block_load<T>(Dinv + (size_t)next_solve * NB * NB, NB, NB, NB, tid, dinv);
    {
        const int rn = nearest(d0);
        if (rn >= 0)
            block_load<T>(R + (int64_t)rn * NB * ld + d0 * NB, ld, min(NB, n - rn * NB), min(NB, n - d0 * NB), tid, mnear);
    }
    __syncthreads();

    for (int s = 0; s < nb; ++s) {
        const int d = UPPER ? nb - 1 - s : s;
        const int dn = min(NB, n - d * NB);   // rows (= columns) of diagonal block d
        const int owner = d % G;
        if (w == owner) {
            // b_d is complete (every earlier stage has been applied to it): x_d = inv(D_dd) * b_d
            const int jd = (d - w) / G;
            for (int e = tid; e < NB * TV_NR; e += TV_THREADS) xs[e] = bacc[jd][e];
            __syncthreads();
            T y[TV_NR];
            block_gemv<T>(dinv, xs, part, tid, y);
            if (q == 0) {
#pragma unroll
                for (int c = 0; c < TV_NR; ++c) {
                    tv_store(rx, (unsigned)((d * NB + i) * TV_NR + c) * 16u, tag, y[c]);   // to the other workgroups
                    xs[i * TV_NR + c] = y[c];
                    if (i < dn && c < nrhs) X[(int64_t)(d * NB + i) * ldx + c] = y[c];        // the result
                }
            }
            __syncthreads();
            next_solve = UPPER ? d - G : d + G;
            if (next_solve >= 0 && next_solve < nb) block_load<T>(Dinv + (size_t)next_solve * NB * NB, NB, NB, NB, tid, dinv);
        } else {
            bool timed_out = false;
            for (int e = tid; e < NB * TV_NR; e += TV_THREADS) {
                T v = T(0);
                int spins = 0;
                for (;;) {
                    asm volatile("" ::: "memory");   // plain buffer intrinsics: keep the load inside the loop
                    if (tv_load(rx, (unsigned)(d * NB * TV_NR + e) * 16u, tag, v)) break;
                    if (++spins > TV_SPIN_LIMIT) { timed_out = true; break; }
                }
                xs[e] = v;
            }
            if (timed_out) {
                s_dead = 1;
                __hip_atomic_store((unsigned long long*)(err + 1), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            if (s_dead) return;
        }
        // subtract T_rd * x_d from the owned blocks beyond d.  The nearest one first, from the block requested a stage ago;
        // then request the nearest block of the NEXT stage, so that its memory latency hides behind the next flag wait.
        const int rn = nearest(d);
        if (rn >= 0) {
            T y[TV_NR];
            block_gemv<T>(mnear, xs, part, tid, y);
            if (q == 0) {
                const int jr = (rn - w) / G;
#pragma unroll
                for (int c = 0; c < TV_NR; ++c) bacc[jr][i * TV_NR + c] -= y[c];
            }
        }
        const int dnext = UPPER ? d - 1 : d + 1;
        if (dnext >= 0 && dnext < nb) {
            const int rnn = nearest(dnext);
            if (rnn >= 0)
                block_load<T>(R + (int64_t)rnn * NB * ld + dnext * NB, ld, min(NB, n - rnn * NB), min(NB, n - dnext * NB), tid,
                              mnear);
        }
        if (rn >= 0) {
            for (int r = UPPER ? rn - G : rn + G; r >= 0 && r < nb; r += UPPER ? -G : G) {
                T m[16], y[TV_NR];
                block_load<T>(R + (int64_t)r * NB * ld + d * NB, ld, min(NB, n - r * NB), dn, tid, m);
                block_gemv<T>(m, xs, part, tid, y);
                if (q == 0) {
                    const int jr = (r - w) / G;
#pragma unroll
                    for (int c = 0; c < TV_NR; ++c) bacc[jr][i * TV_NR + c] -= y[c];
                }
            }
        }
        __syncthreads();  // this stage's updates are in bacc before the next stage reads it
    }
}

// B <- U^-1 L^-1 B for nrhs <= TV_NR per pass (row-major factors R, row-major B); interchanges already applied to B.
template <typename T>
int launch_trsv_coop(Handle* h, int64_t n, int64_t nrhs, const T* R, int64_t ld, T* B, int64_t ldb)
{
    if (n <= 0 || nrhs <= 0) return RFLU_OK;
    const int64_t nb = (n + NB - 1) / NB;
    // workspace: inverted diagonal blocks of L and of U, then the exchange area (one 16-byte granule per value of x)
    const size_t inv_bytes = (size_t)nb * NB * NB * sizeof(T);
    const size_t xchg_bytes = (size_t)nb * NB * TV_NR * 16;
    const size_t need = 2 * inv_bytes + xchg_bytes;
    const bool fresh = need > h->linv_tmp_bytes;
    RFLU_TRY(ensure_buffer(&h->linv_tmp, &h->linv_tmp_bytes, need));
    T* Linv = static_cast<T*>(h->linv_tmp);
    T* Uinv = reinterpret_cast<T*>(static_cast<char*>(h->linv_tmp) + inv_bytes);
    void* xchg = static_cast<char*>(h->linv_tmp) + 2 * inv_bytes;
    // tags: a fresh (or re-purposed) area is wiped once; afterwards every launch uses its own tag from the handle's counter
    if (fresh || h->trsv_tag > 0xfffffff0u || h->trsv_area != xchg) {
        RFLU_HIP(hipMemsetAsync(h->linv_tmp, 0, need, h->stream));
        h->trsv_tag = 0;
        h->trsv_area = xchg;
    }
    RFLU_TRY(launch_diag_inv<T>(h, n, R, ld, Linv));
    {
        ProfScope ps(h, RFLU_K_TRSM, (double)n * NB * NB / 3.0);
        hipLaunchKernelGGL(triu_inv_kernel<T>, dim3((unsigned)nb), dim3(64), 0, h->stream, (int)n, R, ld, Uinv);
        RFLU_HIP(hipGetLastError());
    }
    if (nb > (int64_t)TV_MAX_WGS * TV_KOWN) {
        set_error("launch_trsv_coop: %lld rows exceed %d", (long long)n, NB * TV_MAX_WGS * TV_KOWN);
        return RFLU_ERR_ARG;
    }
    const unsigned grid = (unsigned)std::min<int64_t>(nb, TV_MAX_WGS);
    {   // the stages wait for each other's results: every workgroup of a launch must be resident (asked once per handle)
        if (h->trsv_max_wgs == 0) {
            int a = 0, b = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, reinterpret_cast<const void*>(&trsv_coop_kernel<T, false>), TV_THREADS, 0) != hipSuccess ||
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, reinterpret_cast<const void*>(&trsv_coop_kernel<T, true>), TV_THREADS, 0) != hipSuccess) {
                (void)hipGetLastError();
                a = b = 0;
            }
            h->trsv_max_wgs = std::max(1, std::min(a, b) * h->num_cus);
        }
        if ((int)grid > h->trsv_max_wgs) {
            set_error("launch_trsv_coop: %u cooperating workgroups, but the device holds %d at a time", grid, h->trsv_max_wgs);
            return RFLU_ERR_ARG;
        }
    }
    for (int64_t c0 = 0; c0 < nrhs; c0 += TV_NR) {
        const int nr = (int)std::min<int64_t>(TV_NR, nrhs - c0);
        ProfScope ps(h, RFLU_K_TRSM, 2.0 * (double)n * (double)n * (double)nr, sizeof(T) * (double)n * (double)n);
        hipLaunchKernelGGL((trsv_coop_kernel<T, false>), dim3(grid), dim3(TV_THREADS), 0, h->stream, (int)n, nr, R, ld, Linv,
                           B + c0, ldb, xchg, (unsigned)xchg_bytes, ++h->trsv_tag, h->info_dev);
        hipLaunchKernelGGL((trsv_coop_kernel<T, true>), dim3(grid), dim3(TV_THREADS), 0, h->stream, (int)n, nr, R, ld, Uinv,
                           B + c0, ldb, xchg, (unsigned)xchg_bytes, ++h->trsv_tag, h->info_dev);
        RFLU_HIP(hipGetLastError());
    }
    return RFLU_OK;
}

template int launch_trsv_coop<double>(Handle*, int64_t, int64_t, const double*, int64_t, double*, int64_t);
template int launch_trsv_coop<float>(Handle*, int64_t, int64_t, const float*, int64_t, float*, int64_t);

}  // namespace rflu
