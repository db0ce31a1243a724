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
#include <cstdlib>
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
template <typename T, int NR>
__device__ __forceinline__ void block_gemv(const T (&m)[16], const T* xs, T* part, int tid, T (&y)[NR])
{
    const int i = tid & 63, q = tid >> 6;
    T acc[NR];
#pragma unroll
    for (int c = 0; c < NR; ++c) acc[c] = T(0);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
#pragma unroll
        for (int c = 0; c < NR; ++c) acc[c] += m[k] * xs[(q * 16 + k) * NR + c];
    }
#pragma unroll
    for (int c = 0; c < NR; ++c) part[(q * NB + i) * NR + c] = acc[c];
    __syncthreads();
    if (q == 0) {
#pragma unroll
        for (int c = 0; c < NR; ++c)
            y[c] = (part[i * NR + c] + part[(NB + i) * NR + c]) + (part[(2 * NB + i) * NR + c] + part[(3 * NB + i) * NR + c]);
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
// raw granule now, tag check later (a fetch issued a stage ahead, trsm_chain_kernel)
__device__ __forceinline__ tv_u4 tv_load_raw(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, TV_AUX_SC1); }
__device__ __forceinline__ bool tv_decode(const tv_u4& x, unsigned tag, double& v)
{
    v = __longlong_as_double((long long)(((unsigned long long)x[0] << 32) | (unsigned long long)x[2]));
    return x[1] == tag && x[3] == tag;
}
__device__ __forceinline__ bool tv_decode(const tv_u4& x, unsigned tag, float& v)
{
    v = __uint_as_float(x[0]);
    return x[1] == tag && x[3] == tag;
}
// The same through the caches: for the many workgroups that fetch a block everybody fetches (the first one of an XCD brings the line into
// that XCD's L2, the others hit it there).  A line that was cached before its granule was written shows the old tag: the caller falls
// back to tv_load.
__device__ __forceinline__ bool tv_load_cached(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned tag, double& v)
{
    const tv_u4 x = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    v = __longlong_as_double((long long)(((unsigned long long)x[0] << 32) | (unsigned long long)x[2]));
    return x[1] == tag && x[3] == tag;
}
__device__ __forceinline__ bool tv_load_cached(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned tag, float& v)
{
    const tv_u4 x = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    v = __uint_as_float(x[0]);
    return x[1] == tag && x[3] == tag;
}

// M_r = inv(D_rr) * T_{r,p}  for the block p solved right before r (p = r-1 in the lower, r+1 in the upper triangle): with it
// the solve of block r needs ONE 64x64 product once x_p is known (x_r = y_r - M_r x_p, y_r = inv(D_rr) (b_r - everything
// before p), which its owner has had a whole stage to prepare) instead of two dependent ones.  One workgroup per block.
template <typename T, bool UPPER>
__global__ void __launch_bounds__(256) trsv_sub_kernel(int n, const T* __restrict__ R, int64_t ld, const T* __restrict__ Dinv,
                                                       T* __restrict__ Msub)
{
    __shared__ T sT[NB * (NB + 1)];
    __shared__ T sD[NB * (NB + 1)];
    const int nb = (n + NB - 1) / NB;
    const int r = blockIdx.x, p = UPPER ? r + 1 : r - 1;
    const int tid = threadIdx.x, i = tid & 63, q = tid >> 6;
    T* out = Msub + (size_t)r * NB * NB;
    if (p < 0 || p >= nb) {
        for (int k = 0; k < 16; ++k) out[i * NB + q * 16 + k] = T(0);
        return;
    }
    const int rows_ok = min(NB, n - r * NB), cols_ok = min(NB, n - p * NB);
    for (int e = tid; e < NB * NB; e += 256) {
        const int a = e >> 6, c = e & 63;
        sT[a * (NB + 1) + c] = (a < rows_ok && c < cols_ok) ? R[(int64_t)(r * NB + a) * ld + p * NB + c] : T(0);
        sD[a * (NB + 1) + c] = Dinv[(size_t)r * NB * NB + e];
    }
    __syncthreads();
    T acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = T(0);
    for (int k = 0; k < NB; ++k) {
        const T dk = sD[i * (NB + 1) + k];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] += dk * sT[k * (NB + 1) + q * 16 + j];
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) out[i * NB + q * 16 + k] = acc[k];
}

// One triangle, NR right-hand sides.  Block r of the right-hand sides belongs to workgroup r mod G; blocks are solved in
// order (0, 1, ... in the lower, nb-1, nb-2, ... in the upper triangle), stage d = "x_d is known".  The chain per stage is
//   owner of the NEXT block:  wait for x_d (one hop)  ->  x_next = y_next - M_next x_d (one 64x64 product)  ->  publish;
// everything else trails: every workgroup subtracts T_rd x_d from the blocks r it owns further on (each 64x64 block of the
// triangle is read exactly once, the nearest one from registers filled a stage ago), and the owner of the block after next
// prepares y = inv(D) b for it as soon as x_d has been applied to it.  Workgroups whose turn is more than two stages away
// poll at leisure (with 256 of them polling flat out the hop of the one that matters takes twice as long).
template <typename T, bool UPPER, int NR>
__global__ void __launch_bounds__(TV_THREADS) trsv_chain_kernel(int n, int nrhs, const T* __restrict__ R, int64_t ld,
                                                                const T* __restrict__ Dinv, const T* __restrict__ Msub, T* X, int64_t ldx,
                                                                void* xchg, unsigned xchg_bytes, unsigned tag, int64_t* err)
{
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(xchg, 0, xchg_bytes, 0x00020000);
    __shared__ T xs[NB * NR];                 // x_d of the current stage
    __shared__ T xnext[NB * NR];              // x of the block this workgroup has just solved (next stage's x_d)
    __shared__ T yv[NB * NR];                 // y of the block this workgroup solves next
    __shared__ T part[4 * NB * NR];
    __shared__ T bacc[TV_KOWN][NB * NR];      // the owned blocks of B: they stay here until they become x
    __shared__ int s_dead;
    const int tid = threadIdx.x, i = tid & 63, q = tid >> 6;
    const int nb = (n + NB - 1) / NB;
    const int G = gridDim.x, w = blockIdx.x;
    const int dir = UPPER ? -1 : 1;
    if (tid == 0) s_dead = 0;
    for (int j = 0; j < TV_KOWN; ++j) {
        const int r = w + j * G;
        for (int e = tid; e < NB * NR; e += TV_THREADS) {
            const int k = e / NR, c = e % NR;
            bacc[j][e] = (r < nb && r * NB + k < n && c < nrhs) ? X[(int64_t)(r * NB + k) * ldx + c] : T(0);
        }
    }
    const int d0 = UPPER ? nb - 1 : 0;
    auto owned_beyond = [&](int d) -> int {   // nearest owned block strictly beyond block d in solve order (or -1)
        if (!UPPER) {
            if (d + 1 >= nb) return -1;
            const int r = d + 1 + ((w - (d + 1)) % G + G) % G;
            return r < nb ? r : -1;
        }
        if (d - 1 < 0) return -1;
        const int r = d - 1 - (((d - 1) - w) % G + G) % G;
        return r >= 0 ? r : -1;
    };
    auto slot = [&](int r) -> int { return (r - w) / G; };
    auto load_T = [&](int r, int d, T (&m)[16]) {
        block_load<T>(R + (int64_t)r * NB * ld + d * NB, ld, min(NB, n - r * NB), min(NB, n - d * NB), tid, m);
    };
    // publish the block just solved (y in the q == 0 threads) as x_r: granules for the other workgroups, xnext for this one, X
    auto publish = [&](int r, const T (&y)[NR]) {
        if (q == 0) {
            const int rn = min(NB, n - r * NB);
#pragma unroll
            for (int c = 0; c < NR; ++c) {
                tv_store(rx, (unsigned)((r * NB + i) * NR + c) * 16u, tag, y[c]);
                xnext[i * NR + c] = y[c];
                if (i < rn && c < nrhs) X[(int64_t)(r * NB + i) * ldx + c] = y[c];
            }
        }
    };
    int ns = UPPER ? ((nb - 1) - (((nb - 1) - w) % G + G) % G) : w;   // the block this workgroup solves next (-1: none left)
    if (ns < 0 || ns >= nb) ns = -1;
    T dinv[16], mcrit[16], mnear[16];
    if (ns >= 0) {
        block_load<T>(Dinv + (size_t)ns * NB * NB, NB, NB, NB, tid, dinv);
        block_load<T>(Msub + (size_t)ns * NB * NB, NB, NB, NB, tid, mcrit);
    }
    __syncthreads();
    // blocks without two predecessors: the first one is solved outright, the second one has its y before the first stage
    if (ns == d0 || ns == d0 + dir) {
        T y[NR];
        block_gemv<T, NR>(dinv, bacc[slot(ns)], part, tid, y);
        if (ns == d0) {
            publish(ns, y);
            ns = ns + dir * G;
            if (ns < 0 || ns >= nb) ns = -1;
            if (ns >= 0) {
                block_load<T>(Dinv + (size_t)ns * NB * NB, NB, NB, NB, tid, dinv);
                block_load<T>(Msub + (size_t)ns * NB * NB, NB, NB, NB, tid, mcrit);
            }
        } else if (q == 0) {
#pragma unroll
            for (int c = 0; c < NR; ++c) yv[i * NR + c] = y[c];
        }
    }
    {   // the off-diagonal block of the first stage's nearest trailing block (not the block solved next: that one uses mcrit)
        int rn = owned_beyond(d0);
        if (rn == d0 + dir) rn = owned_beyond(rn);
        if (rn >= 0) load_T(rn, d0, mnear);
    }
    __syncthreads();

    for (int s = 0; s + 1 < nb; ++s) {   // the last block's x has nobody to go to
        const int d = d0 + dir * s, dnext = d + dir;
        // ---- x_d
        if (w == d % G) {
            for (int e = tid; e < NB * NR; e += TV_THREADS) xs[e] = xnext[e];
        } else {
            const bool urgent = ns >= 0 && (ns == dnext || ns == dnext + dir);
            bool timed_out = false;
            for (int e = tid; e < NB * NR; e += TV_THREADS) {
                T v = T(0);
                int spins = 0;
                for (;;) {
                    asm volatile("" ::: "memory");   // plain buffer intrinsics: keep the load inside the loop
                    if (tv_load(rx, (unsigned)(d * NB * NR + e) * 16u, tag, v)) break;
                    if (!urgent) __builtin_amdgcn_s_sleep(8);
                    if (++spins > TV_SPIN_LIMIT) { timed_out = true; break; }
                }
                xs[e] = v;
            }
            if (timed_out) {
                s_dead = 1;
                __hip_atomic_store((unsigned long long*)(err + 1), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
        if (s_dead) return;
        // ---- the chain: the next block is this workgroup's
        if (ns == dnext) {
            T y[NR];
            block_gemv<T, NR>(mcrit, xs, part, tid, y);
            if (q == 0) {
#pragma unroll
                for (int c = 0; c < NR; ++c) y[c] = yv[i * NR + c] - y[c];
            }
            publish(ns, y);
            ns = ns + dir * G;
            if (ns < 0 || ns >= nb) ns = -1;
            if (ns >= 0) {
                block_load<T>(Dinv + (size_t)ns * NB * NB, NB, NB, NB, tid, dinv);
                block_load<T>(Msub + (size_t)ns * NB * NB, NB, NB, NB, tid, mcrit);
            }
        }
        // ---- T_rd x_d off the blocks owned further on: the nearest first (registers), then the request for the next stage's
        int rn = owned_beyond(d);
        if (rn == dnext) rn = owned_beyond(rn);
        if (rn >= 0) {
            T y[NR];
            block_gemv<T, NR>(mnear, xs, part, tid, y);
            if (q == 0) {
#pragma unroll
                for (int c = 0; c < NR; ++c) bacc[slot(rn)][i * NR + c] -= y[c];
            }
        }
        if (s + 2 < nb) {
            int rnn = owned_beyond(dnext);
            if (rnn == dnext + dir) rnn = owned_beyond(rnn);
            if (rnn >= 0) load_T(rnn, dnext, mnear);
        }
        // ---- the block after next is this workgroup's: everything but x_next has reached it, its y can be prepared
        if (rn >= 0 && rn == dnext + dir && rn == ns) {
            __syncthreads();
            T y[NR];
            block_gemv<T, NR>(dinv, bacc[slot(ns)], part, tid, y);
            if (q == 0) {
#pragma unroll
                for (int c = 0; c < NR; ++c) yv[i * NR + c] = y[c];
            }
        }
        if (rn >= 0) {
            for (int r = rn + dir * G; r >= 0 && r < nb; r += dir * G) {
                T m[16], y[NR];
                load_T(r, d, m);
                block_gemv<T, NR>(m, xs, part, tid, y);
                if (q == 0) {
#pragma unroll
                    for (int c = 0; c < NR; ++c) bacc[slot(r)][i * NR + c] -= y[c];
                }
            }
        }
        __syncthreads();  // this stage's updates are in bacc / yv / xnext before the next stage reads them
    }
}

// =====================================================================================================================
// The same solve for a BLOCK of right-hand sides (33 .. a few hundred; round 5): 64 columns per pass, the block products on the
// MFMA units.  With the recursive splitting (driver.cpp: trsm_rec / triu_solve_rec) 64 right-hand sides at n = 16384 were ~1000
// dependent launches of almost no work: 28.8 ms, fifteen times the one-right-hand-side solve, for the same 2 GiB of factor bytes.
// Here the chain of trsv_chain_kernel carries 64 columns at once:
//   * block r of the right-hand sides (64 x 64) belongs to workgroup r mod G and stays in B (global memory; this workgroup is the
//     only one that touches it until it is solved);
//   * stage d: x_d arrives as tagged granules (or from this workgroup's own LDS copy), every workgroup subtracts T_rd * x_d from the
//     blocks it owns further on -- 64 MFMAs per wave, the A operand straight from the factor (each 64 x 64 block of the triangle is
//     read exactly once), the B operand x_d from LDS, the C tile read from / written to B;
//   * the chain per stage is ONE product: the owner of the next block holds y = inv(D) (b - everything but the last stage),
//     prepared while x_d was still on its way, and M = inv(D) T (trsv_sub_kernel): x_next = y - M x_d, published at once.
// Wave w owns rows [16w, 16w+16) of a 64-row block x 64 columns = 4 accumulator fragments.
// Measured (n = 16384, 64 right-hand sides, stamps of the owner of the next block): 9.7 us per stage = 2.1 us for the product (64
// v_mfma_f64_16x16x4 of 64 clocks each per wave) + ~5 us from the previous owner's first granule store to x_d in this workgroup's LDS
// (64 KB of granules through the memory side) + the owner's own previous stage (its block's update and y: two more products).
constexpr int TC_NR = 64;                 // right-hand sides per pass
// Round 5, second step: the columns of a pass are independent, so a pass runs as `chains` chains of NRC = 16 columns side by side
// (blockIdx / G = chain): a stage then moves 16 KB of granules and one MFMA fragment per wave instead of 64 KB and four, and the factor
// is read once per chain (4 GiB instead of 1 per triangle at n = 16384: still under the stage time).  n = 16384, 64 right-hand sides:
// 5.9 -> see DESIGN.md section 7.
constexpr int tc_xld(int nrc) { return nrc + 1; }   // LDS row pitch of x_d ([k][column]): odd -- the four k groups of a fragment read (rows 16 apart) land 16 doubles apart in the banks

template <typename T>
struct TcMfma;
template <>
struct TcMfma<double> {
    typedef double acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t run(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int crow(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <>
struct TcMfma<float> {
    typedef float acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int crow(int lane, int r) { return 4 * (lane >> 4) + r; }
};

// A fragments of a 64 x 64 block M (row-major, leading dimension ldm; rows >= rows_ok / columns >= cols_ok read as zero):
// lane (fi = lane & 15, fk = lane >> 4) of wave w holds M[16w + fi][16 fk + kk], kk = 0..15 -- 128 contiguous bytes of its row (Float64),
// so a whole block arrives as eight 16-byte loads per lane.  (Which k a (lane, kk) slot of the MFMA stands for is free as long as
// the B operand agrees: tc_mma reads x row 16 fk + kk for it.)
template <typename T>
__device__ __forceinline__ void tc_load_a(const T* __restrict__ M, int64_t ldm, int rows_ok, int cols_ok, int wave, int lane, T (&a)[16], bool neg)
{
    constexpr int VW = 16 / (int)sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(VW)));
    const int fi = lane & 15, fk = lane >> 4;
    const int row = wave * 16 + fi;
    const T* Mp = M + (int64_t)row * ldm + 16 * fk;
    if (rows_ok >= NB && cols_ok >= NB && ((reinterpret_cast<uintptr_t>(M) | (uintptr_t)(ldm * (int64_t)sizeof(T))) & 15) == 0) {   // (wave-uniform)
#pragma unroll
        for (int v = 0; v < 16 / VW; ++v) {
            const vec_t x = *reinterpret_cast<const vec_t*>(Mp + v * VW);
#pragma unroll
            for (int e = 0; e < VW; ++e) a[v * VW + e] = neg ? -x[e] : x[e];
        }
        return;
    }
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
        const T v = (row < rows_ok && 16 * fk + kk < cols_ok) ? Mp[kk] : T(0);
        a[kk] = neg ? -v : v;
    }
}

// acc += A * xs   (xs: [64][XLD] in LDS)
template <typename T, int FR, int XLD>
__device__ __forceinline__ void tc_mma(const T (&a)[16], const T* xs, int lane, typename TcMfma<T>::acc_t (&acc)[FR])
{
    const int fi = lane & 15, fk = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
        const T* xr = xs + (16 * fk + kk) * XLD + fi;
#pragma unroll
        for (int t = 0; t < FR; ++t) acc[t] = TcMfma<T>::run(a[kk], xr[16 * t], acc[t]);
    }
}

#ifdef RFLU_TC_TRACE
__device__ long long tc_trace_pub[2][1024];      // [triangle][block]: wall clock when the block's x was published (chain 0)
__device__ long long tc_trace_wg[2][1024][6];
__device__ long long tc_trace_own[2][1024][4];   // [triangle][block]: its owner at the top of the stage that solves it, x_d landed, product done, loop end of that stage
#define TC_STAMP_OWN(r, i) do { if (chain == 0 && tid == 0 && (r) >= 0 && (r) < 1024) tc_trace_own[UPPER ? 1 : 0][r][i] = wall_clock64(); } while (0)    // [triangle][stage][point]: one bystander workgroup (w = 7 of chain 0)
#define TC_STAMP_PUB(r) do { if (chain == 0 && tid == 0 && (r) < 1024) tc_trace_pub[UPPER ? 1 : 0][r] = wall_clock64(); } while (0)
#define TC_STAMP_WG(s, i) do { if (chain == 0 && w == 7 && tid == 0 && (s) < 1024) tc_trace_wg[UPPER ? 1 : 0][s][i] = wall_clock64(); } while (0)
#else
#define TC_STAMP_PUB(r) do { } while (0)
#define TC_STAMP_WG(s, i) do { } while (0)
#define TC_STAMP_OWN(r, i) do { } while (0)
#endif
// Workgroup barrier that waits for LDS traffic only.  __syncthreads() also waits for every outstanding global access of the wave: the
// acknowledgement of the write-through granule / result stores just issued (1-2 us) and the loads issued AHEAD on purpose (the next
// stage's -T block, the next x) -- which is what made a stage cost 9-10 us whatever it moved.  Everything the barriers of
// trsm_chain_kernel order goes through LDS; global data is either private to a thread (a C tile: the same thread reads what it wrote)
// or carries tags.
__device__ __forceinline__ void tc_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <typename T, bool UPPER, int NRC, int RUN>
__global__ void __launch_bounds__(TV_THREADS) trsm_chain_kernel(int n, int nrhs_all, const T* __restrict__ R, int64_t ld,
                                                                const T* __restrict__ Dinv, const T* __restrict__ Msub, T* X, int64_t ldx,
                                                                void* xchg_all, unsigned xchg_bytes, unsigned tag, int64_t* err, int chains, int cached_fetch)
{
    typedef typename TcMfma<T>::acc_t acc_t;
    constexpr int FR = NRC / 16;          // accumulator fragments per wave (16 rows x NRC columns)
    constexpr int TC_XLD = tc_xld(NRC);
    constexpr int GP = NRC / 4;           // granules of x_d a thread fetches (row tid >> 2)
    const int G = (int)gridDim.x / chains, w = (int)blockIdx.x % G, chain = (int)blockIdx.x / G;
    X += chain * NRC;
    const int nrhs = min(NRC, nrhs_all - chain * NRC);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(static_cast<char*>(xchg_all) + (size_t)chain * xchg_bytes, 0, xchg_bytes, 0x00020000);
    __shared__ T xs[2][NB * TC_XLD];   // x_d by parity of the stage
    __shared__ T ys[NB * TC_XLD];      // the block two stages ahead, staged as a B operand for y = inv(D) * b
    __shared__ int s_dead;
    __shared__ int s_ok[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 15;
    const int nb = (n + NB - 1) / NB;
    const int dir = UPPER ? -1 : 1;
    const int d0 = UPPER ? nb - 1 : 0;
    if (tid == 0) s_dead = 0;
    auto rows_of = [&](int r) { return min(NB, n - r * NB); };
    // C-layout access to block r of X: row 16w + crow(lane, q), column 16t + fi
    auto load_c = [&](int r, acc_t (&c)[FR]) {
        const int rn = rows_of(r);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = wave * 16 + TcMfma<T>::crow(lane, q);
#pragma unroll
            for (int t = 0; t < FR; ++t) {
                const int col = 16 * t + fi;
                c[t][q] = (row < rn && col < nrhs) ? X[(int64_t)(r * NB + row) * ldx + col] : T(0);
            }
        }
    };
    auto store_c = [&](int r, const acc_t (&c)[FR]) {
        const int rn = rows_of(r);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = wave * 16 + TcMfma<T>::crow(lane, q);
#pragma unroll
            for (int t = 0; t < FR; ++t) {
                const int col = 16 * t + fi;
                if (row < rn && col < nrhs) X[(int64_t)(r * NB + row) * ldx + col] = c[t][q];
            }
        }
    };
    auto stage_c = [&](T* dst, const acc_t (&c)[FR]) {   // C layout -> [row][TC_XLD] in LDS
        // (one ds_write per element, kept apart by compiler barriers: hipcc 7.2 paired the Float32 writes of this loop into ds_write2_b32
        //  with a wrong first offset -- column 4 instead of 16 for the second fragment -- and a quarter of the staged block kept its old
        //  contents.  Not `volatile`: a volatile LDS write is preceded by s_waitcnt vmcnt(0), i.e. by a wait for the acknowledgement of
        //  every store and the arrival of every load issued ahead -- 2 us, four times per stage)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = wave * 16 + TcMfma<T>::crow(lane, q);
#pragma unroll
            for (int t = 0; t < FR; ++t) {
                dst[row * TC_XLD + 16 * t + fi] = c[t][q];
                asm volatile("" ::: "memory");
            }
        }
    };
    // x_r = c: granules for the other workgroups, the final values into X, and this workgroup's own copy for the stage that uses it
    auto publish = [&](int r, const acc_t (&c)[FR], T* own) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = wave * 16 + TcMfma<T>::crow(lane, q);
#pragma unroll
            for (int t = 0; t < FR; ++t)
                tv_store(rx, (unsigned)((r * NB + row) * NRC + 16 * t + fi) * 16u, tag, c[t][q]);
        }
        stage_c(own, c);
        store_c(r, c);
    };
    // Ownership in RUNS: blocks q = RUN k .. RUN k + RUN - 1 (q counts in solve order) belong to workgroup k mod G.  Inside a run the next
    // x needs no hop -- the owner has x_d in its LDS when it publishes it -- so a chain of nb stages pays nb / RUN hops (a hop: 5-8 us from
    // the publish to the block in the next owner's LDS, against 0.5-2 us per product).
    auto qof = [&](int r) { return UPPER ? nb - 1 - r : r; };
    auto rof = [&](int q) { return UPPER ? nb - 1 - q : q; };
    auto owner_of = [&](int r) { return (qof(r) / RUN) % G; };
    // next owned block at or beyond block r in solve order (-1: none)
    auto owned_from = [&](int r) -> int {
        if (r < 0 || r >= nb) return -1;
        const int q = qof(r), run = q / RUN;
        const int delta = ((w - run) % G + G) % G;
        const int q2 = delta == 0 ? q : (run + delta) * RUN;
        return q2 < nb ? rof(q2) : -1;
    };
    int ns = owned_from(d0);             // the block this workgroup solves next
    // the blocks of the run ns belongs to stay in registers from the moment the run becomes current to their solution (every earlier
    // x_d is subtracted there); runs further on are updated in X
    acc_t racc[RUN][FR];
    int cur_first = -1;                  // first block (solve order) of the run in registers
    auto run_index = [&](int r) -> int { // position of block r in the current run, -1: not in it
        if (cur_first < 0 || r < 0) return -1;
        const int k = qof(r) - qof(cur_first);
        return (k >= 0 && k < RUN && owner_of(r) == w) ? k : -1;
    };
    auto load_run = [&]() {
        cur_first = ns;
#pragma unroll
        for (int k = 0; k < RUN; ++k) {
#pragma unroll
            for (int t = 0; t < FR; ++t) racc[k][t] = acc_t{T(0), T(0), T(0), T(0)};
            const int r = ns < 0 ? -1 : ns + dir * k;
            if (r >= 0 && r < nb && qof(r) / RUN == qof(ns) / RUN) load_c(r, racc[k]);
        }
    };
    acc_t yv[FR];                        // y of block ns (valid from the stage before it is solved)
#pragma unroll
    for (int t = 0; t < FR; ++t) yv[t] = acc_t{T(0), T(0), T(0), T(0)};
    // A fragments of -M and of inv(D) for block ns and (RUN == 2) for the block behind it in its run: requested when a run becomes
    // current, a stage or more ahead of their use
    T mcrit[16], dcrit[16], mcritB[16], dcritB[16];
    auto load_crit = [&]() {
        if (ns >= 0) {
            tc_load_a<T>(Msub + (size_t)ns * NB * NB, NB, NB, NB, wave, lane, mcrit, true);
            tc_load_a<T>(Dinv + (size_t)ns * NB * NB, NB, NB, NB, wave, lane, dcrit, false);
            if constexpr (RUN == 2) {
                const int r2 = ns + dir;
                if (r2 >= 0 && r2 < nb && qof(r2) / RUN == qof(ns) / RUN) {
                    tc_load_a<T>(Msub + (size_t)r2 * NB * NB, NB, NB, NB, wave, lane, mcritB, true);
                    tc_load_a<T>(Dinv + (size_t)r2 * NB * NB, NB, NB, NB, wave, lane, dcritB, false);
                }
            }
        }
    };
    // ns has been solved: on to the next block of the run (its fragments are here) or to the next run (requested now)
    auto advance = [&]() {
        const int prev = ns;
        ns = owned_from(ns + dir);
        if (RUN == 2 && ns >= 0 && qof(ns) / RUN == qof(prev) / RUN) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { mcrit[i] = mcritB[i]; dcrit[i] = dcritB[i]; }
        } else {
            load_crit();
            load_run();
        }
    };
    load_crit();
    load_run();
    tc_barrier();
    // y = inv(D) * c for a block held in registers (C layout), through the LDS staging
    auto make_y = [&](const acc_t (&c)[FR], const T (&dv)[16], acc_t (&y)[FR]) {
        stage_c(ys, c);
        tc_barrier();
#pragma unroll
        for (int t = 0; t < FR; ++t) y[t] = acc_t{T(0), T(0), T(0), T(0)};
        tc_mma<T, FR, TC_XLD>(dv, ys, lane, y);
        tc_barrier();   // ys may be rewritten
    };
    // prologue: the first block is solved outright, the second one has its y before the first stage
    if (ns == d0) {
        acc_t x0[FR];
        make_y(racc[0], dcrit, x0);
        publish(d0, x0, xs[0]);
        advance();
    }
    if (ns == d0 + dir) {
        const int k = run_index(ns);   // (workgroup-uniform)
#pragma unroll
        for (int kk = 0; kk < RUN; ++kk)
            if (k == kk) make_y(racc[kk], dcrit, yv);
    }
    // -T of the blocks of the current run for the stage that comes next, requested a stage ahead: tnear[k] holds -T(tn_blk[k], tn_col)
    T tnear[RUN][16];
    int tn_blk[RUN], tn_col = -1;
    auto request_t = [&](int col, int beyond_q) {   // for the current run's blocks behind position beyond_q (solve order)
        tn_col = col;
#pragma unroll
        for (int k = 0; k < RUN; ++k) {
            const int r = cur_first < 0 ? -1 : cur_first + dir * k;
            tn_blk[k] = -1;
            if (r >= 0 && r < nb && qof(r) / RUN == qof(cur_first) / RUN && qof(r) > beyond_q) {
                tc_load_a<T>(R + (int64_t)r * NB * ld + col * NB, ld, rows_of(r), min(NB, n - col * NB), wave, lane, tnear[k], true);
                tn_blk[k] = r;
            }
        }
    };
    request_t(d0, qof(d0 + dir));
    tc_barrier();

    // Every workgroup handles every stage, and a stage handled strictly in sequence costs {wait, fetch latency, LDS, barrier, product(s),
    // barrier}: that sum, not the hop to the next owner, set the pace (runs of two blocks changed nothing by themselves).  So the fetch
    // of x_(d+1) is ISSUED during stage d, right after x_d has landed, and looked at when stage d + 1 begins: a workgroup that is behind
    // the front finds the block complete and never waits for memory.  (16-column chains only: the raw granules wait in registers.)
    constexpr bool PREFETCH = NRC <= 16;
    tv_u4 praw[PREFETCH ? GP : 1];
    bool pre_valid = false;
    for (int s = 0; s + 1 < nb; ++s) {   // the last block's x has nobody to go to
        const int d = d0 + dir * s, dnext = d + dir;
        T* xd = xs[s & 1];
        if (owned_from(dnext) < 0) break;   // nothing left for this workgroup (workgroup-uniform)
        TC_STAMP_WG(s, 0);
        if (ns == dnext) TC_STAMP_OWN(ns, 0);
        // ---- x_d
        if (w != owner_of(d)) {
            const bool urgent = ns == dnext || ns == dnext + dir;
            // thread -> row k = tid >> 2, columns (tid & 3) * GP + j: all GP granules requested, the missing ones again
            const int k = tid >> 2, cb = (tid & 3) * GP;
            T v[GP];
            unsigned miss = (1u << GP) - 1u;
            int spins = 0;
            bool timed_out = false;
            const unsigned goff = (unsigned)((d * NB + k) * NRC + cb) * 16u;
            if constexpr (PREFETCH) {
                if (pre_valid) {
                    miss = 0;
#pragma unroll
                    for (int j = 0; j < GP; ++j) miss |= tv_decode(praw[j], tag, v[j]) ? 0u : (1u << j);
                }
            }
            // (workgroup-uniform decision: the waiting below has a barrier in it)
            bool all_here = false;
            if (PREFETCH && pre_valid) {
                const bool wave_ok = __ballot(miss == 0) == ~0ull;
                if (lane == 0) s_ok[wave] = wave_ok ? 1 : 0;
                tc_barrier();
                all_here = (s_ok[0] & s_ok[1] & s_ok[2] & s_ok[3]) != 0;
            }
            if (!urgent && !all_here) {
                // a workgroup whose turn is not near waits for ONE granule of x_d with one lane, at leisure, and fetches the block
                // when that one is there: 250 workgroups sweeping 64 KB per poll round took the fabric from the two that matter
                // (10 us per stage instead of 5)
                if (tid == 0) {
                    T probe;
                    while (!tv_load(rx, (unsigned)((d * NB + NB - 1) * NRC + NRC - 1) * 16u, tag, probe)) {
                        asm volatile("" ::: "memory");
                        __builtin_amdgcn_s_sleep(32);
                        if (++spins > TV_SPIN_LIMIT / 8) break;   // (the sweep below times out properly)
                    }
                }
                spins = 0;
                tc_barrier();
                if (cached_fetch == 2) {   // timing experiment (wrong results): a bystander fetches one granule per thread instead of GP
                    (void)tv_load(rx, goff, tag, v[0]);
#pragma unroll
                    for (int j = 1; j < GP; ++j) v[j] = v[0];
                    miss = 0;
                } else if (cached_fetch) {
                    // x_d is (almost certainly) complete in memory: fetch it through the L2; what a cached read misses is fetched again below
                    bool okj[GP];
#pragma unroll
                    for (int j = 0; j < GP; ++j) okj[j] = tv_load_cached(rx, goff + (unsigned)j * 16u, tag, v[j]);
                    miss = 0;
#pragma unroll
                    for (int j = 0; j < GP; ++j) miss |= okj[j] ? 0u : (1u << j);
                }
            }
            while (miss) {
                asm volatile("" ::: "memory");
                // all requests in flight before the first tag is looked at (a test between two loads makes them dependent round trips:
                // 12 us per stage instead of 5)
                bool okj[GP];
#pragma unroll
                for (int j = 0; j < GP; ++j) okj[j] = tv_load(rx, goff + (unsigned)j * 16u, tag, v[j]);
                miss = 0;
#pragma unroll
                for (int j = 0; j < GP; ++j) miss |= okj[j] ? 0u : (1u << j);
                if (miss) {
                    if (!urgent) __builtin_amdgcn_s_sleep(8);
                    if (++spins > TV_SPIN_LIMIT) { timed_out = true; break; }
                }
            }
            TC_STAMP_WG(s, 1);
#pragma unroll
            for (int j = 0; j < GP; ++j) {
                xd[k * TC_XLD + cb + j] = v[j];
                asm volatile("" ::: "memory");   // (see stage_c)
            }
            if (timed_out) {
                s_dead = 1;
                __hip_atomic_store((unsigned long long*)(err + 1), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        tc_barrier();
        if (s_dead) return;
        TC_STAMP_WG(s, 2);
        if (ns == dnext) TC_STAMP_OWN(ns, 1);
        if constexpr (PREFETCH) {
            pre_valid = s + 2 < nb && w != owner_of(dnext) && owned_from(dnext + dir) >= 0;
            if (pre_valid) {
                const unsigned goff = (unsigned)((dnext * NB + (tid >> 2)) * NRC + (tid & 3) * GP) * 16u;
#pragma unroll
                for (int j = 0; j < GP; ++j) praw[j] = tv_load_raw(rx, goff + (unsigned)j * 16u);
            }
        }
        // ---- the chain: the next block is this workgroup's.  x_next = y + (-M) x_d: the accumulators start from y
        if (ns == dnext) {
            acc_t mx[FR];
#pragma unroll
            for (int t = 0; t < FR; ++t) mx[t] = yv[t];
            tc_mma<T, FR, TC_XLD>(mcrit, xd, lane, mx);
            TC_STAMP_OWN(ns, 2);
            publish(ns, mx, xs[(s + 1) & 1]);
            TC_STAMP_PUB(ns);
            advance();
        }
        TC_STAMP_WG(s, 3);
        // ---- T_rd x_d off the blocks owned further on.  The current run (registers) in straight-line code: a loop here makes hipcc wait
        // for EVERY outstanding memory access at its header and at its exit (loads issued ahead, store acknowledgements: ~2 us each)
        int last_q = qof(dnext);
#pragma unroll
        for (int kk = 0; kk < RUN; ++kk) {
            const int r = cur_first < 0 ? -1 : cur_first + dir * kk;
            const bool mine = r >= 0 && r < nb && qof(r) / RUN == qof(cur_first) / RUN && qof(r) > qof(dnext);   // (workgroup-uniform)
            if (mine) {
                last_q = qof(r);
                if (!(tn_blk[kk] == r && tn_col == d))   // (a run that has just become current: its -T was not asked for)
                    tc_load_a<T>(R + (int64_t)r * NB * ld + d * NB, ld, rows_of(r), min(NB, n - d * NB), wave, lane, tnear[kk], true);
                tc_mma<T, FR, TC_XLD>(tnear[kk], xd, lane, racc[kk]);
                // the block after next is this workgroup's next: everything but x_next has reached it -- its y
                if (r == dnext + dir && r == ns) make_y(racc[kk], dcrit, yv);
            }
        }
        TC_STAMP_WG(s, 4);
        // the next stage's -T for the run in registers: on its way while this stage ends and the next x arrives
        if (s + 2 < nb) request_t(dnext, qof(dnext) + 1);
        // runs further on live in X (matrices of more than RUN * G blocks only)
        if (last_q + 1 < nb && owned_from(rof(last_q + 1)) >= 0) {
            for (int r = owned_from(rof(last_q + 1)); r >= 0; r = owned_from(r + dir)) {
                if (run_index(r) >= 0) continue;
                T tf[16];
                tc_load_a<T>(R + (int64_t)r * NB * ld + d * NB, ld, rows_of(r), min(NB, n - d * NB), wave, lane, tf, true);
                acc_t c[FR];
                load_c(r, c);
                tc_mma<T, FR, TC_XLD>(tf, xd, lane, c);
                store_c(r, c);
            }
        }
        tc_barrier();  // everybody is done with xs[s & 1] / ys before the stage after next rewrites them
        TC_STAMP_WG(s, 5);
    }
}

template <typename T, int NR>
static int launch_trsv_pass(Handle* h, unsigned grid, int64_t n, int nr, const T* R, int64_t ld, const T* Linv, const T* Uinv, const T* Lsub,
                            const T* Usub, T* B, int64_t ldb, void* xchg, size_t xchg_bytes)
{
    ProfScope ps(h, RFLU_K_TRSM, 2.0 * (double)n * (double)n * (double)nr, sizeof(T) * (double)n * (double)n);
    hipLaunchKernelGGL((trsv_chain_kernel<T, false, NR>), dim3(grid), dim3(TV_THREADS), 0, h->stream, (int)n, nr, R, ld, Linv, Lsub, B, ldb,
                       xchg, (unsigned)xchg_bytes, ++h->trsv_tag, h->info_dev);
    hipLaunchKernelGGL((trsv_chain_kernel<T, true, NR>), dim3(grid), dim3(TV_THREADS), 0, h->stream, (int)n, nr, R, ld, Uinv, Usub, B, ldb,
                       xchg, (unsigned)xchg_bytes, ++h->trsv_tag, h->info_dev);
    RFLU_HIP(hipGetLastError());
    return RFLU_OK;
}

// B <- U^-1 L^-1 B for nrhs <= TV_NR per pass (row-major factors R, row-major B); interchanges already applied to B.
// wide: passes of TC_NR = 64 right-hand sides on the MFMA units (trsm_chain_kernel) instead of TV_NR = 8 on the vector units.
template <typename T>
int launch_trsv_coop(Handle* h, int64_t n, int64_t nrhs, const T* R, int64_t ld, T* B, int64_t ldb, bool wide)
{
    if (n <= 0 || nrhs <= 0) return RFLU_OK;
    const int64_t nb = (n + NB - 1) / NB;
    // workspace: inverted diagonal blocks of L and of U, their products with the blocks next to the diagonal, then the exchange
    // area (one 16-byte granule per value of x)
    const size_t inv_bytes = (size_t)nb * NB * NB * sizeof(T);
    const size_t xchg_bytes = (size_t)nb * NB * (wide ? TC_NR : TV_NR) * 16;
    const size_t need = 4 * inv_bytes + xchg_bytes;
    const bool fresh = need > h->linv_tmp_bytes;
    RFLU_TRY(ensure_buffer(&h->linv_tmp, &h->linv_tmp_bytes, need));
    char* base = static_cast<char*>(h->linv_tmp);
    T* Linv = reinterpret_cast<T*>(base);
    T* Uinv = reinterpret_cast<T*>(base + inv_bytes);
    T* Lsub = reinterpret_cast<T*>(base + 2 * inv_bytes);
    T* Usub = reinterpret_cast<T*>(base + 3 * inv_bytes);
    void* xchg = base + 4 * inv_bytes;
    // tags: a fresh (or re-purposed) area is wiped once; afterwards every launch uses its own tag from the handle's counter
    if (fresh || h->trsv_tag > 0xfffffff0u || h->trsv_area != xchg) {
        RFLU_HIP(hipMemsetAsync(h->linv_tmp, 0, need, h->stream));
        h->trsv_tag = 0;
        h->trsv_area = xchg;
    }
    RFLU_TRY(launch_diag_inv<T>(h, n, R, ld, Linv));
    {
        ProfScope ps(h, RFLU_K_TRSM, (double)n * NB * NB / 3.0 + 4.0 * (double)n * NB * NB);
        hipLaunchKernelGGL(triu_inv_kernel<T>, dim3((unsigned)nb), dim3(64), 0, h->stream, (int)n, R, ld, Uinv);
        hipLaunchKernelGGL((trsv_sub_kernel<T, false>), dim3((unsigned)nb), dim3(256), 0, h->stream, (int)n, R, ld, Linv, Lsub);
        hipLaunchKernelGGL((trsv_sub_kernel<T, true>), dim3((unsigned)nb), dim3(256), 0, h->stream, (int)n, R, ld, Uinv, Usub);
        RFLU_HIP(hipGetLastError());
    }
    if (nb > (int64_t)TV_MAX_WGS * TV_KOWN) {
        set_error("launch_trsv_coop: %lld rows exceed %d", (long long)n, NB * TV_MAX_WGS * TV_KOWN);
        return RFLU_ERR_ARG;
    }
    const unsigned grid = (unsigned)std::min<int64_t>(nb, TV_MAX_WGS);
    {   // the stages wait for each other's results: every workgroup of a launch must be resident (asked once per handle)
        constexpr int TI = sizeof(T) == 8 ? 0 : 1;   // per element type: the Float64 kernels hold twice the LDS of the Float32 ones
        if (h->trsv_max_wgs[TI] == 0) {
            int a = 0, b = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, reinterpret_cast<const void*>(&trsv_chain_kernel<T, false, TV_NR>), TV_THREADS, 0) != hipSuccess ||
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, reinterpret_cast<const void*>(&trsv_chain_kernel<T, true, TV_NR>), TV_THREADS, 0) != hipSuccess) {
                (void)hipGetLastError();
                a = b = 0;
            }
            int c = 0, d = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&c, reinterpret_cast<const void*>(&trsm_chain_kernel<T, false, TC_NR, 1>), TV_THREADS, 0) != hipSuccess ||
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&d, reinterpret_cast<const void*>(&trsm_chain_kernel<T, true, TC_NR, 1>), TV_THREADS, 0) != hipSuccess) {
                (void)hipGetLastError();
                c = d = 0;
            }
            int e = 0, f = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&e, reinterpret_cast<const void*>(&trsm_chain_kernel<T, false, 16, 2>), TV_THREADS, 0) != hipSuccess ||
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&f, reinterpret_cast<const void*>(&trsm_chain_kernel<T, true, 16, 2>), TV_THREADS, 0) != hipSuccess) {
                (void)hipGetLastError();
                e = f = 0;
            }
            h->trsm16_per_cu[TI] = std::min(e, f);
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&e, reinterpret_cast<const void*>(&trsm_chain_kernel<T, false, 32, 1>), TV_THREADS, 0) != hipSuccess ||
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&f, reinterpret_cast<const void*>(&trsm_chain_kernel<T, true, 32, 1>), TV_THREADS, 0) != hipSuccess) {
                (void)hipGetLastError();
                e = f = 0;
            }
            h->trsm32_per_cu[TI] = std::min(e, f);
            h->trsv_max_wgs[TI] = std::max(1, std::min(std::min(a, b), std::min(c, d)) * h->num_cus);
        }
        if ((int)grid > h->trsv_max_wgs[TI]) {
            set_error("launch_trsv_coop: %u cooperating workgroups, but the device holds %d at a time", grid, h->trsv_max_wgs[TI]);
            return RFLU_ERR_ARG;
        }
    }
#ifdef RFLU_EXPERIMENTS
    // experiments build only (it returns RFLU_OK with B unsolved): timing of the preparation kernels / one triangle alone
    static const int dbg_skip = getenv("RFLU_DBG_SOLVE_SKIP") ? atoi(getenv("RFLU_DBG_SOLVE_SKIP")) : 0;   // 1 = no chain kernels, 2 = L only, 3 = U only
#else
    constexpr int dbg_skip = 0;
#endif
    if (wide && dbg_skip == 1) return RFLU_OK;
    if (wide) {
        // a pass of up to 64 columns as ONE chain of 64, TWO of 32 or FOUR of 16 columns side by side (trsm_chain_kernel); the narrower
        // chains need all their workgroups on the device together
        const int mode = h->tune.trsm_chain_split;   // 0: 64 x 1, 1: 16 x 4 in runs of two blocks, 2: 32 x 2
        for (int64_t c0 = 0; c0 < nrhs; c0 += TC_NR) {
            const int nr = (int)std::min<int64_t>(TC_NR, nrhs - c0);
            const int nrc = mode == 1 ? 16 : mode == 2 ? 32 : TC_NR;
            const int chains = (nr + nrc - 1) / nrc;
            const int run = mode == 1 ? 2 : 1;
            const int per_cu = mode == 1 ? h->trsm16_per_cu[sizeof(T) == 8 ? 0 : 1] : mode == 2 ? h->trsm32_per_cu[sizeof(T) == 8 ? 0 : 1] : 1;
            const int64_t want = (nb + run - 1) / run;
            const int g = (int)std::min<int64_t>(want, (int64_t)per_cu * h->num_cus / chains);
            const bool split = mode != 0 && per_cu > 0 && (g >= 32 || g == want);
            ProfScope ps(h, RFLU_K_TRSM, 2.0 * (double)n * (double)n * (double)nr, sizeof(T) * (double)n * (double)n * (split ? chains : 1));
            const unsigned per_chain = (unsigned)((size_t)nb * NB * nrc * 16);
            const dim3 gs((unsigned)(g * chains));
            if (split && mode == 1) {
                if (dbg_skip != 3) hipLaunchKernelGGL((trsm_chain_kernel<T, false, 16, 2>), gs, dim3(TV_THREADS), 0, h->stream, (int)n, nr, R, ld, Linv, Lsub, B + c0, ldb,
                                   xchg, per_chain, ++h->trsv_tag, h->info_dev, chains, h->tune.trsm_chain_cached);
                if (dbg_skip != 2) hipLaunchKernelGGL((trsm_chain_kernel<T, true, 16, 2>), gs, dim3(TV_THREADS), 0, h->stream, (int)n, nr, R, ld, Uinv, Usub, B + c0, ldb,
                                   xchg, per_chain, ++h->trsv_tag, h->info_dev, chains, h->tune.trsm_chain_cached);
            } else if (split && mode == 2) {
                if (dbg_skip != 3) hipLaunchKernelGGL((trsm_chain_kernel<T, false, 32, 1>), gs, dim3(TV_THREADS), 0, h->stream, (int)n, nr, R, ld, Linv, Lsub, B + c0, ldb,
                                   xchg, per_chain, ++h->trsv_tag, h->info_dev, chains, h->tune.trsm_chain_cached);
                if (dbg_skip != 2) hipLaunchKernelGGL((trsm_chain_kernel<T, true, 32, 1>), gs, dim3(TV_THREADS), 0, h->stream, (int)n, nr, R, ld, Uinv, Usub, B + c0, ldb,
                                   xchg, per_chain, ++h->trsv_tag, h->info_dev, chains, h->tune.trsm_chain_cached);
            } else {
                if (dbg_skip != 3) hipLaunchKernelGGL((trsm_chain_kernel<T, false, TC_NR, 1>), dim3(grid), dim3(TV_THREADS), 0, h->stream, (int)n, nr, R, ld, Linv, Lsub, B + c0, ldb,
                                   xchg, (unsigned)xchg_bytes, ++h->trsv_tag, h->info_dev, 1, h->tune.trsm_chain_cached);
                if (dbg_skip != 2) hipLaunchKernelGGL((trsm_chain_kernel<T, true, TC_NR, 1>), dim3(grid), dim3(TV_THREADS), 0, h->stream, (int)n, nr, R, ld, Uinv, Usub, B + c0, ldb,
                                   xchg, (unsigned)xchg_bytes, ++h->trsv_tag, h->info_dev, 1, h->tune.trsm_chain_cached);
            }
            RFLU_HIP(hipGetLastError());
        }
#ifdef RFLU_TC_TRACE
        if (sizeof(T) == 8) {
            static long long pub[2][1024], wg[2][1024][6], own[2][1024][4];
            RFLU_HIP(hipMemcpyFromSymbol(own, HIP_SYMBOL(tc_trace_own), sizeof(own)));
            RFLU_HIP(hipStreamSynchronize(h->stream));
            RFLU_HIP(hipMemcpyFromSymbol(pub, HIP_SYMBOL(tc_trace_pub), sizeof(pub)));
            RFLU_HIP(hipMemcpyFromSymbol(wg, HIP_SYMBOL(tc_trace_wg), sizeof(wg)));
            for (int tri = 0; tri < 2; ++tri) {
                fprintf(stderr, "[tc trace] triangle %d: publish-to-publish (us) for blocks in solve order 100..116:", tri);
                for (int q = 100; q < 116 && q + 1 < (int)nb; ++q) {
                    const int r0 = tri ? (int)nb - 1 - q : q, r1 = tri ? r0 - 1 : r0 + 1;
                    fprintf(stderr, " %.2f", (pub[tri][r1] - pub[tri][r0]) / 100.0);
                }
                fprintf(stderr, "\n   per block q: previous publish -> top of its stage / x landed / product done / published:");
                for (int q = 100; q < 108 && q + 1 < (int)nb; ++q) {
                    const int r = tri ? (int)nb - 1 - q : q, rp = tri ? r + 1 : r - 1;
                    fprintf(stderr, "  [%d] %.2f %.2f %.2f %.2f", q, (own[tri][r][0] - pub[tri][rp]) / 100.0, (own[tri][r][1] - pub[tri][rp]) / 100.0, (own[tri][r][2] - pub[tri][rp]) / 100.0, (pub[tri][r] - pub[tri][rp]) / 100.0);
                }
                const int qa = 40, qb = (int)std::min<int64_t>(nb, 1024) - 40;
                const int ra = tri ? (int)nb - 1 - qa : qa, rb = tri ? (int)nb - 1 - qb : qb;
                fprintf(stderr, "\n   average %.2f us per block over %d..%d\n", (pub[tri][rb] - pub[tri][ra]) / 100.0 / (qb - qa), qa, qb);
                double acc[5] = {0, 0, 0, 0, 0}, top = 0; int cnt = 0;
                for (int s = 40; s < qb; ++s) {
                    if (!wg[tri][s][0] || !wg[tri][s][5] || !wg[tri][s][1]) continue;
                    for (int i = 0; i < 5; ++i) acc[i] += (wg[tri][s][i + 1] - wg[tri][s][i]) / 100.0;
                    if (wg[tri][s + 1][0]) top += (wg[tri][s + 1][0] - wg[tri][s][5]) / 100.0;
                    ++cnt;
                }
                if (cnt) fprintf(stderr, "   bystander workgroup, %d stages: wait+fetch %.2f, LDS+barrier %.2f, prefetch issue+chain %.2f, owned loop %.2f, -T request+barrier %.2f, to next stage %.2f us\n",
                                 cnt, acc[0] / cnt, acc[1] / cnt, acc[2] / cnt, acc[3] / cnt, acc[4] / cnt, top / cnt);
            }
        }
#endif
        return RFLU_OK;
    }
    for (int64_t c0 = 0; c0 < nrhs; c0 += TV_NR) {
        const int nr = (int)std::min<int64_t>(TV_NR, nrhs - c0);
        if (nr == 1) RFLU_TRY((launch_trsv_pass<T, 1>(h, grid, n, nr, R, ld, Linv, Uinv, Lsub, Usub, B + c0, ldb, xchg, xchg_bytes)));
        else RFLU_TRY((launch_trsv_pass<T, TV_NR>(h, grid, n, nr, R, ld, Linv, Uinv, Lsub, Usub, B + c0, ldb, xchg, xchg_bytes)));
    }
    return RFLU_OK;
}

template int launch_trsv_coop<double>(Handle*, int64_t, int64_t, const double*, int64_t, double*, int64_t, bool);
template int launch_trsv_coop<float>(Handle*, int64_t, int64_t, const float*, int64_t, float*, int64_t, bool);

}  // namespace rflu
