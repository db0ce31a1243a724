// laswp.hip -- row interchanges, the internal layout change, and small utilities.
//
// apply_permutation! (/root/reference/src/lu.jl:164-188) swaps row i with row P[i] sequentially for every pivot of a
// block.  Here a chunk of up to 64 sequential interchanges has already been folded (panel.hip: perm_build_wave) into an
// equivalent list of at most 128 independent row MOVES  new[dst[e]] = old[src[e]], so the kernel is a pure gather /
// scatter of contiguous row segments in the row-major R layout: a wave owns a strip of columns (one 128-byte line per row),
// issues all row reads of a chunk before the first write, and every pivot costs exactly its algorithmic 4*sizeof(T) bytes
// per column of HBM traffic -- no cache-line amplification.
// Roofline: HBM (8 TB/s spec, ~6.3 TB/s achievable); algorithmic bytes per launch = 4*sizeof(T)*ncols*pivots.
#include <stdint.h>
#include "rflu_internal.hpp"
#include "trsm_row.hpp"
#include "laswp_strip.hpp"

namespace rflu {

// Geometry (round 3).  A row segment is 8 lanes x 16 bytes = one 128-byte line (16 Float64 / 32 Float32 columns); a wave holds 8
// row slots, so the <= 128 moves of a chunk are 16 independent 16-byte loads per lane, ALL in flight before the first store
// (16 KB per wave).  Every WAVE owns its column strip for all chunks of the launch: nothing but that wave touches the
// strip (its stores are drained between chunks), so there is no workgroup barrier anywhere (round 2: two __syncthreads per chunk, 8-byte
// accesses, and 64 KB of static LDS in every workgroup -- 2 workgroups per CU -- for the one workgroup that inverts a diagonal
// block).  The next chunk's move list is requested before the current chunk's rows are stored.  VW = 1 is the same kernel with
// one element per lane for column ranges that are not 16-byte aligned.
constexpr int LW_WAVES = 4;                   // independent waves per workgroup

// inv_nb > 0: extra workgroups (the last inv_cnt) invert the leaves' 64x64 diagonal blocks for the fused TRSMs that follow
// (trsm.hip) -- they ride along with the leaf's interchange launch instead of costing a dependent launch of their own; such a
// launch asks for 2 * NB * NB elements of dynamic LDS, every other launch for none.
// A third column range [c2, c2+ncolsC) receives only the chunks after the first: for a pair leaf (panel.hip) these are
// leaf A's own columns, which still need leaf B's interchanges.
template <typename T, int VW, int LW_LPR>
__global__ void __launch_bounds__(64 * LW_WAVES) laswp_kernel(T* __restrict__ R, int64_t ld, int64_t c0, int64_t ncolsA,
                                                              int64_t c1, int64_t ncolsB, int64_t c2, int64_t ncolsC,
                                                              const int* __restrict__ pm_cnt, const int* __restrict__ pm_dst,
                                                              const int* __restrict__ pm_src, int chunk0, int chunk1, int inv_nb,
                                                              int inv_cnt, const T* inv_L, T* inv_out, LaswpGate gate)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lw_smem[];
    if (gate.wait_flag) {   // folded stream gate (factor_leafwise): hold until another stream has passed `wait_val`
        if (threadIdx.x == 0) {
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(gate.wait_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gate.wait_val) {
                __builtin_amdgcn_s_sleep(2);
                if (wall_clock64() - t0 > 200000000LL) {
                    __hip_atomic_fetch_or((unsigned long long*)(gate.info + 1), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        __syncthreads();
    }
    constexpr int SC = LW_LPR * VW;
    const int64_t strips = (ncolsA + SC - 1) / SC + (ncolsB + SC - 1) / SC + (ncolsC + SC - 1) / SC;
    const int64_t swap_blocks = (strips + LW_WAVES - 1) / LW_WAVES;
    if ((int64_t)blockIdx.x >= swap_blocks) {
        const int64_t i = (int64_t)blockIdx.x - swap_blocks;
        if (inv_nb > 0 && i < inv_cnt) {   // workgroup-uniform
            T* sL = reinterpret_cast<T*>(lw_smem);
            diag_inv_block16<T>(inv_nb, inv_L + i * (NB * ld + NB), ld, inv_out + i * NB * NB, sL, sL + NB * NB, threadIdx.x);
        }
    } else {
        const int64_t strip = (int64_t)blockIdx.x * LW_WAVES + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        if (strip < strips)
            laswp_strip<T, VW, LW_LPR>(R, ld, c0, ncolsA, c1, ncolsB, c2, ncolsC, pm_cnt, pm_dst, pm_src, chunk0, chunk1, strip);
    }
    if (gate.signal_flag) {   // the last workgroup to get here publishes `signal_val` (the counter wraps back to zero)
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            if (atomicInc(gate.signal_cnt, gridDim.x - 1) == gridDim.x - 1)
                __hip_atomic_store(gate.signal_flag, gate.signal_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Apply chunks [chunk0, chunk1) to the column ranges [c0, c0+ncolsA) and [c1, c1+ncolsB), and chunks [chunk0+1, chunk1) to
// [c2, c2+ncolsC); optionally invert inv_cnt consecutive inv_nb x inv_nb unit lower diagonal blocks starting at inv_L
// (leading dimension ld) into inv_out in the same launch.
template <typename T>
int launch_laswp3(Handle* h, T* R, int64_t ld, int64_t c0, int64_t ncolsA, int64_t c1, int64_t ncolsB, int64_t c2,
                  int64_t ncolsC, int64_t chunk0, int64_t chunk1, int64_t inv_nb, int64_t inv_cnt, const T* inv_L,
                  T* inv_out, LaswpGate gate)
{
    if (ncolsA < 0) ncolsA = 0;
    if (ncolsB < 0) ncolsB = 0;
    if (ncolsC < 0 || chunk1 - chunk0 < 2) ncolsC = 0;
    const bool swaps = chunk1 > chunk0 && ncolsA + ncolsB + ncolsC > 0;
    if (inv_nb <= 0) inv_cnt = 0;
    if (!swaps && inv_cnt <= 0 && !gate.wait_flag && !gate.signal_flag) return RFLU_OK;
    if (!swaps) { ncolsA = ncolsB = ncolsC = 0; }
    // 16-byte accesses when every column range starts and ends on a 16-byte boundary (ld is a multiple of 16 elements)
    constexpr int VWF = 16 / (int)sizeof(T);
    const bool vec = (reinterpret_cast<uintptr_t>(R) % 16 == 0) && ld % VWF == 0 &&
                     (ncolsA == 0 || (c0 % VWF == 0 && ncolsA % VWF == 0)) && (ncolsB == 0 || (c1 % VWF == 0 && ncolsB % VWF == 0)) &&
                     (ncolsC == 0 || (c2 % VWF == 0 && ncolsC % VWF == 0));
    // lanes per row segment: 8 x 16 bytes = one 128-byte line (RFLU_LASWP_LPR=4: half-line segments, twice the waves -- measured
    // no better: 2.97 vs 3.16 TB/s on the wide launches of an N=16384 factorization)
    const int lpr = h->tune.laswp_lpr == 4 ? 4 : 8;
    const int64_t SC = lpr * (vec ? VWF : 1);
    const int64_t strips = (ncolsA + SC - 1) / SC + (ncolsB + SC - 1) / SC + (ncolsC + SC - 1) / SC;
    int64_t blocks = (strips + LW_WAVES - 1) / LW_WAVES + inv_cnt;
    if (blocks == 0) blocks = 1;   // gates only: one idle workgroup (beyond every range, inv_cnt == 0)
    const double moved = 4.0 * sizeof(T) * (double)NB * ((double)(ncolsA + ncolsB) * (double)(chunk1 - chunk0) +
                                                         (double)ncolsC * (double)(chunk1 - chunk0 - 1));
    ProfScope ps(h, moved >= 32.0 * 1024 * 1024 ? RFLU_K_LASWP_WIDE : RFLU_K_LASWP, moved);
    const size_t lds = inv_cnt > 0 ? 2 * (size_t)NB * NB * sizeof(T) : 0;
#define RFLU_LASWP_LAUNCH(VWX, LPRX)                                                                                          \
    hipLaunchKernelGGL((laswp_kernel<T, VWX, LPRX>), dim3((unsigned)blocks), dim3(64 * LW_WAVES), lds, h->stream, R, ld, c0,    \
                       ncolsA, c1, ncolsB, c2, ncolsC, h->pm_cnt, h->pm_dst, h->pm_src, (int)chunk0, (int)chunk1, (int)inv_nb, \
                       (int)inv_cnt, inv_L, inv_out, gate)
    if (vec && lpr == 4) RFLU_LASWP_LAUNCH(VWF, 4);
    else if (vec) RFLU_LASWP_LAUNCH(VWF, 8);
    else RFLU_LASWP_LAUNCH(1, 8);
#undef RFLU_LASWP_LAUNCH
    RFLU_HIP(hipGetLastError());
    return RFLU_OK;
}

template <typename T>
int launch_laswp2(Handle* h, T* R, int64_t ld, int64_t c0, int64_t ncolsA, int64_t c1, int64_t ncolsB, int64_t chunk0,
                  int64_t chunk1, int64_t inv_nb, const T* inv_L, T* inv_out, LaswpGate gate)
{
    return launch_laswp3<T>(h, R, ld, c0, ncolsA, c1, ncolsB, 0, 0, chunk0, chunk1, inv_nb, 1, inv_L, inv_out, gate);
}


// ---- leaf-wise schedule: everything the critical path does to the NEXT leaf's 64 columns behind a leaf, short of the K = 64 update,
// in ONE launch of ONE workgroup (round 4).  Before: {interchanges on those columns || inverse of the leaf's diagonal block} (10.7 us,
// the inverse the long pole), then a launch of its own for X = inv(L11) * B_top (6.3 us, most of it the round trip of the inverse
// through memory).  Here the workgroup reads the leaf's move list and ALL moved rows of the 64 columns into registers (the loads
// travel while the block is inverted), stores the displaced rows, keeps the 64 rows that end up on top in LDS, multiplies them by
// the inverse it has just built (same v_mfma sequence as trsm_inv64_kernel: bit-identical) and stores X.  The inverse still goes
// to memory for the side stream's solves.
#define RFLU_LA_IDX(i, j) ((i) * NB + ((((j) + (i))) & (NB - 1)))
template <typename T>
__global__ void __launch_bounds__(256) leaf_la_kernel(T* __restrict__ R, int64_t ld, int64_t la0, const int* __restrict__ pm_cnt,
                                                      const int* __restrict__ pm_dst, const int* __restrict__ pm_src, int chunk,
                                                      int c0, const T* inv_L, T* inv_out, LaswpGate gate)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char la_smem[];
    __shared__ int s_src[2 * NB], s_dst[2 * NB], s_top[NB], s_cnt;
    T* sL = reinterpret_cast<T*>(la_smem);
    T* sX = sL + NB * NB;
    T* sB = sX + NB * NB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the leaf's diagonal block and its move list are final since the leaf kernel ended: requested before the gate (which is about
    // the side stream's writes to LA) so that their latency, the gate's and each other's run side by side
    T lv[16];
    diag_inv_load16<T>(NB, inv_L, ld, tid, lv);
    int my_src = 0, my_dst = 0, my_cnt = 0;
    if (tid < 2 * NB) {
        my_src = pm_src[(size_t)chunk * 2 * NB + tid];
        my_dst = pm_dst[(size_t)chunk * 2 * NB + tid];
    }
    if (tid == 0) my_cnt = pm_cnt[chunk];
    if (gate.wait_flag) {   // folded stream gate, as in laswp_kernel
        if (tid == 0) {
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(gate.wait_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gate.wait_val) {
                __builtin_amdgcn_s_sleep(2);
                if (wall_clock64() - t0 > 200000000LL) {
                    __hip_atomic_fetch_or((unsigned long long*)(gate.info + 1), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        __syncthreads();
    }
    // ---- the move list: row s_src[k] -> row s_dst[k], k < cnt <= 128; which row ends up at top position p
    if (tid < 2 * NB) {
        s_src[tid] = my_src;
        s_dst[tid] = my_dst;
    }
    if (tid == 0) s_cnt = my_cnt;
    if (tid < NB) s_top[tid] = c0 + tid;
    __syncthreads();
    const int cnt = s_cnt;
    if (tid < cnt) {
        const int d = s_dst[tid];
        if (d >= c0 && d < c0 + NB) s_top[d - c0] = s_src[tid];   // destinations are distinct
    }
    __syncthreads();
    // ---- every row this launch will overwrite is read first: the moved rows (two threads per move, 32 columns each) and the top
    // rows that stay where they are (four threads per row, 16 columns each)
    constexpr int VW = 16 / (int)sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(VW)));
    const int k = tid >> 1, hf = tid & 1;
    const bool mv = k < cnt;
    vec_t rv[32 / VW];
    if (mv) {
        const T* sp = R + (int64_t)s_src[k] * ld + la0 + hf * 32;
#pragma unroll
        for (int v = 0; v < 32 / VW; ++v) rv[v] = *reinterpret_cast<const vec_t*>(sp + v * VW);
    }
    const int p = tid >> 2, q = tid & 3;
    const bool stay = s_top[p] == c0 + p;
    vec_t tv[16 / VW];
    if (stay) {
        const T* sp = R + (int64_t)(c0 + p) * ld + la0 + q * 16;
#pragma unroll
        for (int v = 0; v < 16 / VW; ++v) tv[v] = *reinterpret_cast<const vec_t*>(sp + v * VW);
    }
    // ---- the inverse of the leaf's diagonal block (-> sX, rotated image, and inv_out) while those rows travel
    diag_inv_block16_pre<T>(lv, inv_out, sL, sX, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // every row has been read by whoever needs it: now they may be overwritten
    if (mv) {
        const int d = s_dst[k];
        if (d >= c0 + NB) {
            T* dp = R + (int64_t)d * ld + la0 + hf * 32;
#pragma unroll
            for (int v = 0; v < 32 / VW; ++v) *reinterpret_cast<vec_t*>(dp + v * VW) = rv[v];
        } else {
#pragma unroll
            for (int v = 0; v < 32 / VW; ++v)
#pragma unroll
                for (int e = 0; e < VW; ++e) sB[RFLU_LA_IDX(d - c0, hf * 32 + v * VW + e)] = rv[v][e];
        }
    }
    if (stay) {
#pragma unroll
        for (int v = 0; v < 16 / VW; ++v)
#pragma unroll
            for (int e = 0; e < VW; ++e) sB[RFLU_LA_IDX(p, q * 16 + v * VW + e)] = tv[v][e];
    }
    __syncthreads();
    // ---- X = inv(L11) * B_top: wave w -> rows 16w .. 16w+15, four column tiles (the operand order of trsm_inv64_kernel)
    {
        typedef typename InvMfma<T>::acc_t acc_t;
        const int fi = lane & 15, fk = lane >> 4;
        acc_t x[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) x[t] = acc_t{T(0), T(0), T(0), T(0)};
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const T a = sX[RFLU_LA_IDX(wave * 16 + fi, kk * 4 + fk)];
#pragma unroll
            for (int t = 0; t < 4; ++t) x[t] = InvMfma<T>::run(a, sB[RFLU_LA_IDX(kk * 4 + fk, t * 16 + fi)], x[t]);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                R[(int64_t)(c0 + wave * 16 + InvMfma<T>::crow(lane, r)) * ld + la0 + t * 16 + fi] = x[t][r];
    }
    if (gate.signal_flag) {
        __syncthreads();
        if (tid == 0) {
            __threadfence();
            if (atomicInc(gate.signal_cnt, gridDim.x - 1) == gridDim.x - 1)
                __hip_atomic_store(gate.signal_flag, gate.signal_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
#undef RFLU_LA_IDX

template <typename T>
int launch_leaf_la(Handle* h, T* R, int64_t ld, int64_t la0, int64_t chunk, int64_t c0, const T* inv_L, T* inv_out, LaswpGate gate)
{
    const size_t lds = 3 * (size_t)NB * NB * sizeof(T);
    bool& attr_set = h->la_attr_set[sizeof(T) == 8 ? 0 : 1];
    if (!attr_set) {
        RFLU_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&leaf_la_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    ProfScope ps(h, RFLU_K_LASWP, 4.0 * sizeof(T) * (double)NB * (double)NB);
    hipLaunchKernelGGL(leaf_la_kernel<T>, dim3(1), dim3(256), lds, h->stream, R, ld, la0, h->pm_cnt, h->pm_dst, h->pm_src, (int)chunk, (int)c0,
                       inv_L, inv_out, gate);
    RFLU_HIP(hipGetLastError());
    return RFLU_OK;
}
template int launch_leaf_la<double>(Handle*, double*, int64_t, int64_t, int64_t, int64_t, const double*, double*, LaswpGate);
template int launch_leaf_la<float>(Handle*, float*, int64_t, int64_t, int64_t, int64_t, const float*, float*, LaswpGate);

template <typename T>
int launch_laswp(Handle* h, T* R, int64_t ld, int64_t c0, int64_t ncols, int64_t chunk0, int64_t chunk1)
{
    return launch_laswp2<T>(h, R, ld, c0, ncols, 0, 0, chunk0, chunk1, 0, nullptr, nullptr, LaswpGate{});
}

template int launch_laswp<double>(Handle*, double*, int64_t, int64_t, int64_t, int64_t, int64_t);
template int launch_laswp<float>(Handle*, float*, int64_t, int64_t, int64_t, int64_t, int64_t);
template int launch_laswp3<double>(Handle*, double*, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, const double*, double*, LaswpGate);
template int launch_laswp3<float>(Handle*, float*, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, const float*, float*, LaswpGate);
template int launch_laswp2<double>(Handle*, double*, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, const double*, double*, LaswpGate);
template int launch_laswp2<float>(Handle*, float*, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, const float*, float*, LaswpGate);

// ---- tiled transpose: out[r][c] = in[c][r]; "rows_out x cols_out" is the shape of `out` seen as row-major ------------
// Used for column-major <-> R layout: a column-major m x n matrix (lda) IS a row-major n x m matrix (ld = lda).
template <typename T>
__global__ void __launch_bounds__(256) transpose_kernel(int64_t rows_out, int64_t cols_out, const T* __restrict__ in,
                                                        int64_t ld_in, T* __restrict__ out, int64_t ld_out)
{
    __shared__ T tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    // diagonal block order: with a power-of-two leading dimension the tiles of one block row start 2^k bytes apart and camp on
    // the same HBM channels; shifting the tile row by the tile column spreads the workgroups in flight over all of them
    const int64_t r0 = (int64_t)((blockIdx.y + blockIdx.x) % gridDim.y) * 64, c0 = (int64_t)blockIdx.x * 64;
    // read in[c0 + i][r0 + tx] (coalesced along in's rows)
    for (int i = ty; i < 64; i += 4) {
        const int64_t ir = c0 + i, ic = r0 + tx;
        tile[i][tx] = (ir < cols_out && ic < rows_out) ? in[ir * ld_in + ic] : T(0);
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int64_t orow = r0 + i, ocol = c0 + tx;
        if (orow < rows_out && ocol < cols_out) out[orow * ld_out + ocol] = tile[tx][i];
    }
}

template <typename T>
int launch_transpose(Handle* h, int64_t rows_out, int64_t cols_out, const T* in, int64_t ld_in, T* out, int64_t ld_out)
{
    if (rows_out <= 0 || cols_out <= 0) return RFLU_OK;
    ProfScope ps(h, RFLU_K_TRANSPOSE, 2.0 * sizeof(T) * (double)rows_out * (double)cols_out);
    dim3 grid((unsigned)((cols_out + 63) / 64), (unsigned)((rows_out + 63) / 64));
    hipLaunchKernelGGL(transpose_kernel<T>, grid, dim3(256), 0, h->stream, rows_out, cols_out, in, ld_in, out, ld_out);
    RFLU_HIP(hipGetLastError());
    return RFLU_OK;
}
template <typename T>
int launch_transpose_on(hipStream_t st, int64_t rows_out, int64_t cols_out, const T* in, int64_t ld_in, T* out, int64_t ld_out)
{
    if (rows_out <= 0 || cols_out <= 0) return RFLU_OK;
    dim3 grid((unsigned)((cols_out + 63) / 64), (unsigned)((rows_out + 63) / 64));
    hipLaunchKernelGGL(transpose_kernel<T>, grid, dim3(256), 0, st, rows_out, cols_out, in, ld_in, out, ld_out);
    RFLU_HIP(hipGetLastError());
    return RFLU_OK;
}
template int launch_transpose_on<double>(hipStream_t, int64_t, int64_t, const double*, int64_t, double*, int64_t);
template int launch_transpose_on<float>(hipStream_t, int64_t, int64_t, const float*, int64_t, float*, int64_t);
template int launch_transpose<double>(Handle*, int64_t, int64_t, const double*, int64_t, double*, int64_t);
template int launch_transpose<float>(Handle*, int64_t, int64_t, const float*, int64_t, float*, int64_t);

// ---- synthetic input: counter-based uniform [0,1), bit-identical to oracle/rflu_oracle.c:rfo_uniform01 -----------------
__device__ __forceinline__ double uniform01(uint64_t seed, uint64_t ctr)
{
    uint64_t z = seed + (ctr + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

template <typename T>
__global__ void __launch_bounds__(256) fill_uniform_kernel(T* A, int64_t m, int64_t n, int64_t ld, int row_major,
                                                           uint64_t seed, int64_t M_global, int64_t i0, int64_t j0,
                                                           double diag_add)
{
    // fast index runs along the contiguous dimension of the destination
    const int64_t fast = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t slow = blockIdx.y;
    const int64_t i = row_major ? slow : fast;
    const int64_t j = row_major ? fast : slow;
    if (i >= m || j >= n) return;
    const int64_t gi = i0 + i, gj = j0 + j;
    double v = uniform01(seed, (uint64_t)(gj * M_global + gi));
    T x = (T)v;
    if (gi == gj) x = (T)(x + (T)diag_add);
    A[row_major ? (i * ld + j) : (i + j * ld)] = x;
}

template <typename T>
int launch_fill_uniform(Handle* h, T* A, int64_t m, int64_t n, int64_t ld, int row_major, uint64_t seed,
                        int64_t M_global, int64_t i0, int64_t j0, double diag_add)
{
    if (m <= 0 || n <= 0) return RFLU_OK;
    const int64_t fast = row_major ? n : m, slow = row_major ? m : n;
    if (slow > 65535 * 1024LL) { set_error("fill: dimension too large"); return RFLU_ERR_ARG; }
    ProfScope ps(h, RFLU_K_MISC, 0.0);
    // gridDim.y is limited to 65535: loop over slabs of the slow dimension
    for (int64_t s0 = 0; s0 < slow; s0 += 65535) {
        const int64_t sn = (slow - s0 < 65535) ? (slow - s0) : 65535;
        dim3 grid((unsigned)((fast + 255) / 256), (unsigned)sn);
        T* base = A + (row_major ? s0 * ld : s0 * ld);
        hipLaunchKernelGGL(fill_uniform_kernel<T>, grid, dim3(256), 0, h->stream, base, row_major ? sn : m,
                           row_major ? n : sn, ld, row_major, seed, M_global, row_major ? i0 + s0 : i0,
                           row_major ? j0 : j0 + s0, diag_add);
    }
    RFLU_HIP(hipGetLastError());
    return RFLU_OK;
}
template int launch_fill_uniform<double>(Handle*, double*, int64_t, int64_t, int64_t, int, uint64_t, int64_t, int64_t,
                                         int64_t, double);
template int launch_fill_uniform<float>(Handle*, float*, int64_t, int64_t, int64_t, int, uint64_t, int64_t, int64_t,
                                        int64_t, double);

__global__ void iota_kernel(int64_t* ipiv, int64_t k0, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) ipiv[k0 + i] = k0 + i + 1;
}

// identity pivots ipiv[k0 .. k0+n) = k0+1 .. k0+n  (NoPivot with a caller-supplied vector, src/lu.jl:111-113)
int launch_iota_ipiv(Handle* h, int64_t* ipiv, int64_t k0, int64_t n)
{
    if (n <= 0 || ipiv == nullptr) return RFLU_OK;
    hipLaunchKernelGGL(iota_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, ipiv, k0, n);
    RFLU_HIP(hipGetLastError());
    return RFLU_OK;
}


// ---- device-side stream gates ------------------------------------------------------------------------------------------
// The leaf-wise schedule (driver.cpp: factor_leafwise) hands work between streams once per 64-column leaf.  A hipEvent record
// + hipStreamWaitEvent pair costs the waiting AND the recording stream 40-50 us of pipeline bubble each (measured with
// scripts/trace_timeline.sh) -- more than the launches the schedule removes -- so the per-leaf edges are ordinary one-wave
// kernels on monotonically increasing 64-bit counters: gate_signal publishes `value` when the stream reaches it (everything
// before it in the stream has completed and released its writes), gate_wait holds its stream until a counter reaches `value`.
// A waiter that sees no progress for ~2 s raises the panel timeout flag (info[1] bit 0) and lets its stream go on.
// ---- do two streams share a hardware pipe? -----------------------------------------------------------------------------------
// HSA queues are spread round-robin over the 4 pipes of the compute micro-engine, and a pipe serves one of its queues at a time: while
// the head packet of one queue waits for its predecessor (every kernel of an in-order stream does), the other queues of that pipe are
// not looked at.  Two BUSY streams on one pipe therefore take turns kernel by kernel (N=4096: 12 -> 20 ms when the update or the
// side stream shares the critical path's pipe; which streams collide depends on how many queues the process created before).
// The probe: stream a runs {spin for `ticks`, mark}, stream b runs {mark}, enqueued in that order.  On different pipes b's mark starts
// while a is still spinning; on the same pipe it starts after a's spin has ended.
__global__ void qprobe_spin_kernel(long long* out, long long ticks)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    *out = (long long)wall_clock64();
}
__global__ void qprobe_mark_kernel(long long* out) { *out = (long long)wall_clock64(); }

// slots: device memory for 3 stamps; returns 1 (shared pipe), 0 (independent) in *shared
int queue_probe(hipStream_t a, hipStream_t b, long long* slots, long long* host3, int* shared)
{
    RFLU_HIP(hipStreamSynchronize(a));
    RFLU_HIP(hipStreamSynchronize(b));
    hipLaunchKernelGGL(qprobe_mark_kernel, dim3(1), dim3(1), 0, a, slots + 1);   // both queues exist and are warm
    hipLaunchKernelGGL(qprobe_mark_kernel, dim3(1), dim3(1), 0, b, slots + 2);
    RFLU_HIP(hipStreamSynchronize(a));
    RFLU_HIP(hipStreamSynchronize(b));
    hipLaunchKernelGGL(qprobe_spin_kernel, dim3(1), dim3(1), 0, a, slots + 0, 30000LL);   // 300 us at 100 MHz
    hipLaunchKernelGGL(qprobe_mark_kernel, dim3(1), dim3(1), 0, a, slots + 1);
    hipLaunchKernelGGL(qprobe_mark_kernel, dim3(1), dim3(1), 0, b, slots + 2);
    RFLU_HIP(hipGetLastError());
    RFLU_HIP(hipStreamSynchronize(a));
    RFLU_HIP(hipStreamSynchronize(b));
    RFLU_HIP(hipMemcpy(host3, slots, 3 * sizeof(long long), hipMemcpyDeviceToHost));
    *shared = host3[2] >= host3[0] ? 1 : 0;
    return RFLU_OK;
}

// Throughput form (the one validate_queues uses): `n` one-thread kernels on each of the two streams, enqueued alternately.  The
// penalty of a shared pipe sets in after a few tens of kernels, so only the second half is timed:
// *us_per_kernel = the slower stream's (last stamp - stamp of kernel n/2) / (n - 1 - n/2).  slots: 8 device words.
int queue_probe_rate(hipStream_t a, hipStream_t b, int n, long long* slots, double* us_per_kernel)
{
    RFLU_HIP(hipStreamSynchronize(a));
    RFLU_HIP(hipStreamSynchronize(b));
    const int mid = n / 2;
    // both queues are held back until everything is enqueued: what is timed is the rate at which two BACKLOGGED queues drain (with
    // the host enqueueing at about the rate the GPU consumes, a shared pipe went unnoticed in one run of fourteen)
    hipLaunchKernelGGL(qprobe_spin_kernel, dim3(1), dim3(1), 0, a, slots + 6, (long long)n * 800);   // 8 us per kernel pair, 100 MHz ticks
    hipLaunchKernelGGL(qprobe_spin_kernel, dim3(1), dim3(1), 0, b, slots + 7, (long long)n * 800);
    for (int i = 0; i < n; ++i) {
        hipLaunchKernelGGL(qprobe_mark_kernel, dim3(1), dim3(1), 0, a, slots + (i < mid ? 4 : (i == mid ? 0 : 1)));
        hipLaunchKernelGGL(qprobe_mark_kernel, dim3(1), dim3(1), 0, b, slots + (i < mid ? 5 : (i == mid ? 2 : 3)));
    }
    RFLU_HIP(hipGetLastError());
    RFLU_HIP(hipStreamSynchronize(a));
    RFLU_HIP(hipStreamSynchronize(b));
    long long hs[4];
    RFLU_HIP(hipMemcpy(hs, slots, sizeof(hs), hipMemcpyDeviceToHost));
    const double d = double(n - 1 - mid);
    const double ta = double(hs[1] - hs[0]) / 100.0 / d, tb = double(hs[3] - hs[2]) / 100.0 / d;
    *us_per_kernel = ta > tb ? ta : tb;
    return RFLU_OK;
}

__global__ void gate_signal_kernel(unsigned long long* flag, unsigned long long value, long long* stamp)
{
    __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (stamp) *stamp = (long long)wall_clock64();   // measurement builds only (RFLU_GATE_TRACE, scripts/gate_trace.py)
}

__global__ void gate_wait_kernel(const unsigned long long* flag, unsigned long long value, int64_t* info)
{
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < value) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > 200000000LL) {   // 100 MHz wall clock: 2 s
            __hip_atomic_fetch_or((unsigned long long*)(info + 1), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
}

int launch_gate_signal(Handle* h, unsigned long long* flag, unsigned long long value, long long* stamp)
{
    hipLaunchKernelGGL(gate_signal_kernel, dim3(1), dim3(1), 0, h->stream, flag, value, stamp);
    RFLU_HIP(hipGetLastError());
    return RFLU_OK;
}

int launch_gate_signal_on(hipStream_t st, unsigned long long* flag, unsigned long long value)
{
    hipLaunchKernelGGL(gate_signal_kernel, dim3(1), dim3(1), 0, st, flag, value, (long long*)nullptr);
    RFLU_HIP(hipGetLastError());
    return RFLU_OK;
}

int launch_gate_wait(Handle* h, const unsigned long long* flag, unsigned long long value)
{
    hipLaunchKernelGGL(gate_wait_kernel, dim3(1), dim3(1), 0, h->stream, flag, value, h->info_dev);
    RFLU_HIP(hipGetLastError());
    return RFLU_OK;
}

}  // namespace rflu
