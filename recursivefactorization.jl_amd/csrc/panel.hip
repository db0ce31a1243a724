// panel.hip -- leaf panel factorization: the unblocked pivoted LU of a tall m x w block (w <= 64).
//
// Replaces _generic_lufact! (/root/reference/src/lu.jl:290-338) at the leaves of the Toledo recursion
// (src/lu.jl:192-195).  Semantics kept: argmax |a_ik| over the not-yet-pivoted rows with strict '>' from 0
// (zero/NaN column -> the row already in position k; ties -> lowest row position, src/lu.jl:298-305), row interchange,
// reciprocal-multiply scaling (src/lu.jl:317-320), zero pivot -> record info once and keep updating (:321-334).
//
// MI355X design -- a latency problem, not a flop problem (w pivot steps, each a reduction over the whole column):
//   * G = ceil(rows/512/RT) workgroups of 512 threads, all co-resident, one thread per matrix row, the row's w <= 64 entries live in
//     REGISTERS for the whole kernel (fully unrolled column loop => static register indices).  The rank-1 updates
//     therefore cost ~(w-k) FMAs per thread per step and no memory traffic at all.
//   * rows never move: every thread tracks the current row POSITION of its row; an interchange just renames two
//     positions.  Rows are written back to their final positions at the end (coalesced through an LDS transpose).
//   * per step ONE cross-workgroup exchange: every workgroup publishes its candidate {position, row values k..w-1}
//     as 8-byte data-tagged granules ({epoch tag, 32 payload bits}, relaxed agent-scope stores = sc1 write-through);
//     wave 0 of every workgroup polls the G candidate headers, all arrive at the same winner, and read the winner's
//     row (already published speculatively) -- no grid barrier, no fences, one hop
//     (cdna_hip_programming.md Guideline 16 form R2; MI355X_MICROARCH.md "handoff-1to1").
//     Records are double-buffered by step parity: a workgroup can be at most one step ahead of the slowest reader.
//   * every spin is bounded; on timeout an error word is set and the host returns RFLU_ERR_TIMEOUT.
//   * the last wave of workgroup 0 turns the w interchanges into a list of (dst,src) row moves for laswp.hip.
// The NoPivot variant (Val(false)) needs no exchange at all: every workgroup factors the w x w top block redundantly in
// LDS and then solves its own rows against U11.
//
// Roofline: neither HBM nor MFMA -- the bound is w x (one cross-CU hop, ~1-2 us).  Algorithmic work reported to the
// timers: m*w^2 flops.
#include <algorithm>

#include "panel_common.hpp"
#include "trsm_row.hpp"

namespace rflu {

#ifndef RFLU_PANEL_F32_TU   // the non-template pieces live in the Float64 translation unit only
__global__ void __launch_bounds__(64) perm_build_kernel(const int64_t* ipiv, int k0, int k1, int* pm_cnt, int* pm_dst,
                                                        int* pm_src)
{
    __shared__ int s_piv[NB];
    __shared__ int s_rows[2 * NB];
    __shared__ int s_content[2 * NB];
    const int lane = threadIdx.x;
    const int chunk = k0 / NB + blockIdx.x;
    const int base = chunk * NB;
    const int w = min(NB, k1 - base);
    s_piv[lane] = (lane < w) ? (int)(ipiv[base + lane] - 1) : base + lane;
    __syncthreads();
    perm_build_wave(s_piv, base, w, lane, s_rows, s_content, pm_cnt + chunk, pm_dst + (size_t)chunk * 2 * NB,
                    pm_src + (size_t)chunk * 2 * NB);
}

int launch_perm_build(Handle* h, const int64_t* ipiv, int64_t k0, int64_t k1, int64_t m)
{
    (void)m;
    if (k1 <= k0) return RFLU_OK;
    const int nchunks = (int)((k1 - k0 + NB - 1) / NB);
    ProfScope ps(h, RFLU_K_MISC, 0.0);
    hipLaunchKernelGGL(perm_build_kernel, dim3(nchunks), dim3(64), 0, h->stream, ipiv, (int)k0, (int)k1, h->pm_cnt,
                       h->pm_dst, h->pm_src);
    RFLU_HIP(hipGetLastError());
    return RFLU_OK;
}
#endif  // RFLU_PANEL_F32_TU

// =====================================================================================================================
// Pivoted leaf panel
// =====================================================================================================================
// ---- all LDS state of the pivoted kernel in ONE object (passed around as a single LDS pointer) ----------------------
template <typename T>
struct PivotLds {
    T prow[NB];             // pivot row of this step (columns k..NB-1 valid)
    T wval[PANEL_WAVES];
    unsigned wpos[PANEL_WAVES];
    unsigned win;           // pivot position of this step
    int dead;
    T scale;                // 1/pivot (1 when the pivot is exactly zero), computed once per step
    int piv[NB];
    int spos[64];
    int rows[2 * NB];
    int content[2 * NB];
};

// ---- the per-step code is split so that ONLY the register-indexed parts (row publish, rank-1 update) are unrolled 64x;
// everything else lives in two out-of-line functions shared by all steps.
//
// step_a: candidate key of this thread's row -> wave argmax -> barrier -> workgroup winner (computed redundantly by every
//         thread from the 8 wave records).  The thread that owns the workgroup's winning row publishes the HEADER
//         {tag, position, a_pk} at once (the only thing the other workgroups need to pick the pivot) and returns true;
//         its row values follow from the unrolled caller while the header is already in flight.
template <typename T>
__device__ __noinline__ bool step_a(PivotLds<T>* sh, u64* scratch, unsigned epoch, int G, int k, int g, int tid, T aval,
                                    unsigned pos, bool act)
{
    const int lane = tid & 63, wave = tid >> 6;
    T key = T(-1);
    unsigned p = POS_NONE;
    if (act) {
        const T v = tabs(aval);
        key = (v > T(0)) ? v : T(0);  // NaN and 0 -> 0: never preferred, ties -> lowest position (src/lu.jl:298-304)
        p = pos;
    }
    wave_argmax<T>(key, p);
    if (lane == 0) { sh->wval[wave] = key; sh->wpos[wave] = p; }
    __syncthreads();
    RFLU_STAMP(scratch, k, 1, g, tid);
    // workgroup winner from the wave records: max of the keys (tree), then min position among the records holding it --
    // independent operations instead of a chain of 8 dependent (key,pos) compare-selects
    T kx[PANEL_WAVES];
    unsigned px[PANEL_WAVES];
#pragma unroll
    for (int x = 0; x < PANEL_WAVES; ++x) { kx[x] = sh->wval[x]; px[x] = sh->wpos[x]; }
    T cv;
    {
        T m01 = tmax(kx[0], kx[1]), m23 = tmax(kx[2], kx[3]), m45 = tmax(kx[4], kx[5]), m67 = tmax(kx[6], kx[7]);
        cv = tmax(tmax(m01, m23), tmax(m45, m67));
    }
    unsigned cp;
    {
        unsigned c[PANEL_WAVES];
#pragma unroll
        for (int x = 0; x < PANEL_WAVES; ++x) c[x] = (kx[x] == cv) ? px[x] : POS_NONE;
        cp = min(min(min(c[0], c[1]), min(c[2], c[3])), min(min(c[4], c[5]), min(c[6], c[7])));
    }
    const bool mine = act && pos == cp;
    if (G == 1) {
        if (mine) {
            sh->win = cp;
            sh->scale = (aval != T(0)) ? T(1) / aval : T(1);
        }
    } else {
        const unsigned tag = epoch + (unsigned)k;
        const unsigned hoff = (unsigned)(k & 1) * PS_BUF_BYTES + (unsigned)g * PS_HDR_BYTES;
        if (mine) Gran<T>::store_hdr(scratch_rsrc(scratch), hoff, tag, cp, aval);
        else if (cp == POS_NONE && tid == 0) Gran<T>::store_hdr(scratch_rsrc(scratch), hoff, tag, POS_NONE, T(0));
    }
    return mine;
}


// step_b: wave 0 polls the G headers (both 16-byte halves of a header in flight together), all workgroups arrive at the
//         same winner (max |a_pk|, lowest position), fetch the winner's row into LDS; barrier; position / ipiv / info
//         bookkeeping for this thread's row.
template <typename T>
__device__ __noinline__ MidOut<T> step_b(PivotLds<T>* sh, u64* scratch, int64_t* info, int64_t* ipiv, unsigned epoch,
                                         int G, int k, int r0, int g, int tid, unsigned pos, bool act)
{
    const int lane = tid & 63, wave = tid >> 6;
    if (G > 1 && wave == 0) {
        const __amdgpu_buffer_rsrc_t rs = scratch_rsrc(scratch);
        const unsigned tag = epoch + (unsigned)k;
        const unsigned base = (unsigned)(k & 1) * PS_BUF_BYTES;
        bool timed_out = false;
        T gv = T(-1);
        unsigned gp = POS_NONE;
        int gg = 0;
        T ga = T(0);  // signed a_pk of this lane's best candidate
        for (int x = lane; x < G; x += 64) {
            int spins = 0;
            for (;;) {
                unsigned xp;
                T xv;
                asm volatile("" ::: "memory");  // the buffer loads are plain (non-atomic) intrinsics: keep them in the loop
                if (Gran<T>::load_hdr(rs, base + (unsigned)x * PS_HDR_BYTES, tag, xp, xv)) {
                    if (xp != POS_NONE) {
                        const T av = tabs(xv);
                        const T xk = (av > T(0)) ? av : T(0);
                        if (better<T>(xk, xp, gv, gp)) { gv = xk; gp = xp; gg = x; ga = xv; }
                    }
                    break;
                }
                if (++spins > SPIN_LIMIT) { timed_out = true; break; }
                if (spins > 4) __builtin_amdgcn_s_sleep(1);
            }
        }
        {   // global winner: max key, lowest position; its workgroup index travels via the winning lane
            const T mykey = gv;
            const unsigned mypos = gp;
            wave_argmax<T>(gv, gp);
            const u64 who = __ballot(mykey == gv && mypos == gp && mypos != POS_NONE);
            const int wl = who ? (__ffsll((long long)who) - 1) : 0;
            gg = __builtin_amdgcn_readlane(gg, wl);
            ga = readlane_val(ga, wl);
        }
        RFLU_STAMP(scratch, k, 3, g, tid);
        if (lane >= k && lane < NB && gp != POS_NONE) {
            const unsigned roff = base + PS_HDR_REGION + (unsigned)gg * PS_ROW_BYTES + (unsigned)lane * PS_VAL_BYTES;
            T xv = T(0);
            int spins = 0;
            for (;;) {
                asm volatile("" ::: "memory");
                if (Gran<T>::load(rs, roff, tag, xv)) break;
                if (++spins > SPIN_LIMIT) { timed_out = true; break; }
            }
            sh->prow[lane] = xv;
        }
        if (lane == 0) {
            sh->win = gp;
            sh->scale = (ga != T(0)) ? T(1) / ga : T(1);  // overlaps with the row fetch above
        }
        if (__any(timed_out)) {
            if (lane == 0) {
                __hip_atomic_store((u64*)(info + 1), (u64)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sh->win = POS_NONE;
                sh->dead = 1;
            }
        }
    }
    __syncthreads();
    RFLU_STAMP(scratch, k, 4, g, tid);
    MidOut<T> o;
    o.scale = T(1);
    o.pos = pos;
    o.flags = (act ? 2u : 0u) | (sh->dead ? 4u : 0u);
    const unsigned win_pos = sh->win;
    if (win_pos == POS_NONE) return o;
    const T piv = sh->prow[k];
    const bool has = (piv != T(0));
    const unsigned kpos = (unsigned)(r0 + k);
    if (g == 0 && tid == 0) {
        ipiv[r0 + k] = (int64_t)win_pos + 1;
        if (!has && info[0] == 0) info[0] = (int64_t)r0 + k + 1;
    }
    o.scale = sh->scale;
    if (act) {
        if (pos == win_pos) {
            o.pos = kpos;      // pivot row: final position r0+k, no further updates
            o.flags &= ~2u;
        } else {
            if (pos == kpos) o.pos = win_pos;  // displaced row takes the pivot's old position
            o.flags |= 1u;
        }
    }
    return o;
}

// One pivot step, K a compile-time constant so that every register-array index below is static.
template <typename T, int K>
__device__ __forceinline__ void pivot_step(const PanelArgs<T>& p, PivotLds<T>* sh, T (&a)[NB], unsigned& pos, bool& act,
                                           bool& dead, PermState& perm, int g, int tid)
{
    if (K >= p.w || dead) return;
    RFLU_STAMP(p.scratch, K, 0, g, tid);
    const bool mine = step_a<T>(sh, p.scratch, p.epoch, p.G, K, g, tid, a[K], pos, act);
    RFLU_STAMP(p.scratch, K, 2, g, tid);
    if (mine) {  // this thread owns the workgroup's candidate row: hand out columns K..NB-1
        if (p.G == 1) {
#pragma unroll
            for (int j = K; j < NB; ++j) {
                T v = a[j];
                asm volatile("" : "+v"(v));  // keep hipcc from fusing the copies into a memcpy out of a scratch-resident a[]
                sh->prow[j] = v;
            }
        } else {
            const __amdgpu_buffer_rsrc_t rs = scratch_rsrc(p.scratch);
            const unsigned roff = (unsigned)(K & 1) * PS_BUF_BYTES + PS_HDR_REGION + (unsigned)g * PS_ROW_BYTES;
            const unsigned tag = p.epoch + (unsigned)K;
#pragma unroll
            for (int j = K; j < NB; ++j) Gran<T>::store(rs, roff + j * PS_VAL_BYTES, tag, a[j]);
        }
    }
    const MidOut<T> o = step_b<T>(sh, p.scratch, p.info, p.ipiv, p.epoch, p.G, K, p.r0, g, tid, pos, act);
    if (g == 0 && (tid >> 6) == PANEL_WAVES - 1) {  // interchange bookkeeping: one wave, registers only (sh->win is stable
        const unsigned wp = sh->win;                // until the next step's barrier)
        if (wp != POS_NONE) perm_state_step(perm, p.r0, K, __builtin_amdgcn_readfirstlane((int)wp), tid & 63);
    }
    RFLU_STAMP(p.scratch, K, 5, g, tid);
    pos = o.pos;
    act = (o.flags & 2u) != 0;
    dead = (o.flags & 4u) != 0;
    if (o.flags & 1u) {
        const T l = a[K] * o.scale;  // reciprocal-multiply (src/lu.jl:317-320); scale == 1 after a zero pivot
        a[K] = l;
#pragma unroll
        for (int j = K + 1; j < NB; ++j) a[j] -= l * sh->prow[j];
    }
    RFLU_STAMP(p.scratch, K, 6, g, tid);
}

template <typename T, int K0, int K1>
struct PivotSteps {
    static __device__ __forceinline__ void run(const PanelArgs<T>& p, PivotLds<T>* sh, T (&a)[NB], unsigned& pos,
                                               bool& act, bool& dead, PermState& perm, int g, int tid)
    {
        if constexpr (K0 < K1) {
            pivot_step<T, K0>(p, sh, a, pos, act, dead, perm, g, tid);
            PivotSteps<T, K0 + 1, K1>::run(p, sh, a, pos, act, dead, perm, g, tid);
        }
    }
};

template <typename T, int RT>
__global__ void __launch_bounds__(PANEL_THREADS) panel_pivot_kernel(PanelArgs<T> p)
{
    static_assert(RT == 1, "one row per thread");
    __shared__ PivotLds<T> s_lds;
    PivotLds<T>* const sh = &s_lds;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.x;
    const int w = p.w;
    const int row_base = p.r0 + g * PANEL_THREADS;

    RFLU_STAMP(p.scratch, NB, 0, g, tid);
    T a[1][NB];
    const int row = row_base + tid;
    bool act = row < p.m;                          // still a pivot candidate (not yet chosen, inside the matrix)
    unsigned pos = act ? (unsigned)row : POS_NONE; // current row position of this thread's row
    if (tid == 0) sh->dead = 0;
    load_row_direct<T>(p.R, p.ld, row, act, p.c0, w, a[0]);
    __syncthreads();
    RFLU_STAMP(p.scratch, NB, 1, g, tid);

    bool dead = false;  // set (workgroup-uniformly) after a timeout: skip the remaining steps quickly
    PermState perm = perm_state_init(lane);
    PivotSteps<T, 0, NB>::run(p, sh, a[0], pos, act, dead, perm, g, tid);

    RFLU_STAMP(p.scratch, NB, 2, g, tid);
    store_row_direct<T>(p.R, p.ld, pos, p.c0, w, a[0]);
    __syncthreads();
    RFLU_STAMP(p.scratch, NB, 3, g, tid);

    if (g == 0 && wave == PANEL_WAVES - 1) {
        const int chunk = p.r0 / NB;
        perm_state_finish(perm, p.r0, lane, sh->rows, p.pm_cnt + chunk, p.pm_dst + (size_t)chunk * 2 * NB,
                          p.pm_src + (size_t)chunk * 2 * NB);
    }
}

// =====================================================================================================================
// NoPivot leaf panel (Val(false) / NoPivot()): no exchange between workgroups.
//   kernel 1 (one workgroup): unpivoted LU of the w x w top block, in place.
//   kernel 2 (G workgroups) : every row below solves  l_i * U11 = a_i  against the factored top block held in LDS.
// =====================================================================================================================
constexpr int NPT_THREADS = 256;   // 64 rows x 4 column phases: every thread has work, half the waves at the two barriers per column
template <typename T>
__global__ void __launch_bounds__(NPT_THREADS) panel_nopivot_top_kernel(PanelArgs<T> p)
{
    __shared__ T s_U[NB * TILE_LD];
    constexpr int PANEL_WAVES = NPT_THREADS / 64;   // (shadows the 8 waves of the cooperative leaves)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int w = p.w;
    for (int rr = wave; rr < NB; rr += PANEL_WAVES) {
        T v = T(0);
        if (rr < w && lane < w) v = p.R[(int64_t)(p.r0 + rr) * p.ld + p.c0 + lane];
        s_U[rr * TILE_LD + lane] = v;
    }
    __syncthreads();
    // thread (i = tid>>2, part = tid&3) updates columns j = k+1+part, +4, ... of row i
    const int i = tid >> 2, part = tid & 3;
    for (int k = 0; k < w; ++k) {
        const T piv = s_U[k * TILE_LD + k];
        const bool has = (piv != T(0));
        const T inv = has ? T(1) / piv : T(1);
        T l = T(0);
        if (i > k && i < w) {
            l = s_U[i * TILE_LD + k];
            if (has) l *= inv;
        }
        __syncthreads();
        if (i > k && i < w) {
            if (part == 0) s_U[i * TILE_LD + k] = l;
            for (int j = k + 1 + part; j < w; j += 4) s_U[i * TILE_LD + j] -= l * s_U[k * TILE_LD + j];
        }
        if (tid == 0 && !has && p.info[0] == 0) p.info[0] = (int64_t)p.r0 + k + 1;
        __syncthreads();
    }
    for (int rr = wave; rr < w; rr += PANEL_WAVES)
        if (lane < w) p.R[(int64_t)(p.r0 + rr) * p.ld + p.c0 + lane] = s_U[rr * TILE_LD + lane];
}

template <typename T, int RT, int K0, int K1>
struct NoPivotSteps {
    static __device__ __forceinline__ void run(int w, const T* s_U, T (&a)[RT][NB])
    {
        if constexpr (K0 < K1) {
            if (K0 < w) {
                const T piv = s_U[K0 * TILE_LD + K0];
                const bool has = (piv != T(0));
                const T inv = has ? T(1) / piv : T(1);
#pragma unroll
                for (int q = 0; q < RT; ++q) {
                    T l = a[q][K0];
                    if (has) l *= inv;
                    a[q][K0] = l;
#pragma unroll
                    for (int j = K0 + 1; j < NB; ++j) a[q][j] -= l * s_U[K0 * TILE_LD + j];
                }
            }
            NoPivotSteps<T, RT, K0 + 1, K1>::run(w, s_U, a);
        }
    }
};

template <typename T, int RT>
__global__ void __launch_bounds__(PANEL_THREADS) panel_nopivot_rows_kernel(PanelArgs<T> p)
{
    __shared__ T s_tile[64 * TILE_LD];
    __shared__ T s_U[NB * TILE_LD];  // factored top block L11\U11 (only U11 is used)
    __shared__ int s_spos[64];

    if (p.info[0] == 0) return;   // regular case: panel_nopivot_rows_mfma_kernel has done these rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.x;
    const int w = p.w;
    const int row_base = p.r0 + w + g * (PANEL_THREADS * RT);

    for (int rr = wave; rr < NB; rr += PANEL_WAVES) {
        T v = T(0);
        if (rr < w && lane < w) v = p.R[(int64_t)(p.r0 + rr) * p.ld + p.c0 + lane];
        s_U[rr * TILE_LD + lane] = v;
    }
    T a[RT][NB];
    unsigned pos[RT];
#pragma unroll
    for (int q = 0; q < RT; ++q) {
        const int row = row_base + q * PANEL_THREADS + tid;
        pos[q] = (row < p.m) ? (unsigned)row : POS_NONE;
    }
    load_rows<T, RT>(p.R, p.ld, row_base, p.m, p.c0, w, a, s_tile, wave, lane);  // contains __syncthreads
    NoPivotSteps<T, RT, 0, NB>::run(w, s_U, a);
    store_rows<T, RT>(p.R, p.ld, p.c0, w, a, pos, s_tile, s_spos, wave, lane);
}

// NoPivot leaf, round 2: the rows below are  L21 = A21 * inv(U11)  as an MFMA product instead of a 64-step substitution per
// row (59 -> ~10 us at 16384 rows).  After the top block's LU one launch of two workgroups inverts L11 (for the TRSMs that
// follow, as the pivoted path does in its interchange launch) and U11; the product kernel needs no LDS.  A zero pivot
// (info != 0) makes inv(U11) meaningless: then -- and only then -- the substitution kernel does the rows, with the
// reference's zero-pivot-continue semantics (src/lu.jl:316-330); both kernels are launched and one of them returns at once.
template <typename T>
__global__ void __launch_bounds__(256) panel_nopivot_inv_kernel(PanelArgs<T> p, T* __restrict__ linv, T* __restrict__ uinv)
{
    __shared__ T sL[NB * NB];
    __shared__ T sX[NB * NB];
    const T* blk = p.R + (int64_t)p.r0 * p.ld + p.c0;
    if (blockIdx.x == 0) diag_inv_block4<T, false>(p.w, blk, p.ld, linv, sL, sX, threadIdx.x);
    else if (p.info[0] == 0) diag_inv_block4<T, true>(p.w, blk, p.ld, uinv, sL, sX, threadIdx.x);
}

template <typename T>
struct NpMfma;
template <>
struct NpMfma<double> {
    typedef double acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t run(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int crow(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <>
struct NpMfma<float> {
    typedef float acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int crow(int lane, int r) { return 4 * (lane >> 4) + r; }
};

// one workgroup = 64 rows below the top block, wave = 16 of them: X = A21 * inv(U11), in place
template <typename T>
__global__ void __launch_bounds__(256) panel_nopivot_rows_mfma_kernel(PanelArgs<T> p, const T* __restrict__ uinv)
{
    if (p.info[0] != 0) return;   // a zero pivot somewhere: the substitution kernel takes over (uniform for the whole grid)
    typedef typename NpMfma<T>::acc_t acc_t;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 15, fk = lane >> 4;
    const int w = p.w;
    const int64_t row0 = (int64_t)p.r0 + w + (int64_t)blockIdx.x * 64 + wave * 16;
    T a[16];
    {
        const bool rok = row0 + fi < p.m;
        const T* Ap = p.R + (row0 + fi) * p.ld + p.c0 + fk;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) a[kk] = (rok && kk * 4 + fk < w) ? Ap[kk * 4] : T(0);
    }
    acc_t x[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) x[t] = acc_t{T(0), T(0), T(0), T(0)};
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)
#pragma unroll
        for (int t = 0; t < 4; ++t) x[t] = NpMfma<T>::run(a[kk], uinv[(kk * 4 + fk) * NB + t * 16 + fi], x[t]);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = row0 + NpMfma<T>::crow(lane, r);
            const int col = t * 16 + fi;
            if (row < p.m && col < w) p.R[row * p.ld + p.c0 + col] = x[t][r];
        }
}

template <typename T>
int launch_panel(Handle* h, T* R, int64_t ld, int64_t m, int64_t r0, int64_t c0, int64_t w, int64_t* ipiv, int pivot)
{
    if (w <= 0) return RFLU_OK;
    if (w > NB || r0 % NB != 0 || m - r0 < w) {
        set_error("launch_panel: unsupported geometry m=%lld r0=%lld w=%lld", (long long)m, (long long)r0, (long long)w);
        return RFLU_ERR_ARG;
    }
    const int64_t rows = m - r0;
    int rt = 1;
    while ((rows + (int64_t)PANEL_THREADS * rt - 1) / ((int64_t)PANEL_THREADS * rt) > MAX_PANEL_WGS) rt *= 2;
    if (rt > 1) {
        set_error("launch_panel: %lld rows exceed the supported %d", (long long)rows, MAX_PANEL_WGS * PANEL_THREADS);
        return RFLU_ERR_ARG;
    }
    PanelArgs<T> p;
    p.R = R; p.ld = ld; p.m = (int)m; p.r0 = (int)r0; p.c0 = (int)c0; p.w = (int)w;
    p.ipiv = ipiv; p.info = h->info_dev; p.scratch = h->pscratch;
    p.G = (int)((rows + (int64_t)PANEL_THREADS * rt - 1) / ((int64_t)PANEL_THREADS * rt));
    p.pm_cnt = h->pm_cnt; p.pm_dst = h->pm_dst; p.pm_src = h->pm_src;
    if (pivot && p.G > 1 && p.G > h->panel_max_wgs) {   // the workgroups spin on each other: all must be resident at once
        set_error("launch_panel: %d cooperating workgroups, but the device holds %d at a time", p.G, h->panel_max_wgs);
        return RFLU_ERR_ARG;
    }
    p.epoch = h->epoch;
    h->epoch += (unsigned)NB;
    if (h->epoch > 0xfffff000u) {  // tag wrap: wipe the records and restart the epoch counter
        RFLU_HIP(hipMemsetAsync(h->pscratch, 0, h->pscratch_bytes, h->stream));
        h->epoch = 1;
        p.epoch = h->epoch;
        h->epoch += (unsigned)NB;
    }
    ProfScope ps(h, RFLU_K_PANEL, (double)rows * (double)w * (double)w);
#ifdef RFLU_EXPERIMENTS
    if (pivot && w == NB && panel_use_blocked(h, rows, sizeof(T))) {
        // the sub-panel kernel (panel_blocked.hip); XCD-local records for the short panels as below
        const int64_t local_rows = h->tune.panel_local_rows >= 0 ? h->tune.panel_local_rows : (sizeof(T) == 4 ? 8192 : 4096);
        const bool loc = h->panel_local == 1 || (h->panel_local == 2 && h->num_cus == 256 && rows <= local_rows);
        RFLU_TRY(launch_panel_blocked<T>(h, p, loc ? 1 : 0));
        return RFLU_OK;
    }
#endif
    if (pivot && rows <= PANEL_THREADS && h->panel_single) {   // one workgroup, LDS only (panel_single.hip)
        RFLU_TRY(launch_panel_single<T>(h, p));
        return RFLU_OK;
    }
    if (pivot) {
        // Full leaves of 2..64 workgroups: the pipelined cooperative leaf with a communication wave (panel_local.hip; one header
        // per lane of that wave).  A narrower last leaf (w < NB, matrices whose width is no multiple of 64) and RFLU_PANEL_LOCAL=0
        // take the two-trip kernel above.
        const int64_t local_min = h->tune.panel_local_min;   // panels of 257..512 rows with RFLU_PANEL_SINGLE=0: 8 workgroups of 64 rows on one XCD
        const bool tiny_local = h->panel_local == 2 && h->num_cus == 256 && rows > local_min && rows <= 512 && !h->coop_launch;
        if (h->panel_local > 0 && w == NB && (p.G >= 2 || tiny_local) && p.G <= std::min(h->panel_local_maxg, 64)) {
            // Panels of at most 4096 rows (<= 16 workgroups of 256 rows) run the XCD-local variant: all participants on ONE XCD, plain-store
            // records that stay in that XCD's L2, a 0.34 us hop instead of 0.56-0.75.  The launch has 8 G workgroups of which the 7 G on
            // the other XCDs exit at once; they still have to be placed, which next to a big update costs more than the hop saves (N=8192
            // with every panel local: 31.9 vs 26.4 ms) -- but not once the panel is this short and the update streams leave most CUs
            // alone: N=2048 5.76 -> 5.39 ms, N=4096 11.97 -> 11.02, N=8192 26.48 -> 25.54, N=16384 80.46 -> 79.49 (6144 rows: 26.8 at
            // N=8192).  RFLU_PANEL_LOCAL_ROWS=0 switches it off, RFLU_PANEL_LOCAL=1 forces it for every panel.
            // Float32: the update is half as heavy, the 32 participants of an 8192-row panel still find their XCD (N=16384 61.9 -> 59.9-60.7 ms,
            // N=8192 22.45 -> 22.18; 12288 rows: 62.1)
            const int64_t local_rows = h->tune.panel_local_rows >= 0 ? h->tune.panel_local_rows : (sizeof(T) == 4 ? 8192 : 4096);
            const bool loc = h->panel_local == 1 || (h->num_cus == 256 && rows <= local_rows && (rows > 512 || tiny_local));
            RFLU_TRY(launch_panel_local<T>(h, p, loc ? 8 : 1, loc ? h->panel_xcc : 0, loc ? h->panel_xcc : -1, loc));
            return RFLU_OK;
        }
        void* kargs[] = {&p};
        if (h->coop_launch && p.G > 1) RFLU_HIP(hipLaunchCooperativeKernel(reinterpret_cast<const void*>(&panel_pivot_kernel<T, 1>), dim3(p.G), dim3(PANEL_THREADS), kargs, 0, h->stream));
        else hipLaunchKernelGGL((panel_pivot_kernel<T, 1>), dim3(p.G), dim3(PANEL_THREADS), 0, h->stream, p);
    } else {
        hipLaunchKernelGGL((panel_nopivot_top_kernel<T>), dim3(1), dim3(NPT_THREADS), 0, h->stream, p);
        // inverses: L11 -> this leaf's slot of h->linv (what launch_diag_inv would compute), U11 -> the spare last slot
        T* linv = static_cast<T*>(h->linv) + (r0 / NB) * NB * NB;
        T* uinv = static_cast<T*>(h->linv) + (h->pm_chunks - 1) * NB * NB;
        hipLaunchKernelGGL((panel_nopivot_inv_kernel<T>), dim3(2), dim3(256), 0, h->stream, p, linv, uinv);
        const int64_t below = rows - w;
        if (below > 0) {
            hipLaunchKernelGGL((panel_nopivot_rows_mfma_kernel<T>), dim3((unsigned)((below + 63) / 64)), dim3(256), 0, h->stream, p, uinv);
            const int gb = (int)((below + (int64_t)PANEL_THREADS * rt - 1) / ((int64_t)PANEL_THREADS * rt));
            hipLaunchKernelGGL((panel_nopivot_rows_kernel<T, 1>), dim3(gb), dim3(PANEL_THREADS), 0, h->stream, p);
        }
    }
    RFLU_HIP(hipGetLastError());
    return RFLU_OK;
}

template <typename T>
static int panel_resident_limit_t(int num_cus)
{
    int worst = 1 << 30;
    auto ask = [&](const void* fn) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, PANEL_THREADS, 0) != hipSuccess) { (void)hipGetLastError(); nb = 0; }
        worst = std::min(worst, nb * num_cus);
    };
    ask(reinterpret_cast<const void*>(&panel_pivot_kernel<T, 1>));
    return worst;
}
// Two translation units compile this file in parallel (build.py): panel.hip itself instantiates the Float64 kernels and
// holds the non-template functions, panel_f32.hip (#define RFLU_PANEL_F32_TU, #include "panel.hip") the Float32 kernels.
#ifdef RFLU_PANEL_F32_TU
template int launch_panel<float>(Handle*, float*, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t*, int);
int panel_resident_limit_f32(int num_cus) { return panel_resident_limit_t<float>(num_cus); }
#else
int panel_resident_limit_f64(int num_cus) { return panel_resident_limit_t<double>(num_cus); }
template int launch_panel<double>(Handle*, double*, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t*, int);

int panel_resident_limit(int num_cus)
{
    int lim = std::min(std::min(panel_resident_limit_f64(num_cus), panel_resident_limit_f32(num_cus)),
                       std::min(panel_local_resident_limit_f64(num_cus), panel_local_resident_limit_f32(num_cus)));
#ifdef RFLU_EXPERIMENTS
    lim = std::min(lim, std::min(panel_blocked_resident_limit_f64(num_cus), panel_blocked_resident_limit_f32(num_cus)));
#endif
    return lim;
}

// The sub-panel kernel takes a full pivoted leaf whenever it does not need a bigger CU reservation (a multiple of 32) than the
// 512-row workgroups of the older leaves would: Float64 workgroups hold 448 rows, so panels of 14337..16384 rows (and 28673..32768)
// stay with the older kernel -- the block columns of N = 16384 whose update, not whose panel, sets the pace.
// (the sub-panel leaf is an experiment that measured no faster than the default leaves -- DESIGN.md section 9: it is compiled only
//  into an RFLU_EXPERIMENTS build, `RFLU_EXPERIMENTS=1 python recursivefactorization.jl_amd/build.py`)
bool panel_use_blocked(const Handle* h, int64_t rows, size_t esize)
{
#ifndef RFLU_EXPERIMENTS
    (void)h; (void)rows; (void)esize;
    return false;
#else
    if (!h->panel_blocked || h->coop_launch) return false;
    const int64_t rpw = esize == 8 ? PANEL_BLOCKED_ROWS_F64 : PANEL_BLOCKED_ROWS_F32;
    const int64_t g = (rows + rpw - 1) / rpw, g_old = (rows + PANEL_THREADS - 1) / PANEL_THREADS;
    if (g > 64) return false;
    return (g + 31) / 32 <= (g_old + 31) / 32;
#endif
}

int64_t panel_plan_wgs(const Handle* h, int64_t rows, size_t esize, int pivot)
{
    rows = std::max<int64_t>(rows, 1);
    if (pivot && panel_use_blocked(h, rows, esize)) {
        const int64_t rpw = esize == 8 ? PANEL_BLOCKED_ROWS_F64 : PANEL_BLOCKED_ROWS_F32;
        return (rows + rpw - 1) / rpw;
    }
    const int64_t rpw = (pivot && h->panel_local > 0) ? panel_local_rows_per_wg(h, rows, esize) : PANEL_THREADS;
    return (rows + rpw - 1) / rpw;
}

size_t panel_scratch_bytes() { return (PX_OFFSET_WORDS + PX_BYTES / 8 + RFLU_TRACE_ALL_WORDS) * sizeof(u64); }  // records | trace stamps | pair slots | all-workgroup trace
size_t panel_trace_all_offset_bytes() { return (PX_OFFSET_WORDS + PX_BYTES / 8) * sizeof(u64); }
size_t panel_trace_all_words() { return RFLU_TRACE_ALL_WORDS; }
size_t panel_trace_offset_bytes() { return PS_TOTAL_WORDS * sizeof(u64); }
#endif

}  // namespace rflu
