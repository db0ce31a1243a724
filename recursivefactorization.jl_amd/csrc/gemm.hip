// gemm.hip -- trailing update  C <- C - A*B  on the MFMA matrix cores of gfx950 (CDNA4).
//
// Replaces schur_complement! (/root/reference/src/lu.jl:265-284: C[m,n] = (0 - sum_k A[m,k]*B[k,n]) + C[m,n]) and
// supplies the off-diagonal work of the recursive TRSM (src/lu.jl:235).  Dominant kernel of the path: N^3/2 of the
// 2N^3/3 LU flops plus ~90 % of the TRSM's N^3/6.
//
// Shape: all operands row-major (R layout).  One workgroup = 256 threads = 4 waves computes a 128x128 tile of C; each
// wave owns a 64x64 sub-tile as 4x4 MFMA fragments of 16x16 (v_mfma_f64_16x16x4_f64, or v_mfma_f32_16x16x4_f32 for
// Float32 -- both take ONE scalar per lane for A and B).  K is walked in slabs of 16 through a double-buffered LDS
// image (one barrier per slab): global loads for slab t+1 are issued before the 64 MFMAs of slab t and written to the
// other LDS buffer after them, so HBM/L2 latency hides behind ~4096 cycles of matrix work per wave.
//
// LDS images.  hipcc pairs the fragment reads into ds_read2_b64 and the staging writes are ds_write_b64: both are serviced
// in groups of 16 CONSECUTIVE lanes on a 32-bank map, bank = (byte address / 4) % 32 (MI355X_MICROARCH.md "LDS").  Round 1
// laid the images out for the 64-bank / 32-lane rule of a lone ds_read_b64 and paid for it: SQ_LDS_BANK_CONFLICT 8.1e9 cycles
// where the vendor BLAS kernel of the same tile shape shows 0 (scripts/pmc_gemm.sh), the B staging writes 8-way conflicted.
//   As[i][k], row stride SA = BK+1 = 17 doubles: the 16 lanes of a group read rows i..i+15 at one k -> dword 34*i, and
//             34*i mod 32 = 2*i: 16 distinct bank pairs; the staging writes (8 rows x 2 k-halves per group) land on
//             2*r + {0,16}: distinct as well
//   Bs[k][c + (c >> 4)], row stride SB = 144 doubles: a one-double gap after every 16 columns.  A group's fragment read
//             covers 16 consecutive columns of one block (contiguous, all 32 banks once); a staging write has lane l at
//             columns 8l..8l+7 -> dword 16*l + 2*(l/2) + 2e = {2a, 16+2a}: distinct bank pairs (was: 16*l -> 8-way)
// Two workgroups per CU (2 x 72 KiB LDS, <=256 VGPRs) keep each SIMD's matrix pipe fed across barriers.
//
// Roofline: bound = fp64 MFMA, 78.6 TFLOP/s chip peak (256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz);
// algorithmic work 2*M*N*K flops per launch.
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "rflu_internal.hpp"

namespace rflu {

template <typename T>
struct Mfma;

template <>
struct Mfma<double> {
    typedef double acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t run(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    // c - a*b in one instruction: for the Float64 MFMAs the BLGP field is a set of NEGATE bits (bit 0: A) -> "neg:[1,0,0]"
    static constexpr bool HAS_NEG = true;
    static __device__ __forceinline__ acc_t run_neg(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 1);
    }
    // f64 C/D fragment: col = lane & 15, row = (lane >> 4) + 4*r   (cdna_hip_programming.md section 3)
    static __device__ __forceinline__ int crow(int lane, int r) { return (lane >> 4) + 4 * r; }
};

template <>
struct Mfma<float> {
    typedef float acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    static constexpr bool HAS_NEG = false;   // BLGP is a lane-group broadcast pattern for the Float32 MFMAs (no negate bits)
    static __device__ __forceinline__ acc_t run_neg(float a, float b, acc_t c) { return run(-a, b, c); }   // never instantiated in a hot loop
    // f32 C/D fragment: col = lane & 15, row = 4*(lane >> 4) + r
    static __device__ __forceinline__ int crow(int lane, int r) { return 4 * (lane >> 4) + r; }
};

constexpr int G_BM = 128, G_BN = 128, G_BK = 16;
constexpr int G_SA = G_BK + 1;
constexpr int G_SB = G_BN + 16;
constexpr int G_STAGE = G_BM * G_SA + G_BK * G_SB;  // elements per LDS stage
#ifndef RFLU_GEMM_GROUP_M
#define RFLU_GEMM_GROUP_M 8
#endif
constexpr int G_GROUP_M = RFLU_GEMM_GROUP_M;        // tile rows walked together (L2 reuse of the B panel)
#ifndef RFLU_GEMM_SSTORE_AT
#define RFLU_GEMM_SSTORE_AT 4
#endif
#ifndef RFLU_GEMM_PRIO
#define RFLU_GEMM_PRIO 1
#endif
// Wave priority during a slab's MFMA burst, back to 0 for the LDS writes and the barrier: with two workgroups per CU the wave
// that is inside its burst keeps the matrix pipe while the other one's staging instructions fill the gaps
// (sustained 15872^2 x 512: 64.3 -> 65.4 TFLOP/s; 1 and 3 measure the same).  0 = leave the priority alone.
constexpr int G_PRIO = RFLU_GEMM_PRIO;
constexpr int G_SSTORE_AT = RFLU_GEMM_SSTORE_AT;
#ifndef RFLU_GEMM_CSPLIT
#define RFLU_GEMM_CSPLIT 3
#endif
constexpr int G_CSPLIT = RFLU_GEMM_CSPLIT;          // C fragments rows [0, G_CSPLIT) are requested before the first LDS fill, the rest after it
    // before which 4-deep step of a slab the next slab is written to LDS (4 = after the last)

template <typename T>
struct GemmArgs {
    int M, N, K;
    const T* A;
    int64_t lda;
    const T* B;
    int64_t ldb;
    T* C;
    int64_t ldc;
    int tiles_m, tiles_n;
    int vec_ok;  // operands 16-byte aligned with even strides: full tiles may use 16-byte loads
    int flags;   // bit1: non-temporal C accesses (bit0, "operand slab before the C tile", is always on since round 4)
    // "first columns first": the tiles of the first na_tiles_n tile columns take the lowest block indices (they are dispatched,
    // hence finished, first) and the last of them to finish publishes sig_val in *sig_flag -- the lookahead schedule's
    // critical path needs only those columns of a bulk update (driver.cpp: factor_lookahead, GemmSignal)
    int na_tiles_n;
    unsigned long long* sig_flag;
    unsigned long long sig_val;
    unsigned* sig_cnt;
#if defined(RFLU_GEMM_TRACE)
    unsigned long long* stamps;   // experiments (scripts/gemm_phase_trace.py): 8 words per workgroup
#endif
};

#if defined(RFLU_GEMM_TRACE)
static unsigned long long* g_gemm_stamps = nullptr;
extern "C" void rflu_debug_gemm_stamps(unsigned long long* p) { g_gemm_stamps = p; }
#define GEMM_STAMP(k)                                                                                                  \
    do {                                                                                                               \
        if (g.stamps && tid == 0) g.stamps[(size_t)blockIdx.x * 8 + (k)] = wall_clock64(); \
    } while (0)
#else
#define GEMM_STAMP(k) do { } while (0)
#endif

// One 128x128 tile of C (see gemm_sub_kernel).  INTERIOR: see the call site.
template <typename T, bool C_FIRST, bool INTERIOR>
__device__ __forceinline__ void gemm_tile(const GemmArgs<T>& g, T* smem, const int m0, const int n0)
{
    typedef typename Mfma<T>::acc_t acc_t;
    constexpr int VW = 16 / (int)sizeof(T);  // elements per 16-byte vector
    typedef T vec_t __attribute__((ext_vector_type(VW)));
    constexpr bool NEGMOD = C_FIRST && Mfma<T>::HAS_NEG;
    constexpr bool NEGACC = C_FIRST && !Mfma<T>::HAS_NEG;
    // SPLITC: part of the C tile requested after the first LDS fill, first and last K slab written out (below).  Float64 only:
    // Float32 measures 3 % slower with it (15872^2 x 512: 124.4 vs 127.4 TFLOP/s, also with the sign flip of its accumulators moved
    // to their first use) and keeps the round-3 order: whole C tile, then the first fill, one loop.
    constexpr bool SPLITC = INTERIOR && sizeof(T) == 8;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    // ---- global -> register staging map: A slab 128x16 (8 consecutive k per thread), B slab 16x128 (8 consecutive j)
    const int a_row = tid >> 1, a_kb = (tid & 1) * 8;
    const int b_k = tid >> 4, b_jb = (tid & 15) * 8;
    const T* Ap = g.A + (int64_t)(m0 + a_row) * g.lda + a_kb;
    const T* Bp = g.B + (int64_t)b_k * g.ldb + n0 + b_jb;
    const bool a_row_ok = (m0 + a_row) < g.M;
    const bool full_mn = INTERIOR || (g.vec_ok && (m0 + G_BM <= g.M) && (n0 + G_BN <= g.N));

#if defined(RFLU_GEMM_TRACE)
    if (g.stamps && tid == 0) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g.stamps[(size_t)blockIdx.x * 8 + 4] = ((unsigned long long)xcc << 32) | hwid;
    }
#endif
    GEMM_STAMP(0);
    T ra[8], rb[8];

    auto gload = [&](int k0) {
        if (INTERIOR || (full_mn && (k0 + G_BK <= g.K))) {
#pragma unroll
            for (int v = 0; v < 8 / VW; ++v) {
                vec_t x = *reinterpret_cast<const vec_t*>(Ap + k0 + v * VW);
                vec_t y = *reinterpret_cast<const vec_t*>(Bp + (int64_t)k0 * g.ldb + v * VW);
#pragma unroll
                for (int e = 0; e < VW; ++e) {
                    ra[v * VW + e] = x[e];
                    rb[v * VW + e] = y[e];
                }
            }
        } else {
            const bool bk_ok = (k0 + b_k) < g.K;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                ra[e] = (a_row_ok && (k0 + a_kb + e) < g.K) ? Ap[k0 + e] : T(0);
                rb[e] = (bk_ok && (n0 + b_jb + e) < g.N) ? Bp[(int64_t)k0 * g.ldb + e] : T(0);
            }
        }
    };
    auto sstore = [&](int stage) {
        T* As = smem + stage * G_STAGE;
        T* Bs = As + G_BM * G_SA;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            // C_FIRST: the accumulators start as C itself and the products enter negated -- by the MFMA's own negate bit (Float64)
            // or through negated accumulators (Float32), see NEGMOD / NEGACC; otherwise the subtraction happens in the epilogue
            As[a_row * G_SA + a_kb + e] = ra[e];
            Bs[b_k * G_SB + b_jb + (b_jb >> 4) + e] = rb[e];
        }
    };

    // The first operand slab is requested BEFORE the C tile: the memory counter retires in order, so the wait in front of
    // sstore(0) covers these few loads only and the C loads keep flying behind the LDS fill, the barrier and the first MFMAs
    // (each fragment is waited for where it is first used) -- IF the compiler can say so: the counter of a wave counts to 63,
    // and with the slab's 8 loads and all 64 C loads in flight it emitted vmcnt(0), i.e. every workgroup sat through its whole C
    // tile before its first LDS fill (11 us of a 127 us tile, scripts/gemm_phase_trace.py).  So the last G_CSPLIT..3 rows of
    // fragments are requested after the first barrier: 8 + 48 loads in flight at the wait, vmcnt(48).
    const bool ntc = (g.flags & 2) != 0;
    gload(0);
    // For a fixed (i,j,r) sixteen lanes cover 16 consecutive columns of one row of C.
    acc_t acc[4][4];
    auto load_c = [&](int i_lo, int i_hi) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < i_lo || i >= i_hi) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wr * 64 + i * 16 + Mfma<T>::crow(lane, r);
                const T* crow_p = g.C + (int64_t)row * g.ldc + n0 + wc * 64 + (lane & 15);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int col = n0 + wc * 64 + j * 16 + (lane & 15);
                    T cv = T(0);
#ifdef RFLU_GEMM_EXP_NOCLOAD   // timing experiment (wrong results): the tile without its C read
                    if (false)
#else
                    if (C_FIRST && (full_mn || (row < g.M && col < g.N)))
#endif
                        cv = ntc ? __builtin_nontemporal_load(crow_p + j * 16) : crow_p[j * 16];
                    acc[i][j][r] = (NEGACC && !SPLITC) ? -cv : cv;
                }
            }
        }
    };
    const int nk = (g.K + G_BK - 1) / G_BK;
    load_c(0, SPLITC ? G_CSPLIT : 4);   // (the other copies keep their K loop free of waits for C: everything before the first fill)
#if defined(RFLU_GEMM_TRACE) && RFLU_GEMM_TRACE == 2   // prologue split: loads issued / first slab arrived and written / barrier passed
    GEMM_STAMP(6);
    sstore(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    GEMM_STAMP(7);
    __syncthreads();
#else
    sstore(0);
    __syncthreads();
#endif
    if (SPLITC) load_c(G_CSPLIT, 4);
    GEMM_STAMP(1);
#if defined(RFLU_GEMM_TRACE) && RFLU_GEMM_TRACE != 2
    if (g.stamps && tid == 0) g.stamps[(size_t)blockIdx.x * 8 + 6] = clock64();
#endif

    const int a_frag = (wr * 64 + (lane & 15)) * G_SA + (lane >> 4);
    const int b_frag = (lane >> 4) * G_SB + wc * 68 + (lane & 15);   // column c sits at c + (c >> 4): wc*64 -> wc*68

    // one K slab; `more`: another slab follows (its global loads and its LDS fill belong to this one)
    auto slab = [&](int kt, bool more, auto first) {
        const int cur = kt & 1;
#ifndef RFLU_GEMM_EXP_NOLOAD   // timing experiments only (wrong results): what a slab costs without its global loads / its barrier
        if (more) gload((kt + 1) * G_BK);
#endif
        const T* As = smem + cur * G_STAGE;
        const T* Bs = As + G_BM * G_SA;
        if (G_PRIO) __builtin_amdgcn_s_setprio(G_PRIO);
#pragma unroll
        for (int kk = 0; kk < G_BK / 4; ++kk) {
            // G_SSTORE_AT < 4: the next slab goes into the OTHER LDS buffer (free since the last barrier) in the middle of this slab's
            // MFMAs, so that its 16 LDS writes and the wait for its global loads sit in the shadow of matrix instructions
            if (kk == G_SSTORE_AT && more) sstore(cur ^ 1);
            T a[4], b[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                a[t] = As[a_frag + t * 16 * G_SA + kk * 4];
                b[t] = Bs[b_frag + kk * 4 * G_SB + t * 17];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (NEGACC && SPLITC && decltype(first)::value && kk == 0) {
                    // Float32: the accumulators hold -(c - sum a*b).  The sign flip waits for the C fragment, so it sits here, in
                    // front of the fragment's first use, not behind the load (a wait for the whole tile before the first LDS fill)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = -acc[i][j];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = NEGMOD ? Mfma<T>::run_neg(a[i], b[j], acc[i][j]) : Mfma<T>::run(a[i], b[j], acc[i][j]);
            }
        }
        if (G_PRIO) __builtin_amdgcn_s_setprio(0);
        if (G_SSTORE_AT >= G_BK / 4 && more) sstore(cur ^ 1);
#ifndef RFLU_GEMM_EXP_NOBARRIER
        __syncthreads();
#endif
    };
    // The first slab is a copy of its own: the waits for the C fragments belong to it alone (in the loop they would be executed --
    // as waits for the operand loads just issued -- in every slab)
    if (SPLITC) {   // nk >= 2; first and last slab written out: no branch around a load anywhere
        slab(0, true, std::true_type{});
        for (int kt = 1; kt + 1 < nk; ++kt) slab(kt, true, std::false_type{});
        slab(nk - 1, false, std::false_type{});
    } else {
        for (int kt = 0; kt < nk; ++kt) slab(kt, kt + 1 < nk, std::false_type{});
    }

    GEMM_STAMP(2);
#if defined(RFLU_GEMM_TRACE) && RFLU_GEMM_TRACE != 2
    if (g.stamps && tid == 0) g.stamps[(size_t)blockIdx.x * 8 + 7] = clock64();
#endif
    // ---- epilogue: C = C_in - A*B   (C_FIRST: acc already holds it)
    // (interior tiles without the per-element bounds tests: predicated, each of the 64 loads of the prologue and the 64 stores here
    // is ~6 scalar/vector instructions around one memory instruction)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + wr * 64 + i * 16 + Mfma<T>::crow(lane, r);
            if (full_mn || row < g.M) {
                T* crow_p = g.C + (int64_t)row * g.ldc + n0 + wc * 64 + (lane & 15);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int col = n0 + wc * 64 + j * 16 + (lane & 15);
                    if (full_mn || col < g.N) {
                        if (C_FIRST) {
                            const T out = NEGACC ? -acc[i][j][r] : acc[i][j][r];
                            if (ntc) __builtin_nontemporal_store(out, crow_p + j * 16);
                            else crow_p[j * 16] = out;
                        } else {
                            crow_p[j * 16] = crow_p[j * 16] - acc[i][j][r];
                        }
                    }
                }
            }
        }
    }
    GEMM_STAMP(3);
#if defined(RFLU_GEMM_TRACE)
    if (g.stamps) { __builtin_amdgcn_s_waitcnt(0x0F70); GEMM_STAMP(5); }   // stores acknowledged
#endif
}

// C_FIRST: the accumulators start as the C tile (its read overlaps with the first operand slabs, the epilogue only
// stores): +28 % at K = 256, +8 % at K = 512 and, since the negation moved onto the MFMA (round 3), +3..7 % from K = 1024 up too.
template <typename T, bool C_FIRST>
__global__ void __launch_bounds__(256, 2) gemm_sub_kernel(GemmArgs<T> g)
{
    typedef typename Mfma<T>::acc_t acc_t;
    constexpr int VW = 16 / (int)sizeof(T);  // elements per 16-byte vector
    typedef T vec_t __attribute__((ext_vector_type(VW)));
    // C_FIRST computes c - sum a*b in the accumulators.  Round 2 negated every A element on its way into LDS (8 VALU ops per thread
    // and slab, between the wait for the global loads and the LDS writes of the slab transition).  Float64: the MFMA negates A itself
    // (NEGMOD; sustained 15872^2 x 512: 61.4 -> 64.4 TFLOP/s).  Float32 has no such modifier: the accumulators hold -(c - sum a*b)
    // instead, negated once on the way in and once on the way out (NEGACC; negation is exact, so the results are bit-identical).
    constexpr bool NEGMOD = C_FIRST && Mfma<T>::HAS_NEG;
    constexpr bool NEGACC = C_FIRST && !Mfma<T>::HAS_NEG;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* smem = reinterpret_cast<T*>(smem_raw);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    // ---- workgroup -> tile: bijective XCD-aware remap (block b runs on XCD b % 8; give each XCD a contiguous range
    // of tiles so neighbours share operand panels in that XCD's private L2), then GROUP_M-row grouped ordering.
    int tile_m, tile_n;
    const int n_first = g.na_tiles_n * g.tiles_m;                 // workgroups of the "first columns" region (0: none)
    const bool in_first = (int)blockIdx.x < n_first;
    {
        // two regions, each with the remap of its own: blocks [0, n_first) -> tile columns [0, na), the rest -> [na, tiles_n)
        const int nwg = in_first ? n_first : (int)gridDim.x - n_first;
        const int bid = in_first ? (int)blockIdx.x : (int)blockIdx.x - n_first;
        const int tn = in_first ? g.na_tiles_n : g.tiles_n - g.na_tiles_n;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
        const int per_group = G_GROUP_M * tn;
        const int group = wg / per_group;
        const int first_m = group * G_GROUP_M;
        const int gsz = min(g.tiles_m - first_m, G_GROUP_M);
        const int in_group = wg - group * per_group;
        tile_m = first_m + in_group % gsz;
        tile_n = in_group / gsz + (in_first ? 0 : g.na_tiles_n);
    }
    const int m0 = tile_m * G_BM, n0 = tile_n * G_BN;

    const bool full_mn = g.vec_ok && (m0 + G_BM <= g.M) && (n0 + G_BN <= g.N);
    // INTERIOR (workgroup-uniform): a full tile of an aligned problem whose K is a multiple of the slab depth and at least two
    // slabs -- every bulk update.  Its copy of the tile code has no bounds test and, more important, no branch around a load: the
    // compiler places a wait for "all but the n most recent loads" only where it can COUNT the loads issued since, and behind any
    // join it waits for everything.
    if (full_mn && (g.K % G_BK) == 0 && g.K >= 2 * G_BK) gemm_tile<T, C_FIRST, true>(g, smem, m0, n0);
    else gemm_tile<T, C_FIRST, false>(g, smem, m0, n0);
    if (in_first && g.sig_flag) {   // workgroup-uniform
        __syncthreads();
        if (tid == 0) {
            __threadfence();
            if (atomicInc(g.sig_cnt, (unsigned)n_first - 1) == (unsigned)n_first - 1)
                __hip_atomic_store(g.sig_flag, g.sig_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---- latency-bound updates of the panel recursion: K a small multiple of 64, few tiles -------------------------------------
// The recursion inside a block column issues C -= A*B with K = 64, 128, (256) and N = K on the critical path.  With the
// kernel above such a launch is a chain of K/16 dependent {global load -> LDS -> barrier -> MFMA} rounds (18 / 28 us
// standalone at M = 16384; 2-3x that next to the bulk update that owns most CUs).  Here a workgroup (4 waves) takes a
// 64x64 tile and issues EVERY load of a 64-wide K chunk at once:
//   * A never touches LDS: with the K index permuted as k = 16*kk + s (kk = lane>>4, s = MFMA step), the A operand of
//     lane (i, kk) for the 16 steps of a chunk is 16 CONTIGUOUS elements of row i -- four 16-byte loads;
//   * B (64 x 64 per chunk) goes through LDS once.  Layout Bs[k][c + (c >> 4)], row stride 68 doubles: a staging write has the
//     16 lanes of a group at rows r..r+3 x column blocks q = 0..3 -> dword 136*r + 34*q = 8*r + 2*q (mod 32): 16 distinct bank
//     pairs (the old odd stride 65 put the four column blocks of a row on the same banks: 4-way conflicts, 38 % of the
//     kernel's LDS cycles); a fragment read covers 16 consecutive columns of one block;
//   * the C tile is read into registers up front, so one memory round trip covers everything the tile needs.
// The sum over k is the same set of products; only the order inside the MFMA accumulation differs.
constexpr int S_BM = 64, S_BN = 64, S_KC = 64;
constexpr int S_SB = S_BN + 4;   // 64 columns + one gap double per 16-column block: column c sits at c + (c >> 4)

template <typename T, int NCHUNK>
__global__ void __launch_bounds__(256) gemm_skinny_kernel(GemmArgs<T> g)
{
    typedef typename Mfma<T>::acc_t acc_t;
    constexpr int VW = 16 / (int)sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(VW)));
    __shared__ T Bs[NCHUNK][S_KC * S_SB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile_m = blockIdx.x % g.tiles_m, tile_n = blockIdx.x / g.tiles_m;
    const int m0 = tile_m * S_BM, n0 = tile_n * S_BN;
    const int li = lane & 15, kk = lane >> 4;
    const int arow = m0 + wave * 16 + li;
    const bool arow_ok = arow < g.M;
    const T* Ap = g.A + (int64_t)(arow_ok ? arow : 0) * g.lda + kk * 16;

    // every load of the tile is issued before the first use: A operands (registers), B chunks, the C tile
    T ra[NCHUNK][16];
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
#pragma unroll
        for (int v = 0; v < 16 / VW; ++v) {
            const vec_t x = *reinterpret_cast<const vec_t*>(Ap + c * S_KC + v * VW);
#pragma unroll
            for (int e = 0; e < VW; ++e) ra[c][v * VW + e] = arow_ok ? x[e] : T(0);
        }
    }
    // B chunk: 64 rows x 64 columns, 16 elements per thread (row tid>>2, columns (tid&3)*16 ..)
    const int brow = tid >> 2, bcol = (tid & 3) * 16;
    T rb[NCHUNK][16];
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
        const T* Bp = g.B + (int64_t)(c * S_KC + brow) * g.ldb + n0 + bcol;
        if (n0 + S_BN <= g.N) {
#pragma unroll
            for (int v = 0; v < 16 / VW; ++v) {
                const vec_t x = *reinterpret_cast<const vec_t*>(Bp + v * VW);
#pragma unroll
                for (int e = 0; e < VW; ++e) rb[c][v * VW + e] = x[e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) rb[c][e] = (n0 + bcol + e < g.N) ? Bp[e] : T(0);
        }
    }
    // C tile: for a fixed (j, r) sixteen lanes cover 16 consecutive columns of one row
    T cin[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = m0 + wave * 16 + Mfma<T>::crow(lane, r);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + j * 16 + li;
            cin[j][r] = (row < g.M && col < g.N) ? g.C[(int64_t)row * g.ldc + col] : T(0);
        }
    }
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
#pragma unroll
        for (int e = 0; e < 16; ++e) Bs[c][brow * S_SB + bcol + (bcol >> 4) + e] = rb[c][e];   // bank-conflict free, see below
    }
    __syncthreads();
    acc_t acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = acc_t{0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            const T* brow_p = &Bs[c][(kk * 16 + st) * S_SB + li];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = Mfma<T>::run(ra[c][st], brow_p[j * 17], acc[j]);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = m0 + wave * 16 + Mfma<T>::crow(lane, r);
        if (row < g.M) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = n0 + j * 16 + li;
                if (col < g.N) g.C[(int64_t)row * g.ldc + col] = cin[j][r] - acc[j][r];
            }
        }
    }
}

template <typename T>
int launch_gemm(Handle* h, int64_t M, int64_t N, int64_t K, const T* A, int64_t lda, const T* B, int64_t ldb, T* C,
                int64_t ldc, GemmSignal sig)
{
    if (M <= 0 || N <= 0 || K <= 0) return RFLU_OK;
    GemmArgs<T> g;
    g.na_tiles_n = 0; g.sig_flag = nullptr; g.sig_val = 0; g.sig_cnt = nullptr;
#if defined(RFLU_GEMM_TRACE)
    g.stamps = g_gemm_stamps;
#endif
    g.M = (int)M; g.N = (int)N; g.K = (int)K;
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
    g.tiles_m = (int)((M + G_BM - 1) / G_BM);
    g.tiles_n = (int)((N + G_BN - 1) / G_BN);
    constexpr int VW = 16 / (int)sizeof(T);
    g.vec_ok = ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) % 16 == 0) && (lda % VW == 0) &&
               (ldb % VW == 0);
    g.flags = h->tune.gemm_flags;
    {   // small K, few tiles: the latency-optimised kernel (measured crossover, scripts/microbench_gemm_small.py)
        const int skinny_max_k = h->tune.skinny_max_k, skinny_wide = h->tune.skinny_wide;
        if (g.vec_ok && (K == S_KC || K == 2 * S_KC) && K <= skinny_max_k && (N <= 2 * K || (skinny_wide && K == S_KC))) {
            g.tiles_m = (int)((M + S_BM - 1) / S_BM);
            g.tiles_n = (int)((N + S_BN - 1) / S_BN);
            ProfScope ps(h, RFLU_K_GEMM_SMALL, 2.0 * (double)M * (double)N * (double)K,
                         sizeof(T) * ((double)M * K + (double)K * N + 2.0 * (double)M * N));
            const dim3 grid((unsigned)(g.tiles_m * g.tiles_n));
            if (K == S_KC) hipLaunchKernelGGL((gemm_skinny_kernel<T, 1>), grid, dim3(256), 0, h->stream, g);
            else           hipLaunchKernelGGL((gemm_skinny_kernel<T, 2>), grid, dim3(256), 0, h->stream, g);
            RFLU_HIP(hipGetLastError());
            return RFLU_OK;
        }
    }
    const size_t lds = 2 * (size_t)G_STAGE * sizeof(T);
    bool& attr_set = h->gemm_attr_set[sizeof(T) == 8 ? 0 : 1];   // per handle = per device (not a process-wide static)
    if (!attr_set) {
        RFLU_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_sub_kernel<T, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        RFLU_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_sub_kernel<T, false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    // K < 256: the leaf-wise schedule's per-leaf updates (K = 64) and the recursion's small merges -- HBM/latency-bound launches
    // that are accounted with the small-K kernel, so that RFLU_K_GEMM is the MFMA-bound bulk update alone
    ProfScope ps(h, K < 256 ? RFLU_K_GEMM_SMALL : RFLU_K_GEMM, 2.0 * (double)M * (double)N * (double)K,
                 sizeof(T) * ((double)M * K + (double)K * N + 2.0 * (double)M * N));  // A, B once; C in and out
    if (sig.flag && sig.first_cols > 0) {   // only the tiled kernel knows the two-region order (callers use it for K >= 256)
        g.na_tiles_n = (int)std::min<int64_t>((sig.first_cols + G_BN - 1) / G_BN, g.tiles_n);
        g.sig_flag = sig.flag; g.sig_val = sig.val; g.sig_cnt = sig.cnt;
    }
    const int64_t nwg = (int64_t)g.tiles_m * g.tiles_n;
    // C_FIRST (accumulators start as the C tile) for every K since round 3: with the products' sign on the MFMA it also wins at
    // large K (15360^2 x K, TFLOP/s: K=1024 62.7 -> 67.2, 2048 67.4 -> 69.2, 4096 68.0 -> 69.7; N=65536 3219 -> 3124 ms); round 2
    // switched to the late read-modify-write from K = 1024 on because the per-slab negation cost more there
    // (RFLU_GEMM_CFIRST_BELOW=1024 restores that)
    const int64_t cfirst_below = h->tune.gemm_cfirst_below;
    if (K < cfirst_below) hipLaunchKernelGGL((gemm_sub_kernel<T, true>), dim3((unsigned)nwg), dim3(256), lds, h->stream, g);
    else          hipLaunchKernelGGL((gemm_sub_kernel<T, false>), dim3((unsigned)nwg), dim3(256), lds, h->stream, g);
    RFLU_HIP(hipGetLastError());
    return RFLU_OK;
}

template int launch_gemm<double>(Handle*, int64_t, int64_t, int64_t, const double*, int64_t, const double*, int64_t,
                                 double*, int64_t, GemmSignal);
template int launch_gemm<float>(Handle*, int64_t, int64_t, int64_t, const float*, int64_t, const float*, int64_t, float*,
                                int64_t, GemmSignal);


// ---- clock keeper ----------------------------------------------------------------------------------------------------
// Register-only fp64 MFMA loop, no memory traffic: `wgs` workgroups of 256 threads spin for `iters` x 64 MFMAs per wave.
// MI355X power management lowers the clock while the chip is lightly loaded (a panel on 32 CUs) and takes ~15 ms of
// sustained load to bring it back (scripts/microbench_gemm_seq.py: a 3 ms idle gap costs the next GEMM 11 %); see
// DESIGN.md "clocks".  Operands are lane-dependent non-trivial values: with zeros the power draw -- and the effect -- vanish.
__global__ void __launch_bounds__(256) heat_kernel(int iters, double* sink)
{
    typedef double acc_t __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63;
    double a = 0.37 + 0.001 * lane, b = -0.61 + 0.002 * lane;
    acc_t c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0}, c3 = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, b, c3, 0, 0, 0);
        }
        a = -a;
    }
    const double r = c0[0] + c1[1] + c2[2] + c3[3];
    if (r == 12345.678 && sink) sink[0] = r;   // never true: keeps the loop alive
}

// ~`usec` microseconds of MFMA load on `cus` CUs of the stream h->stream (64 MFMAs x 64 clocks per iteration per wave,
// one wave per SIMD, at a nominal 2.4 GHz)
int launch_heat(Handle* h, int cus, double usec)
{
    const int iters = (int)(usec * 2400.0 / (64.0 * 64.0));
    if (iters <= 0 || cus <= 0) return RFLU_OK;
    hipLaunchKernelGGL(heat_kernel, dim3((unsigned)cus), dim3(256), 0, h->stream, iters, (double*)nullptr);
    RFLU_HIP(hipGetLastError());
    return RFLU_OK;
}

}  // namespace rflu
