// gemm_tile.hpp -- one 128x128 tile of  C <- C - A*B  on the MFMA matrix cores of gfx950: the body shared by the per-update
// launches of gemm.hip (gemm_sub_kernel) and the persistent update engine (engine.hip).  See gemm.hip for the design notes
// (LDS images, bank rules, the order of the C prologue) and DESIGN.md for the measurements.
#pragma once

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
    const bool wt = (g.flags & 4) != 0;
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
                            // flags bit 2: write-through (sc1) stores -- the resident engine publishes a finished tile without an
                            // agent-scope release fence, i.e. without writing back its whole XCD's L2 (engine.hip)
                            if (wt) __hip_atomic_store(crow_p + j * 16, out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            else if (ntc) __builtin_nontemporal_store(out, crow_p + j * 16);
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

}  // namespace rflu
