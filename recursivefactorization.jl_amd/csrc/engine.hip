// engine.hip -- the persistent update engine: every trailing update of the block-column schedule as ONE resident kernel that
// pulls its work from device-side counters.
//
// What it replaces.  schur_complement! (/root/reference/src/lu.jl:265-284), the block-row ldiv! (:235) and apply_permutation!
// (:164-188) of the trailing matrix were three launch sequences per block column on a CU-masked "update stream" (driver.cpp:
// factor_lookahead): interchanges -> block-row solve -> one bulk GEMM whose workgroups take their tiles in a fixed order.  The
// bulk GEMM reached 46-50 TFLOP/s there against 57.7 alone on the same CUs: every launch ramps up and tails off, the stream's
// own interchanges / solves run with the matrix cores idle (7 % of its time), and a block column's update cannot start before the
// previous one has drained although only the SAME columns depend on each other.
//
// Design.  The trailing matrix is a set of COLUMN BLOCKS (W columns).  Column block cb has to receive a fixed list of operations
// in a fixed order (engine.hpp: eng_op) -- BIG(b): block column b as a whole (K = W) for every block column at least two to its
// left, then LEAF(g): the eight leaves of the block column right in front of it and the leaves of its own block column, one at a
// time (K = 64: the leaf-wise schedule's window, driver.cpp: factor_leafwise) -- each in two stages ("sequences", seq = 2 op + stage):
//     stage 0: the operation's interchanges on its columns + X = inv(L11) * A12      (units of 32 columns, one workgroup each)
//     stage 1: A22 -= A21 * X                                                        (units of one 128 x 128 tile)
// and different column blocks are independent of each other.  Per column block a 64-bit claim word (seq << 32 | next unit) and a
// done counter live in device memory; the 2 x (CUs of the mask) resident workgroups loop { scan the claim words (one lane per column
// block) -> pick the eligible unit of highest priority (the leftmost column block: what the chain of leaves needs next) ->
// fetch-and-add on its claim word -> agent-scope acquire -> run the unit -> agent-scope release -> count it done; the last finisher
// of a sequence publishes the next one }.  An operation is eligible once the critical-path stream's leaf counter says that the
// leaves it applies are factored (and their diagonal inverses / move lists written); a column block publishes how many of its
// operations are complete (`prog`, with a half step when a leaf window's first tile column is), which the critical-path stream waits for
// in front of a leaf's lookahead columns.  No workgroup
// ever waits for a particular other workgroup: a unit is claimed when everything it reads is final, so residency is a matter of
// speed, not of correctness.  The interchanges that later leaves owe the FINISHED columns to their left are units of lowest priority.
// Inter-workgroup visibility follows MI355X_MICROARCH.md ("Workgroup dispatch ..."): producer = every wave drains its stores,
// barrier, one lane's agent-scope release fence (buffer_wbl2 sc1) + s_waitcnt, then the counter; consumer = relaxed poll of the
// claim word, one lane's agent-scope acquire (buffer_inv sc1), barrier, plain loads.
//
// Roofline: the stage-1 units are gemm_tile (gemm_tile.hpp), fp64 MFMA bound, 2 * 128 * 128 * W flops each.
#include <limits.h>

#include "engine.hpp"
#include "gemm_tile.hpp"
#include "laswp_strip.hpp"

namespace rflu {

constexpr int EP_COLS = ENG_PREP_COLS;   // columns of a stage-0 unit
constexpr int EP_XLD = EP_COLS + 1;     // LDS row pitch of the staged block: odd -- the four k groups of a fragment read (rows 16 apart) land 16 doubles apart in the banks

template <typename T>
__device__ __forceinline__ int eng_units(const EngArgs<T>& a, int cb, unsigned seq)
{
    return eng_units_of(eng_op(a.g, cb, (int)(seq >> 1)), (int)(seq & 1u), a.g.m);
}

// (the control words live in global memory; said explicitly, because a pointer that reaches a non-inlined function through a structure is a
// generic one, and generic accesses are flat instructions: slower, and counted on the LDS counter as well)
typedef unsigned long long __attribute__((address_space(1))) EngGWord;
__device__ __forceinline__ unsigned long long eng_load(const unsigned long long* p)
{
    return __hip_atomic_load((const EngGWord*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void eng_store(unsigned long long* p, unsigned long long v)
{
    __hip_atomic_store((EngGWord*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long eng_add(unsigned long long* p, unsigned long long v)
{
    return __hip_atomic_fetch_add((EngGWord*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void eng_max(unsigned long long* p, unsigned long long v)
{
    __hip_atomic_fetch_max((EngGWord*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void eng_release()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the compiler may drop the wait behind buffer_wbl2 (MI355X_MICROARCH.md)
}

// ---- stage 0: the operation's interchanges on 32 of its columns, then X = inv(L11) * A12 on those columns ----------------------------
// The solve is trsm_fused_kernel's left-looking walk over 64-row blocks with pre-inverted diagonal blocks (trsm.hip), for a block
// row of any height: the solved blocks X_e are read back from memory (this workgroup's own stores, drained + barrier) instead of
// being kept in LDS, so W = 512 ... 2048 rows cost 24 KB of LDS; a leaf (64 rows) is the loop's first round alone.  Wave w owns
// rows [16w, 16w+16) of a 64-row block x 32 columns.
// The unit functions are real calls (inlined, the three of them spill), and across a call boundary -- or through an argument struct
// whose address has been taken, which then lives in scratch memory -- hipcc no longer knows which address space a pointer belongs to:
// every global access of the first build was a FLAT instruction, 460 in the tile function alone.  Flat loads count on vmcnt AND
// lgkmcnt, so every wait for an LDS read of the GEMM's inner loop was also a wait for the operand slab requested ahead.  The units
// therefore get their pointers by VALUE with the address space in the type (a cast global -> generic at the top of the function is what
// the address-space inference reads; generic -> global -> generic is folded away, an __builtin_assume is ignored).
template <typename T>
struct EngUnit {
    typedef T __attribute__((address_space(1))) GT;
    typedef int __attribute__((address_space(1))) GI;
    typedef T __attribute__((address_space(3))) LT;
    GT* R;
    const GT* linv;
    const GI* pm_cnt;
    const GI* pm_dst;
    const GI* pm_src;
    LT* smem;
    int64_t ld;
    EngGeo g;
    int gemm_flags;
    int write_through;
    int solve_rl;
};
template <typename T>
__device__ __forceinline__ EngUnit<T> eng_unit_args(const EngArgs<T>& a, T* smem)
{
    EngUnit<T> u;
    u.R = (typename EngUnit<T>::GT*)a.R;
    u.linv = (const typename EngUnit<T>::GT*)a.linv;
    u.pm_cnt = (const typename EngUnit<T>::GI*)a.pm_cnt;
    u.pm_dst = (const typename EngUnit<T>::GI*)a.pm_dst;
    u.pm_src = (const typename EngUnit<T>::GI*)a.pm_src;
    u.smem = (typename EngUnit<T>::LT*)smem;
    u.ld = a.ld;
    u.g = a.g;
    u.gemm_flags = a.gemm_flags;
    u.write_through = a.write_through;
    u.solve_rl = a.solve_rl;
    return u;
}

template <typename T>
__device__ __attribute__((noinline)) void eng_prep_unit(const EngUnit<T> a, const EngOp o, int u)
{
    T* const smem = (T*)a.smem;
    typedef typename Mfma<T>::acc_t acc_t;
    constexpr int VW = 16 / (int)sizeof(T);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // (function arguments arrive in vector registers; what is the same in every lane is made scalar again for Float64: the block-row solve below
    // holds its block row in 128 VGPRs and has none to spare for copies of a pitch or a base address -- it reloaded them from scratch some ten times
    // per 64-row step: N=16384 71.5 -> 70.3 ms; Float32 has the registers and measured no faster)
    constexpr bool UNI = sizeof(T) == 8;
    auto uni = [](int v) { return UNI ? __builtin_amdgcn_readfirstlane(v) : v; };
    auto uni64 = [](unsigned long long v) {
        return UNI ? ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)v) : v;
    };
    const int j0 = uni(o.j0), jb = uni(o.jb);
    const int c0 = uni(o.c_lo + u * EP_COLS);
    const int nc = uni(min(EP_COLS, o.nc - u * EP_COLS));
    T* const R = (T*)uni64((unsigned long long)(T*)a.R);
    const int64_t ld = (int64_t)uni64((unsigned long long)a.ld);
    const int* const pm_cnt = (const int*)a.pm_cnt;
    const int* const pm_dst = (const int*)a.pm_dst;
    const int* const pm_src = (const int*)a.pm_src;
    if (a.g.pivot) {
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        if (nc % VW == 0) {
            constexpr int SC = 8 * VW;
            if (wv < (nc + SC - 1) / SC)
                laswp_strip<T, VW, 8>(R, ld, c0, nc, 0, 0, 0, 0, pm_cnt, pm_dst, pm_src, o.chunk0, o.chunk1, wv);
        } else {
            if (wv < (nc + 7) / 8)
                laswp_strip<T, 1, 8>(R, ld, c0, nc, 0, 0, 0, 0, pm_cnt, pm_dst, pm_src, o.chunk0, o.chunk1, wv);
        }
        __syncthreads();   // (every chunk of laswp_strip ends with s_waitcnt vmcnt(0): the rows are in place)
    }
    const T* L = R + (int64_t)j0 * ld + j0;
    const T* Linv = (const T*)uni64((unsigned long long)(const T*)a.linv) + (int64_t)(j0 / NB) * NB * NB;
    T* B = R + (int64_t)j0 * ld + c0;
    const int nblk = (jb + NB - 1) / NB;
    const int fi = lane & 15, fk = lane >> 4;
    const int arow = wave * 16 + fi;
    T* Xd = smem;
    constexpr int EP_MAXBLK = 8;   // block rows of up to 8 x 64 = 512 rows take the right-looking walk
    if (nblk <= EP_MAXBLK && a.solve_rl) {
        // RIGHT-looking over the 64-row blocks, the whole block row in registers (round 6).  The left-looking walk below read the solved
        // blocks X_e back from memory for every later block d -- per block a store, a wait for its acknowledgement, a barrier and a round of
        // dependent loads: ~16 us x 8 blocks for a 512-row block row, the longest unit of the engine and stage 0 of every whole-block-column
        // operation on every column block's sequential list (strips + solves made free are worth 6 ms of 72.7 at N=16384: DESIGN.md section
        // 9).  Here wave w holds ITS 16 rows of all (up to) 8 blocks (8 x 2 accumulators = 128 VGPRs for Float64), X_d goes from the inverse
        // product to LDS and straight into the updates B_e -= L_ed X_d of the blocks below; L's blocks do not depend on X, so their loads are
        // free to run ahead.  Every accumulator receives the same products in the same order as before (for a fixed block: d ascending):
        // bit-identical.
        acc_t acc[EP_MAXBLK][2];
#pragma unroll
        for (int e = 0; e < EP_MAXBLK; ++e) {
            const int rows_e = e < nblk ? min(NB, jb - e * NB) : 0;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wave * 16 + Mfma<T>::crow(lane, r);
                    const int col = t * 16 + fi;
                    acc[e][t][r] = (row < rows_e && col < nc) ? B[(int64_t)(e * NB + row) * ld + col] : T(0);
                }
        }
        // L rows of block e for the update with X_d (negated), and the rows of the inverted diagonal block d: both independent of the
        // solve's results, requested a step ahead of their use
        auto load_l = [&](T (&av)[16], int e, int d) {
            const bool rok = arow < min(NB, jb - e * NB);
            const T* Lp = L + (int64_t)(e * NB + arow) * ld + d * NB + 16 * fk;
            if (rok) {
                typedef T ep_vec __attribute__((ext_vector_type(VW)));
#pragma unroll
                for (int v = 0; v < 16 / VW; ++v) {
                    const ep_vec y = *reinterpret_cast<const ep_vec*>(Lp + v * VW);
#pragma unroll
                    for (int q = 0; q < VW; ++q) av[v * VW + q] = -y[q];
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) av[kk] = T(0);
            }
        };
        auto load_inv = [&](T (&ai)[16], int d) {
            const T* Ip = Linv + (int64_t)d * NB * NB + arow * NB + 16 * fk;
            typedef T ep_vec __attribute__((ext_vector_type(VW)));
#pragma unroll
            for (int v = 0; v < 16 / VW; ++v) {
                const ep_vec x = *reinterpret_cast<const ep_vec*>(Ip + v * VW);
#pragma unroll
                for (int q = 0; q < VW; ++q) ai[v * VW + q] = x[q];
            }
        };
        auto mul_l = [&](const T (&av)[16], acc_t (&c)[2]) {
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                c[0] = Mfma<T>::run(av[kk], Xd[(16 * fk + kk) * EP_XLD + fi], c[0]);
                c[1] = Mfma<T>::run(av[kk], Xd[(16 * fk + kk) * EP_XLD + 16 + fi], c[1]);
            }
        };
        // (Float64: 128 VGPRs of accumulators + two L images leave no room to keep the next inverse rows in flight as well -- requested at the
        // top of their step instead; Float32 has the registers)
        constexpr bool PRE_INV = sizeof(T) == 4;
        T ai[16], avA[16], avB[16];
        if (PRE_INV) load_inv(ai, 0);
#pragma unroll
        for (int d = 0; d < EP_MAXBLK; ++d) {
            if (d < nblk) {   // (workgroup-uniform)
                const int rows_d = min(NB, jb - d * NB);
                if (!PRE_INV) load_inv(ai, d);
                if (d + 1 < nblk) load_l(avA, d + 1, d);   // (the first block below: in flight during the inverse product)
                // stage this wave's rows of block d as a B operand, then X_d = inv(L_dd) * (what the block has become)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) Xd[(wave * 16 + Mfma<T>::crow(lane, r)) * EP_XLD + t * 16 + fi] = acc[d][t][r];
                __syncthreads();
                acc_t x[2] = {acc_t{T(0), T(0), T(0), T(0)}, acc_t{T(0), T(0), T(0), T(0)}};
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) {
                    const T q0 = Xd[(16 * fk + kk) * EP_XLD + fi];
                    const T q1 = Xd[(16 * fk + kk) * EP_XLD + 16 + fi];
                    x[0] = Mfma<T>::run(ai[kk], q0, x[0]);
                    x[1] = Mfma<T>::run(ai[kk], q1, x[1]);
                }
                if (PRE_INV && d + 1 < nblk) load_inv(ai, d + 1);   // (the next diagonal block's inverse: in flight during the updates)
                __syncthreads();   // (everybody has read the staged block: it may become X_d)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = wave * 16 + Mfma<T>::crow(lane, r);
                        const int col = t * 16 + fi;
                        const bool in = row < rows_d && col < nc;
                        if (in) B[(int64_t)(d * NB + row) * ld + col] = x[t][r];
                        Xd[row * EP_XLD + col] = in ? x[t][r] : T(0);
                    }
                if (d + 1 < nblk) {
                    __syncthreads();
                    // the blocks below: B_e -= L_ed * X_d (slot (fk, kk) of the MFMA stands for k = 16 fk + kk: a lane's 16 entries of L are
                    // 128 contiguous bytes of its row).  Two register images of L rows in rotation: block e + 1 is requested before block e is
                    // multiplied (the compiler's wait counts leave the younger request in flight)
#pragma unroll
                    for (int e = d + 1; e < EP_MAXBLK; e += 2) {
                        if (e < nblk) {
                            if (e + 1 < nblk) load_l(avB, e + 1, d);
                            mul_l(avA, acc[e]);
                            if (e + 2 < nblk) load_l(avA, e + 2, d);
                            if (e + 1 < nblk) mul_l(avB, acc[e + 1 < EP_MAXBLK ? e + 1 : e]);
                        }
                    }
                    __syncthreads();   // (X_d has been read by everybody: the next block may be staged)
                }
            }
        }
        return;
    }
    // (block rows of more than 512 rows: the left-looking walk, X_e read back from memory)
    for (int d = 0; d < nblk; ++d) {
        const int rows_d = min(NB, jb - d * NB);
        acc_t acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wave * 16 + Mfma<T>::crow(lane, r);
                const int col = t * 16 + fi;
                acc[t][r] = (row < rows_d && col < nc) ? B[(int64_t)(d * NB + row) * ld + col] : T(0);
            }
        const bool rok = arow < rows_d;
        for (int e = 0; e < d; ++e) {
            // (slot (fk, kk) of the MFMA stands for k = 16 fk + kk: a lane's 16 entries of L are 128 contiguous bytes of its row -- 16-byte
            // loads -- where k = 4 kk + fk made them sixteen predicated 8-byte loads; the strips + solves were 10 % of the engine's time)
            T av[16], b0[16], b1[16];
            const T* Lp = L + (int64_t)(d * NB + arow) * ld + e * NB + 16 * fk;
            const T* Xe = B + (int64_t)(e * NB + 16 * fk) * ld + fi;
            if (rok) {
                typedef T ep_vec __attribute__((ext_vector_type(VW)));
#pragma unroll
                for (int v = 0; v < 16 / VW; ++v) {
                    const ep_vec x = *reinterpret_cast<const ep_vec*>(Lp + v * VW);
#pragma unroll
                    for (int q = 0; q < VW; ++q) av[v * VW + q] = -x[q];
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) av[kk] = T(0);
            }
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                b0[kk] = fi < nc ? Xe[(int64_t)kk * ld] : T(0);
                b1[kk] = 16 + fi < nc ? Xe[(int64_t)kk * ld + 16] : T(0);
            }
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                acc[0] = Mfma<T>::run(av[kk], b0[kk], acc[0]);
                acc[1] = Mfma<T>::run(av[kk], b1[kk], acc[1]);
            }
        }
        // stage acc as a B operand, then X_d = inv(L_dd) * acc
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) Xd[(wave * 16 + Mfma<T>::crow(lane, r)) * EP_XLD + t * 16 + fi] = acc[t][r];
        T ai[16];
        {
            const T* Ip = Linv + (int64_t)d * NB * NB + arow * NB + 16 * fk;
            typedef T ep_vec __attribute__((ext_vector_type(VW)));
#pragma unroll
            for (int v = 0; v < 16 / VW; ++v) {
                const ep_vec x = *reinterpret_cast<const ep_vec*>(Ip + v * VW);
#pragma unroll
                for (int q = 0; q < VW; ++q) ai[v * VW + q] = x[q];
            }
        }
        __syncthreads();
        acc_t x[2] = {acc_t{T(0), T(0), T(0), T(0)}, acc_t{T(0), T(0), T(0), T(0)}};
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const T q0 = Xd[(16 * fk + kk) * EP_XLD + fi];
            const T q1 = Xd[(16 * fk + kk) * EP_XLD + 16 + fi];
            x[0] = Mfma<T>::run(ai[kk], q0, x[0]);
            x[1] = Mfma<T>::run(ai[kk], q1, x[1]);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wave * 16 + Mfma<T>::crow(lane, r);
                const int col = t * 16 + fi;
                if (row < rows_d && col < nc) B[(int64_t)(d * NB + row) * ld + col] = x[t][r];
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // X_d is in memory before any wave of this workgroup reads it back
        __syncthreads();                                     // (and everybody has read the staged block)
    }
}

// ---- stage 1: one 128 x 128 tile of A22 -= A21 * X --------------------------------------------------------------------------------
template <typename T>
__device__ __attribute__((noinline)) void eng_gemm_unit(const EngUnit<T> a, const EngOp o, int t)
{
    constexpr int VW = 16 / (int)sizeof(T);
    T* const smem = (T*)a.smem;
    // (what is the same in every lane made scalar again, as in eng_prep_unit: the tile code then computes its addresses and loop conditions on
    // the scalar unit, as it does inside gemm_sub_kernel, whose arguments are kernel arguments)
    auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    auto uni64 = [](unsigned long long v) {
        return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)v);
    };
    T* const Rg = (T*)uni64((unsigned long long)(T*)a.R);
    const int64_t ld = (int64_t)uni64((unsigned long long)a.ld);
    const int oj0 = uni(o.j0), ojb = uni(o.jb), oclo = uni(o.c_lo);
    t = uni(t);
    const int je = oj0 + ojb;
    GemmArgs<T> g;
    g.M = uni(a.g.m) - je;
    g.N = uni(o.nc);
    g.K = ojb;
    g.A = Rg + (int64_t)je * ld + oj0;
    g.B = Rg + (int64_t)oj0 * ld + oclo;
    g.C = Rg + (int64_t)je * ld + oclo;
    g.lda = g.ldb = g.ldc = ld;
    g.tiles_m = (g.M + G_BM - 1) / G_BM;
    g.tiles_n = (g.N + G_BN - 1) / G_BN;
    g.vec_ok = (reinterpret_cast<uintptr_t>(Rg) % 16 == 0) && (ld % VW == 0) && (oclo % VW == 0) && (oj0 % VW == 0);
    g.flags = uni(a.gemm_flags | (a.write_through ? 4 : 0));   // write-through C stores, no release fence behind a tile
    g.na_tiles_n = 0; g.sig_flag = nullptr; g.sig_val = 0; g.sig_cnt = nullptr;
    // G_GROUP_M tile rows are walked together, column after column: consecutive claims share the B panel, then the A panels
    int tile_m, tile_n;
    if (uni(o.type) == ENG_OP_LEAF) {
        // a leaf's window, first tile column first: the critical-path stream waits for exactly those columns (the next leaf's
        // lookahead strip is the leftmost of the window) and is told when they are complete, not when the whole window is
        tile_n = t / g.tiles_m;
        tile_m = t - tile_n * g.tiles_m;
    } else {
        const int per_group = G_GROUP_M * g.tiles_n;
        const int group = t / per_group;
        const int first_m = group * G_GROUP_M;
        const int gsz = min(g.tiles_m - first_m, G_GROUP_M);
        const int in_group = t - group * per_group;
        tile_m = first_m + in_group % gsz;
        tile_n = in_group / gsz;
    }
    const int m0 = tile_m * G_BM, n0 = tile_n * G_BN;
    const bool full_mn = g.vec_ok && (m0 + G_BM <= g.M) && (n0 + G_BN <= g.N);
    if (full_mn && (g.K % G_BK) == 0 && g.K >= 2 * G_BK) gemm_tile<T, true, true>(g, smem, m0, n0);
    else gemm_tile<T, true, false>(g, smem, m0, n0);
}

// ---- deferred interchanges on the finished column block cb.  Left op 0: unit u = the 64 columns of the block column's leaf u, which
// still owe the interchanges of the leaves behind it in the same block column; left op lk >= 1: block column cb + lk as a whole, four
// wave strips (of one 128-byte line per row) per unit
template <typename T>
__device__ __attribute__((noinline)) void eng_left_unit(const EngUnit<T> a, int cb, int lk, int u)
{
    constexpr int VW = 16 / (int)sizeof(T);
    constexpr int SC = 8 * VW;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int LPB = a.g.W / NB;
    const int pb = eng_pb(a.g, cb);
    const int cb0 = cb * a.g.Wc;
    const int ncb_cols = min(a.g.Wc, a.g.n - cb0);
    int c0, nc, chunk0, chunk1;
    if (lk == 0) {
        c0 = cb0 + u * NB;
        nc = min(NB, ncb_cols - u * NB);
        chunk0 = c0 / NB + 1;   // (the strip is the columns of leaf c0 / NB: it owes the leaves behind it in its block column)
        chunk1 = pb * LPB + eng_leaves_of_block(a.g, pb);
    } else {
        c0 = cb0 + u * 4 * SC;
        nc = min(4 * SC, ncb_cols - u * 4 * SC);
        chunk0 = (pb + lk) * LPB;
        chunk1 = chunk0 + eng_leaves_of_block(a.g, pb + lk);
    }
    if (nc <= 0 || chunk1 <= chunk0) return;
    T* const R = (T*)a.R;
    const int* const pm_cnt = (const int*)a.pm_cnt;
    const int* const pm_dst = (const int*)a.pm_dst;
    const int* const pm_src = (const int*)a.pm_src;
    if (nc % VW == 0) {
        for (int s = wave; s < (nc + SC - 1) / SC; s += 4)
            laswp_strip<T, VW, 8>(R, a.ld, c0, nc, 0, 0, 0, 0, pm_cnt, pm_dst, pm_src, chunk0, chunk1, s);
    } else {
        for (int s = wave; s < (nc + 7) / 8; s += 4)
            laswp_strip<T, 1, 8>(R, a.ld, c0, nc, 0, 0, 0, 0, pm_cnt, pm_dst, pm_src, chunk0, chunk1, s);
    }
}

enum { ENG_NONE = 0, ENG_MAIN = 1, ENG_LEFT = 2, ENG_EXIT = 3 };

// LDS words shared by the kernel and its scan function (static, in front of the dynamic tile buffers)
__shared__ int eng_sel[4];        // what wave 0 selected: kind, column block, sequence, unit
__shared__ int eng_wst[4];        // wave 0's own state between two scans: cb_lo, fin_b, column block and sequence of its last unit
__shared__ long long eng_nap;     // (accounting instantiation) time asleep with nothing eligible
__shared__ long long eng_ph[8];   // (accounting instantiation) the scan, phase by phase: [0] epoch / gate sample, [1] exit words, [2] claim words + choice, [3] ticket
                                  // (+ waiting for a later stage's leaves), [4] scan of the deferred interchanges, [5] acquire + hand-over, [6] sweeps, [7] calls
#define ENG_PH(k) do { if (TRACE && lane == 0) { const long long t_ = wall_clock64(); eng_ph[k] += t_ - ph_t; ph_t = t_; } } while (0)

// ---- the scheduler: wave 0 of a workgroup looks for its next unit and takes it ------------------------------------------------------------
// A function of its own, NOT inlined (round 6): inside the kernel its values lived across the calls of the unit functions (256 VGPRs each), and
// the register allocator answered with scratch reloads all along the scan -- every one of them a trip to memory behind a tile that has just
// streamed the caches empty, on the path every unit of every stage waits for.  Here it has the register file to itself; the arguments are read
// from the kernel's argument segment (scalar loads), the state it keeps between two calls sits in four LDS words.
template <typename T, bool TRACE>
__device__ __attribute__((noinline)) void eng_scan(unsigned long long kernarg)
{
    // (the kernel's argument segment; the address arrives in a vector register pair like every function argument: made scalar again so that
    // the fields are read with scalar loads)
    const unsigned long long ku = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(kernarg >> 32)) << 32) |
                                  (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)kernarg);
    EngArgs<T> a;
    {
        static_assert(sizeof(EngArgs<T>) % 4 == 0, "copied word by word");
        const __attribute__((address_space(4))) unsigned* kp = (const __attribute__((address_space(4))) unsigned*)ku;
        unsigned* ad = reinterpret_cast<unsigned*>(&a);
#pragma unroll
        for (int i = 0; i < (int)(sizeof(EngArgs<T>) / 4); ++i) ad[i] = kp[i];
    }
    EngState* const st = a.st;
    const int lane = threadIdx.x & 63;
    int cb_lo = eng_wst[0];          // column blocks in front of this one have nothing left for the main scan (monotone)
    int fin_b = eng_wst[1];          // workgroup 0: block rows [0, fin_b) have been reported final (EngArgs::rows_final)
    const int last_cb = eng_wst[2];  // the stage this workgroup's last unit belonged to (-1: none)
    const unsigned last_seq = (unsigned)eng_wst[3];
    unsigned my_xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my_xcc));
    my_xcc &= 7u;
    const bool may_retire = a.retire_leaf > 0 && (int)my_xcc == a.retire_xcc;
    // The workgroups that serve the leaf windows only (K = 64: what the chain of leaves waits for finds a free workgroup at once instead of
    // queueing behind 127-us tiles of the block-column updates): those with blockIdx % 8 in [1, leaf_xcds] and blockIdx / 8 < leaf_wgs --
    // blockIdx % 8 is the XCD a workgroup of the first dispatch round lands on, so by default 16 workgroups of one XCD (N=16384: 97 -> 84 ms)
    const bool leaf_only = a.leaf_xcds > 0 && (int)(blockIdx.x & 7) >= 1 && (int)(blockIdx.x & 7) <= a.leaf_xcds && (int)(blockIdx.x >> 3) < a.leaf_wgs;

    // every LEAF op of the block column pb is complete: on its own column blocks (they have nothing else left) and on those of the
    // `ahead` block columns to its right
    // (`but_last`: every LEAF op but those of the block column's LAST leaf -- what the block column's own deferred interchanges wait for: they
    // move rows in the columns of the earlier leaves, which the last leaf's windows do not read, and with that they, and BIG(pb) on the column
    // block the chain needs next, start one leaf window earlier: the window of the last leaf on the block columns to the right)
    auto leaf_ops_complete = [&](int pb, bool but_last) -> bool {
        const int c0 = eng_first_cb(a.g, pb), n0 = eng_cbs_of_block(a.g, pb);
        for (int c = c0; c < c0 + n0; ++c)
            if ((unsigned)(eng_load(&st->cb[c].claim) >> 32) != ENG_SEQ_DONE) return false;   // (the own column blocks have no LEAF op of the last leaf)
        const int skip = but_last && eng_leaves_of_block(a.g, pb) > 1 ? 1 : 0;
        for (int q = pb + 1; q <= pb + eng_ahead(a.g); ++q) {
            const int c1 = eng_first_cb(a.g, q), n1 = eng_cbs_of_block(a.g, q);
            for (int c = c1; c < c1 + n1; ++c)
                if ((int)eng_load(&st->cb[c].prog) < 2 * (eng_ops_through_block(a.g, c, pb) - skip)) return false;
        }
        return true;
    };
    auto left_op_ok = [&](int cb, int lk) -> bool {
        const int pb = eng_pb(a.g, cb);
        if (lk == 0) return leaf_ops_complete(pb, true);
        return (int)eng_load(&st->cb[eng_first_cb(a.g, pb)].bigdone) >= eng_big_users(a.g, pb);   // nobody reads this block column's L any more
    };
    auto leaves_done = [&]() -> int {
        const unsigned long long v = eng_load(a.leaf_gate);
        return v > a.gate_base ? (int)(v - a.gate_base) : 0;
    };

    int kind = ENG_NONE, sel_cb = 0, sel_unit = 0;
    unsigned sel_seq = 0;
    long long ph_t = TRACE ? wall_clock64() : 0;
    if (TRACE && lane == 0) eng_ph[7] += 1;
    // ---- host entry: how many rows of the factors are final (workgroup 0 looks once per round).  Block row b is final when
    // nothing will write to it any more: every operation on column block b is complete, the leaves of block column b have
    // reached column block b + 1, BIG(b) is complete everywhere, and the interchanges of block column b have reached its own
    // columns and every column block to its left (later block columns only move rows below)
    if (a.rows_final && blockIdx.x == 0 && fin_b < a.g.nbp) {
        for (;;) {
            const int b = fin_b;
            if (b >= a.g.nbp) break;
            const int fb = eng_first_cb(a.g, b);
            bool fin = leaves_done() >= b * (a.g.W / NB) + eng_leaves_of_block(a.g, b) && leaf_ops_complete(b, false) &&
                       (int)eng_load(&st->cb[fb].bigdone) >= eng_big_users(a.g, b);
            if (fin && a.g.pivot) {
                const unsigned long long ld = eng_load(&st->cb[fb].leftdone);
                fin = (int)(ld & 0xffffffffull) >= fb && (!eng_big_waits_for_left(a.g, b) || (int)(ld >> 32) >= eng_cbs_of_block(a.g, b));
            }
            if (!fin) break;
            fin_b = b + 1;
            if (lane == 0) {
                const int rows = fin_b >= a.g.nbp && a.g.nbp * a.g.W >= a.g.mn ? a.g.m : min(fin_b * a.g.W, a.g.m);
                // (the last block row takes the rows below the square part along: nothing writes them after the last panel)
                __hip_atomic_store(a.rows_final, (unsigned long long)rows, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    // a ticket (the value a fetch-and-add on column block cb's claim word returned): this workgroup's unit if the unit exists
    auto accept = [&](int cb, unsigned long long w) {
        const unsigned seq = (unsigned)(w >> 32), u = (unsigned)w;
        if (seq == ENG_SEQ_DONE) return;
        const EngOp o = eng_op(a.g, cb, (int)(seq >> 1));
        if ((int)u >= eng_units_of(o, (int)(seq & 1u), a.g.m)) return;   // (past the end: nobody's)
        kind = ENG_MAIN; sel_cb = cb; sel_seq = seq; sel_unit = (int)u;
        // The scan saw an eligible sequence; the add may have landed in a LATER one (published in between).  A later
        // stage 1 is ready by construction; a later stage 0 needs its leaves -- practically never the case (a whole
        // sequence would have to complete between this wave's scan and its add), but then the claim is held until
        // they are there (bounded; the leaves do not depend on this workgroup)
        if ((seq & 1u) == 0) {
            const long long t0 = wall_clock64();
            const bool wl = o.type == ENG_OP_BIG && eng_big_waits_for_left(a.g, (int)(seq >> 1));
            while ((leaves_done() < o.need ||
                    (wl && (int)(eng_load(&st->cb[eng_first_cb(a.g, (int)(seq >> 1))].leftdone) >> 32) < eng_cbs_of_block(a.g, (int)(seq >> 1)))) &&
                   eng_load(&st->abort) == 0) {
                __builtin_amdgcn_s_sleep(16);
                if (wall_clock64() - t0 > 400000000LL) break;   // (the idle timeout of the others raises the flag)
            }
        }
    };
    for (int attempt = 0; kind == ENG_NONE; ++attempt) {
        // what this sweep is based on: if it finds nothing, the wave sleeps on these three words until one of them moves
        const unsigned long long seen_epoch = eng_load(&st->epoch), seen_gate = eng_load(a.leaf_gate);
        const unsigned long long seen_arr = a.arrived ? eng_load(a.arrived) : 0ull;
        // (the three words must have been SAMPLED before anything the sweep looks at is requested: loads to different channels
        // are served in any order, and a sweep older than the epoch it is paired with sleeps through the publication it missed.
        // That includes the count of unfinished column blocks: read in front of the epoch, the last publication of a
        // factorization could fall between the two reads -- the workgroup then slept on the final epoch of a finished
        // factorization and raised the timeout flag 4 s later: one call in ~200 at N=8192, scripts/engine_stress.py)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (TRACE && lane == 0) eng_ph[6] += 1;
        ENG_PH(0);
        if (eng_load(&st->abort) != 0 || eng_load(&st->remaining) == 0) { kind = ENG_EXIT; break; }
        ENG_PH(1);
        const int pd = seen_gate > a.gate_base ? (int)(seen_gate - a.gate_base) : 0;
        const int have = a.arrived ? (int)seen_arr : a.g.n;   // columns in place (host entry: they arrive while we run)
        // the chain has reached its short panels: the workgroups on ITS XCD go home (one count each, the critical path waits for the sum)
        if (may_retire && pd >= a.retire_leaf) {
            if (lane == 0) __hip_atomic_fetch_add(&st->retired, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            kind = ENG_EXIT;
            break;
        }
        // ---- main units: one lane per column block -----------------------------------------------------------------
        int best = INT_MAX;
        int best2 = INT_MAX, best3 = INT_MAX;   // the runners-up of the chunk the best was found in: tried before a new sweep when the ticket comes back empty
        int first_live = INT_MAX;
        for (int base = cb_lo & ~63; base < a.g.ncb; base += 64) {
            const int cb = base + lane;
            int key = INT_MAX;
            bool live = false;
            if (cb < a.g.ncb) {
                const unsigned long long w = eng_load(&st->cb[cb].claim);
                const unsigned seq = (unsigned)(w >> 32), u = (unsigned)w;
                if (seq != ENG_SEQ_DONE) {
                    live = true;
                    const EngOp o = eng_op(a.g, cb, (int)(seq >> 1));
                    bool ok = o.need <= pd && (int)u < eng_units_of(o, (int)(seq & 1u), a.g.m) && min(a.g.n, (cb + 1) * a.g.Wc) <= have;
                    if (leaf_only && o.type == ENG_OP_BIG) ok = false;
                    // a block column as a whole is applied with ITS interchanges complete on all its columns (engine.hpp)
                    if (ok && o.type == ENG_OP_BIG && (seq & 1u) == 0 && eng_big_waits_for_left(a.g, (int)(seq >> 1)))
                        ok = (int)(eng_load(&st->cb[eng_first_cb(a.g, (int)(seq >> 1))].leftdone) >> 32) >= eng_cbs_of_block(a.g, (int)(seq >> 1));
                    if (ok) {
                        key = a.policy ? (((o.j0 / NB) << 10) | cb) : ((1 << 24) | cb);
                        // host entry: a block row can leave only when the panel has reached EVERY column block; with the
                        // leftmost-first rule alone the far right is served last and the way back starts when the
                        // factorization ends.  A whole-block-column operation that lags the chain by `host_lag` block columns
                        // or more goes first, oldest panel first.
                        if (a.host_lag > 0 && o.type == ENG_OP_BIG && pd / (a.g.W / NB) - (int)(seq >> 1) >= a.host_lag)
                            key = ((int)(seq >> 1) << 10) | cb;
                    }
                }
            }
            const unsigned long long lv = __ballot(live);
            if (lv != 0 && first_live == INT_MAX) first_live = base + __ffsll((long long)lv) - 1;
            int mk = key;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mk = min(mk, __shfl_xor(mk, o));
            if (mk < best) {
                best = mk;
                // (keys are distinct: they carry the column block)
                int k2 = key > best ? key : INT_MAX;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) k2 = min(k2, __shfl_xor(k2, o));
                int k3 = key > k2 ? key : INT_MAX;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) k3 = min(k3, __shfl_xor(k3, o));
                best2 = k2; best3 = k3;
            }
            if (a.policy == 0 && a.host_lag <= 0 && best != INT_MAX) break;   // leftmost first: nothing further right can beat it
        }
        if (first_live != INT_MAX) cb_lo = first_live;
        ENG_PH(2);
        if (best != INT_MAX) {
            // fetch-and-add, not compare-and-swap: with a few hundred workgroups arriving together a CAS hands out ONE unit per
            // round trip (the losers rescan: 384 tiles took 0.6 ms to hand out).  Whatever (sequence, unit) the add returns is
            // this workgroup's if the unit exists; a count past the end is nobody's (the next sequence starts from zero again).
            // When a stage of a few units is published (16 block-row solve units, say) every workgroup that comes free in those microseconds
            // picks it, most tickets come back empty, and each loser used to sweep again: the next-best units of the same sweep are tried first
            // (their eligibility is as old as the best one's was; whatever the add returns is validated as before).
            int cand = best;
            for (int tries = 0; tries < 3 && cand != INT_MAX && kind == ENG_NONE; ++tries) {
                const int cb = cand & 1023;
                unsigned long long w = 0;
                if (lane == 0) w = eng_add(&st->cb[cb].claim, 1ull);
                w = __shfl(w, 0);
                accept(cb, w);
                cand = tries == 0 ? best2 : best3;
            }
            ENG_PH(3);
            continue;   // (past the end: look again)
        }
        // ---- deferred interchanges on the finished column blocks ---------------------------------------------------
        if (a.g.pivot) {
            int bestl = INT_MAX;
            const int nleftcb = min(a.g.ncb, eng_first_cb(a.g, a.g.nbp));   // column blocks of the served block columns
            for (int base = 0; base < nleftcb; base += 64) {
                const int cb = base + lane;
                int key = INT_MAX;
                if (cb < nleftcb) {
                    const unsigned long long w = eng_load(&st->cb[cb].lclaim);
                    const unsigned lk = (unsigned)(w >> 32), u = (unsigned)w;
                    if (lk != ENG_SEQ_DONE && eng_left_need(a.g, cb, (int)lk) <= pd && (int)u < eng_left_units<T>(a.g, cb, (int)lk)) {
                        if (left_op_ok(cb, (int)lk)) key = ((int)lk << 10) | cb;
                    }
                }
                int mk = key;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) mk = min(mk, __shfl_xor(mk, o));
                bestl = min(bestl, mk);
            }
            ENG_PH(4);
            if (bestl != INT_MAX) {
                const int cb = bestl & 1023;
                unsigned long long w = 0;
                if (lane == 0) w = eng_add(&st->cb[cb].lclaim, 1ull);
                w = __shfl(w, 0);
                const unsigned lk = (unsigned)(w >> 32), u = (unsigned)w;
                if (lk != ENG_SEQ_DONE && (int)u < eng_left_units<T>(a.g, cb, (int)lk)) {
                    kind = ENG_LEFT; sel_cb = cb; sel_seq = lk; sel_unit = (int)u;
                    const long long t0 = wall_clock64();
                    // (a later left op than the one the scan saw: its own conditions, see the scan)
                    auto left_ok = [&]() -> bool {
                        return leaves_done() >= eng_left_need(a.g, cb, (int)lk) && left_op_ok(cb, (int)lk);
                    };
                    while (!left_ok() && eng_load(&st->abort) == 0) {
                        __builtin_amdgcn_s_sleep(16);
                        if (wall_clock64() - t0 > 400000000LL) break;
                    }
                }
                continue;
            }
        }
        // nothing is eligible right now: wait (one lane's worth of loads per round, not a sweep) until the critical path or
        // the engine itself has published something since this sweep began
        {
            const long long t0 = wall_clock64();
            bool gave_up = false;
            int naps = 0;
            for (;;) {
                if (eng_load(&st->epoch) != seen_epoch || eng_load(a.leaf_gate) != seen_gate ||
                    (a.arrived && eng_load(a.arrived) != seen_arr) || eng_load(&st->abort) != 0 || eng_load(&st->remaining) == 0)
                    break;
                naps = min(naps + 1, leaf_only ? 2 : 16);
                for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(32);
                if (wall_clock64() - t0 > 400000000LL) { gave_up = true; break; }   // 4 s without any news: something upstream is stuck
            }
            if (TRACE && lane == 0) eng_nap += wall_clock64() - t0;
            if (TRACE) ph_t = wall_clock64();
            if (gave_up) {
                if (lane == 0) {
                    __hip_atomic_fetch_or((unsigned long long*)(a.info + 1), 17ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    eng_store(&st->abort, 1ull);
                }
                kind = ENG_EXIT;
                break;
            }
        }
    }
    if (lane == 0) {
        // the acquire (an L1 invalidate) once per (column block, sequence): everything a tile of a stage reads was final when the
        // stage was published, and the tiles other workgroups write during it are not read by this one -- the second and later
        // tiles of the same stage keep their L1 (the U12 strip they share) and save the fence
        const bool same_stage = kind == ENG_MAIN && (sel_seq & 1u) != 0 && sel_cb == last_cb && sel_seq == last_seq;
        if ((kind == ENG_MAIN || kind == ENG_LEFT) && !same_stage) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        eng_sel[0] = kind; eng_sel[1] = sel_cb; eng_sel[2] = (int)sel_seq; eng_sel[3] = sel_unit;
    }
    ENG_PH(5);
    if (lane == 0) {   // (wave 0's state between two scans)
        eng_wst[0] = cb_lo; eng_wst[1] = fin_b;
        eng_wst[2] = kind == ENG_MAIN ? sel_cb : -1; eng_wst[3] = (int)sel_seq;
    }
}

// ---- a unit is finished: count it, and if it was the last of its stage, publish the next one.  One lane; a function of its own for the same
// reason as the scan (the path from the end of a stage's last unit to the publication of the next stage is what every column block's sequence
// is made of)
template <typename T, bool TRACE>
__device__ __attribute__((noinline)) void eng_complete(unsigned long long kernarg, int kind, int cb, unsigned seq, int unit)
{
    const unsigned long long ku = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(kernarg >> 32)) << 32) |
                                  (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)kernarg);
    EngArgs<T> a;
    {
        const __attribute__((address_space(4))) unsigned* kp = (const __attribute__((address_space(4))) unsigned*)ku;
        unsigned* ad = reinterpret_cast<unsigned*>(&a);
#pragma unroll
        for (int i = 0; i < (int)(sizeof(EngArgs<T>) / 4); ++i) ad[i] = kp[i];
    }
    EngState* const st = a.st;
    kind = __builtin_amdgcn_readfirstlane(kind); cb = __builtin_amdgcn_readfirstlane(cb);
    seq = (unsigned)__builtin_amdgcn_readfirstlane((int)seq); unit = __builtin_amdgcn_readfirstlane(unit);
    // a Schur tile stored write-through has nothing left in this XCD's L2 (its stores are acknowledged: s_waitcnt above)
    if (!(a.write_through && kind == ENG_MAIN && (seq & 1u) != 0)) eng_release();
    EngCB* c = &st->cb[cb];
    if (kind == ENG_MAIN) {
        const int units = eng_units(a, cb, seq);
        // done = finished units (low word) | finished units of a leaf window's FIRST tile column (high word)
        const EngOp o = eng_op(a.g, cb, (int)(seq >> 1));
        const int tiles_m = (a.g.m - (o.j0 + o.jb) + G_BM - 1) / G_BM;
        const bool first_col = (seq & 1u) != 0 && o.type == ENG_OP_LEAF && unit < tiles_m;
        const unsigned long long dd = eng_add(&c->done, 1ull + (first_col ? 1ull << 32 : 0ull)) +
                                      1ull + (first_col ? 1ull << 32 : 0ull);
        const unsigned long long d = dd & 0xffffffffull;
        // Publications carry no release fence (round 6: each used to start with an ACQ_REL fence and publish through four RELEASE
        // atomics -- an L2 write-back of this XCD in front of every one of them, on the path every stage of every column block's
        // sequence goes through).  None is needed: whatever a unit wrote is at the memory side BEFORE its count (tiles are stored
        // write-through and acknowledged, strips are released by their own workgroup), the workgroup that sees the last count only
        // announces it, and the consumers take their units with an acquire (same box, alternating: N=12288 through the engine 45.2 -> 44.0
        // ms, Float32 N=16384 55.3 -> 54.9, Float64 N=16384 unchanged).  What has to hold is the ORDER of the publisher's own words:
        // the counter reads zero before the next sequence can be counted, and the epoch moves after everything it announces.
        if (first_col && (int)(dd >> 32) == tiles_m && (int)d != units) {
            // the window's first tile column is complete: the critical path may go on (prog = 2 * completed ops + 1).  A maximum, not a
            // store: the workgroup that completes the whole sequence may publish 2 * (op + 1) BEFORE this one, delayed between its
            // count and this line, gets here -- a plain store would take the word back to 2 * op + 1 for good
            eng_max(&c->prog, 2ull * (unsigned long long)(seq >> 1) + 1ull);
        }
        if ((int)d == units) {
            if (TRACE && o.type == ENG_OP_LEAF && cb == (o.j0 + o.jb + NB) / a.g.Wc)
                a.trace[(o.j0 / NB) * 4 + ((seq & 1u) ? 3 : 1)] = wall_clock64();
            if (TRACE && o.type == ENG_OP_LEAF && cb == eng_first_cb(a.g, o.j0 / a.g.W + 1) && o.j0 / NB < 2048)
                a.trace[(2048 + o.j0 / NB) * 4 + ((seq & 1u) ? 3 : 1)] = wall_clock64();
            if (TRACE && o.type == ENG_OP_BIG && cb == eng_first_cb(a.g, o.j0 / a.g.W + eng_ahead(a.g) + 1) && o.j0 / a.g.W < 512)
                a.trace[(1024 + o.j0 / a.g.W) * 4 + ((seq & 1u) ? 3 : 1)] = wall_clock64();
            eng_store(&c->done, 0ull);
            const unsigned end = 2u * (unsigned)eng_nops(a.g, cb);
            unsigned ns = seq + 1;
            while (ns < end && eng_units(a, cb, ns) == 0) ++ns;   // (an operation with no columns left, a panel with no rows below it)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if ((ns >> 1) != (seq >> 1)) {
                for (unsigned k = seq >> 1; k < (ns >> 1); ++k)   // (the operations just completed, skipped ones included)
                    if ((int)k < eng_nbig(a.g, cb))
                        eng_add(&st->cb[eng_first_cb(a.g, (int)k)].bigdone, 1ull);
                eng_max(&c->prog, 2ull * (unsigned long long)(ns >> 1));
            }
            if (ns >= end) {
                eng_store(&c->claim, (unsigned long long)ENG_SEQ_DONE << 32);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (whoever reads remaining == 0 finds every claim word closed)
                eng_add(&st->remaining, ~0ull);
            } else {
                eng_store(&c->claim, (unsigned long long)ns << 32);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            eng_add(&st->epoch, 1ull);
        }
    } else {
        const int units = eng_left_units<T>(a.g, cb, (int)seq);
        const unsigned long long d = eng_add(&c->ldone, 1ull) + 1;
        if ((int)d == units) {
            if (TRACE && seq == 0 && cb == eng_first_cb(a.g, eng_pb(a.g, cb)) && eng_pb(a.g, cb) < 512)
                a.trace[(1536 + eng_pb(a.g, cb)) * 4 + 1] = wall_clock64();
            eng_store(&c->ldone, 0ull);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int nleft = eng_nleft(a.g, cb);
            int nk = (int)seq + 1;
            while (nk < nleft && eng_left_units<T>(a.g, cb, nk) == 0) ++nk;
            const int pbl = eng_pb(a.g, cb);
            for (int k = (int)seq; k < nk; ++k)   // the left ops just completed (k = 0: this block column's own interchanges; k >= 1:
                eng_add(&st->cb[eng_first_cb(a.g, pbl + k)].leftdone, k == 0 ? 1ull << 32 : 1ull);   // block column pbl + k has reached it)
            eng_store(&c->lprog, (unsigned long long)nk);
            if (nk >= nleft) {
                eng_store(&c->lclaim, (unsigned long long)ENG_SEQ_DONE << 32);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                eng_add(&st->remaining, ~0ull);
            } else {
                eng_store(&c->lclaim, (unsigned long long)nk << 32);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            eng_add(&st->epoch, 1ull);
        }
    }
}

template <typename T, bool TRACE>
__global__ void __launch_bounds__(256, 2) engine_kernel(EngArgs<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char eng_smem_raw[];
    T* smem = reinterpret_cast<T*>(eng_smem_raw);
    const EngUnit<T> ua = eng_unit_args<T>(a, smem);
    EngState* const st = a.st;
    const int tid = threadIdx.x;
    if (tid == 0) {
        unsigned my_xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my_xcc));
        eng_add(&st->xcc_wgs[my_xcc & 7u], 1ull);
        eng_wst[0] = 0; eng_wst[1] = 0; eng_wst[2] = -1; eng_wst[3] = 0;
        eng_nap = 0;
        for (int k = 0; k < 8; ++k) eng_ph[k] = 0;
    }

    // measurement (RFLU_ENGINE_TRACE, the TRACE instantiation): where a workgroup's time goes -- [0] whole-block-column tiles, [1] leaf-window tiles,
    // [2] strips + solves, [3] deferred interchanges, [4] everything between two units (scan, claim, waiting, completion), [5] of [4] the part spent
    // asleep with nothing eligible, [6] of [4] from the end of a unit to its count / publication being out; summed over the workgroups
    long long acct[7] = {0, 0, 0, 0, 0, 0, 0};
    long long acct_t = TRACE ? wall_clock64() : 0;
    for (;;) {
        if (tid < 64) eng_scan<T, TRACE>((unsigned long long)__builtin_amdgcn_kernarg_segment_ptr());
        __syncthreads();
        const int kind = eng_sel[0], cb = eng_sel[1], unit = eng_sel[3];
        const unsigned seq = (unsigned)eng_sel[2];
        if (TRACE && tid == 0) { const long long t = wall_clock64(); acct[4] += t - acct_t; acct_t = t; }
        if (kind == ENG_EXIT) break;
        int acct_k = 3;
        if (kind == ENG_MAIN) {
            const EngOp o = eng_op(a.g, cb, (int)(seq >> 1));
            acct_k = (seq & 1u) == 0 ? 2 : (o.type == ENG_OP_BIG ? 0 : 1);
            if (TRACE && tid == 0 && o.type == ENG_OP_LEAF && unit == 0 && cb == (o.j0 + o.jb + NB) / a.g.Wc)
                a.trace[(o.j0 / NB) * 4 + ((seq & 1u) ? 2 : 0)] = wall_clock64();
            // (second set, leaves < 2048: the same leaf on the first column block of the NEXT block column)
            if (TRACE && tid == 0 && o.type == ENG_OP_LEAF && unit == 0 && cb == eng_first_cb(a.g, o.j0 / a.g.W + 1) && o.j0 / NB < 2048)
                a.trace[(2048 + o.j0 / NB) * 4 + ((seq & 1u) ? 2 : 0)] = wall_clock64();
            // (third set: BIG(b) on the column block the chain needs next, the first one of block column b + ahead + 1: slots (1024 + b) * 4 ..)
            if (TRACE && tid == 0 && o.type == ENG_OP_BIG && unit == 0 && cb == eng_first_cb(a.g, o.j0 / a.g.W + eng_ahead(a.g) + 1) && o.j0 / a.g.W < 512)
                a.trace[(1024 + o.j0 / a.g.W) * 4 + ((seq & 1u) ? 2 : 0)] = wall_clock64();
            if ((seq & 1u) == 0) eng_prep_unit<T>(ua, o, unit);
            else eng_gemm_unit<T>(ua, o, unit);
        } else {
            // (fourth set: the block column's own deferred interchanges, left op 0 on its first column block: slots (1536 + b) * 4 + {0, 1})
            if (TRACE && tid == 0 && seq == 0 && unit == 0 && cb == eng_first_cb(a.g, eng_pb(a.g, cb)) && eng_pb(a.g, cb) < 512)
                a.trace[(1536 + eng_pb(a.g, cb)) * 4] = wall_clock64();
            eng_left_unit<T>(ua, cb, (int)seq, unit);
        }
        // ---- completion: drain every wave's stores, one lane releases and counts ------------------------------------------------
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (TRACE && tid == 0) { const long long t = wall_clock64(); acct[acct_k] += t - acct_t; acct_t = t; }
        __syncthreads();
        if (tid == 0) eng_complete<T, TRACE>((unsigned long long)__builtin_amdgcn_kernarg_segment_ptr(), kind, cb, seq, unit);
        if (TRACE && tid == 0) acct[6] += wall_clock64() - acct_t;
        __syncthreads();
    }
    if (TRACE && tid == 0) {
        acct[5] = eng_nap;
        for (int k = 0; k < 7; ++k) eng_add((unsigned long long*)&a.trace[4096 * 4 + k], (unsigned long long)acct[k]);
        for (int k = 0; k < 8; ++k) eng_add((unsigned long long*)&a.trace[4096 * 4 + 8 + k], (unsigned long long)eng_ph[k]);
    }
}

size_t engine_lds_bytes(size_t esize)
{
    return std::max<size_t>(2 * (size_t)G_STAGE * esize, (size_t)NB * EP_XLD * esize);
}

template <typename T>
int launch_engine(Handle* h, hipStream_t stream, const EngArgs<T>& a, int wgs)
{
    const size_t lds = engine_lds_bytes(sizeof(T));
    bool& attr_set = h->eng_attr_set[sizeof(T) == 8 ? 0 : 1];
    if (a.trace) {   // the accounting instantiation (RFLU_ENGINE_TRACE): the shipped one carries neither its counters nor their code
        RFLU_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&engine_kernel<T, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((engine_kernel<T, true>), dim3((unsigned)wgs), dim3(256), lds, stream, a);
        RFLU_HIP(hipGetLastError());
        return RFLU_OK;
    }
    if (!attr_set) {
        RFLU_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&engine_kernel<T, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL((engine_kernel<T, false>), dim3((unsigned)wgs), dim3(256), lds, stream, a);
    RFLU_HIP(hipGetLastError());
    return RFLU_OK;
}

template int launch_engine<double>(Handle*, hipStream_t, const EngArgs<double>&, int);
template int launch_engine<float>(Handle*, hipStream_t, const EngArgs<float>&, int);

// ---- the critical-path stream's side of the protocol ---------------------------------------------------------------------------------
__global__ void eng_wait_kernel(const unsigned long long* flag, unsigned long long value, unsigned long long* abort, int64_t* info)
{
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < value) {
        if (__hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;   // somebody has already given up
        __builtin_amdgcn_s_sleep(4);
        if (wall_clock64() - t0 > 400000000LL) {   // 4 s: the engine is stuck (or gone): raise the timeout flag, release everybody
            __hip_atomic_fetch_or((unsigned long long*)(info + 1), 33ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(abort, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

__global__ void eng_wait_retired_kernel(const unsigned long long* retired, const unsigned long long* target, unsigned long long* abort, int64_t* info)
{
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(retired, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < __hip_atomic_load(target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        if (__hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
        __builtin_amdgcn_s_sleep(4);
        if (wall_clock64() - t0 > 400000000LL) {
            __hip_atomic_fetch_or((unsigned long long*)(info + 1), 33ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(abort, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
}

int launch_eng_wait_retired(Handle* h, int xcc)
{
    EngState* st = static_cast<EngState*>(h->eng_state);
    hipLaunchKernelGGL(eng_wait_retired_kernel, dim3(1), dim3(1), 0, h->stream, &st->retired, &st->xcc_wgs[xcc & 7], &st->abort, h->info_dev);
    RFLU_HIP(hipGetLastError());
    return RFLU_OK;
}

int launch_eng_wait(Handle* h, const unsigned long long* flag, unsigned long long value)
{
    EngState* st = static_cast<EngState*>(h->eng_state);
    hipLaunchKernelGGL(eng_wait_kernel, dim3(1), dim3(1), 0, h->stream, flag, value, &st->abort, h->info_dev);
    RFLU_HIP(hipGetLastError());
    return RFLU_OK;
}

}  // namespace rflu
