// panel_blocked.hip -- pivoted leaf panel (64 columns) as 8 sub-panels of 8 columns: ONE wave of a workgroup carries the whole
// pivot chain of the workgroup's rows, the other waves apply the eliminations to the columns behind the sub-panel.
//
// Same semantics as _generic_lufact! (/root/reference/src/lu.jl:290-338) and the same arithmetic per entry as the leaves of
// panel_local.hip / panel_single.hip / panel.hip: argmax |a_ik| with strict '>' from 0 and lowest position on ties (:298-305),
// interchange by position renaming, reciprocal-multiply scaling (:317-320), zero pivot -> info once, keep updating (:321-334).
// Every entry receives the multiply-adds of the unblocked algorithm with the same operands in the same order, so factors and
// pivots are bit-identical to those kernels (tests/test_gpu_configs.py).
//
// Why: in panel_local.hip a column costs {row waves: barrier, hand-over read, argmax, record, barrier} -> {communication wave:
// combine, publish, poll, reduce, hand-over, barrier} -- two LDS hand-overs and two workgroup barriers (~2800 clocks) around
// the one thing that cannot be avoided, the exchange between the workgroups (~3100 clocks for 32 of them).  Per column the
// chain only ever needs TWO entries of a row (a_c and a_{c+1}); the other 62 multiply-adds per row are throughput work that can
// lag.  So here
//   * the CHAIN wave (wave UW, raised priority) holds the 8 columns of the current sub-panel for ALL rows of the workgroup
//     (UW rows per lane, 8 columns each, in registers), searches, publishes, polls and eliminates by itself: no barrier and no
//     LDS access inside a column;
//   * the UPDATE waves (waves 0..UW-1, one matrix row per thread, all 64 columns in registers) meet the chain wave only at the
//     7 sub-panel boundaries: they take the 8 multipliers of their row from LDS, the 8 pivot rows' entries behind the
//     sub-panel from their owners (one more exchange through tagged records, LDS only for a lone workgroup), finish those
//     rows (u_t = raw_t - sum_{s<t} l_ts u_s, the eliminations the owner would have applied), apply the 8 eliminations to the
//     next sub-panel's columns first (hand-over to the chain wave) and to the rest while the chain wave is already at work.
// Inside a sub-panel the exchange protocol is that of panel_local.hip, one step earlier: header H(c) = {position, a_c,
// a_{c+1} missing one elimination, l_{c-1}} and row record Rw(c) = entries c+2.. of the sub-panel, all missing that same one
// elimination; every reader finishes them itself.
// Roofline: latency -- per column one exchange between the workgroups + ~1000 clocks of one wave; m*w^2 flops reported.
#include <algorithm>
#include <type_traits>

#include "panel_common.hpp"
#include "panel_xchg.hpp"

namespace rflu {

constexpr int SB = 8;            // columns of a sub-panel
constexpr int NSUB = NB / SB;    // sub-panels of a leaf
constexpr unsigned PB_SLOT_BYTES = NB * PS_VAL_BYTES;   // record of one pivot row at a sub-panel boundary: 64 granules

template <typename T>
struct BlkArgs {
    PanelArgs<T> p;
    int stride;      // !LOCAL: participants are the blocks with blockIdx % stride == sel
    int sel;
    int want_xcc;    // LOCAL: participants are the blocks running on this XCC
    int poll_delay;  // clocks between a workgroup's header publish and its poll round (first step; then adapted)
    int poll_adapt;
};

template <typename T, int UW>
struct BLds {
    static constexpr int ROWS = UW * 64;
    static constexpr int SLD = sizeof(T) == 8 ? 10 : 12;   // slab row stride in elements (80 / 48 bytes: 16-byte aligned, banks spread)
    T slab[ROWS * SLD];      // the NEXT sub-panel's columns of every row, all earlier eliminations applied: update lanes -> chain wave
    T lcol[2][SB][ROWS];     // by parity of the sub-panel: column i after elimination i (the multiplier l_i of every row that took part)
    T U[2][SB][NB];          // by parity: the sub-panel's pivot rows behind the sub-panel, complete
    T raw[SB][NB];           // a lone workgroup: pivot row i behind the sub-panel as its owner holds it (eliminations of earlier sub-panels only)
    T lp[SB][SB];            // ... and lp[i][s]: its multiplier in elimination s < i of the sub-panel
    T stage[UW][NB];         // per update wave: staging of a record publish
    T rowtmp[SB];            // chain wave: a candidate row on its way from one lane's registers to the lanes
    T pivsave[2][SB][SB];    // by parity: the sub-panel's columns of its pivot rows, complete (written when a row retires)
    unsigned fpos[ROWS];     // final position of every row
    unsigned piv[NB];        // workgroup 0: the pivots' positions
    int pivloc[2][SB];       // by parity: local row of the sub-panel's pivot i (-1: another workgroup's)
    int rows[NB];
    int zinfo;
    int dead;
    // progress counters (only ever grow; LDS operations of one wave are performed in order, so data written before a counter is
    // there when the new count is seen)
    int step;                // chain wave: columns decided, eliminated and exported to lcol / pivloc so far
    int subdone;             // chain wave: sub-panels whose pivsave entries are complete
    int nearseq;             // pivot rows whose next-sub-panel entries are in U so far
    int rawseq;              // a lone workgroup: pivot rows written to raw / lp so far
    int useq;                // update waves: pivot rows complete in U so far
    int next_cnt;            // update waves: deliveries of a next sub-panel to the slab (UW per sub-panel)
    int done;                // chain wave: final positions written
};

// LDS progress counters: plain reads / writes (the hardware keeps a wave's LDS operations in order), fenced against the compiler
__device__ __forceinline__ int lds_peek(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_post(int* p, int v)
{
    asm volatile("" ::: "memory");
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// wait until *p >= want (or the workgroup has given up); false on a timeout of its own
__device__ __forceinline__ bool lds_wait(const int* p, int want, const int* dead)
{
    int spins = 0;
    while (lds_peek(p) < want) {
        __builtin_amdgcn_s_sleep(1);
        if (((++spins) & 1023) == 0 && (lds_peek(dead) != 0 || spins > (SPIN_LIMIT << 2))) { asm volatile("" ::: "memory"); return false; }
    }
    asm volatile("" ::: "memory");
    return true;
}

// ---- integer keys of the pivot search (see panel_xchg.hpp): 64 bits for Float64, 32 for Float32; 0 for zero / NaN entries
template <typename T>
struct BKey;
template <>
struct BKey<double> {
    typedef u64 key_t;
    static __device__ __forceinline__ u64 of(double v)
    {
        const u64 b = (u64)__double_as_longlong(v) & 0x7fffffffffffffffull;
        return (__builtin_fabs(v) > 0.0) ? b : 0ull;
    }
    static __device__ __forceinline__ unsigned hi(u64 k) { return (unsigned)(k >> 32); }
    static __device__ __forceinline__ unsigned lo(u64 k) { return (unsigned)k; }
};
template <>
struct BKey<float> {
    typedef unsigned key_t;
    static __device__ __forceinline__ unsigned of(float v)
    {
        return (__builtin_fabsf(v) > 0.0f) ? (__float_as_uint(v) & 0x7fffffffu) : 0u;
    }
    static __device__ __forceinline__ unsigned hi(unsigned k) { return k; }
    static __device__ __forceinline__ unsigned lo(unsigned) { return 0u; }
};

// Compile-time loop: the body sees its index as a constant.  The chain wave's register array is indexed through this only --
// with run-time-looking loop indices (#pragma unroll) the array is still in memory when the optimiser merges the arms of the
// row-slot switches below into ONE access with a selected address, and then it stays in scratch memory for good.
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}
#define SFOR(var, B, E, ...) static_for<B, E>([&](auto var##_c) __attribute__((always_inline)) { constexpr int var = decltype(var##_c)::value; __VA_ARGS__ })

template <typename T, int SLD, int OFF>
__device__ __forceinline__ void slab_store8(T* slab, int r, const T (&v)[NB])
{
    constexpr int VW = 16 / (int)sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(VW)));
    vec_t* d = reinterpret_cast<vec_t*>(slab + (size_t)r * SLD);
    SFOR(jv, 0, SB / VW, {
        vec_t x;
        SFOR(el, 0, VW, { x[el] = v[OFF + jv * VW + el]; });
        d[jv] = x;
    });
}
template <typename T, int SLD, int OFF>
__device__ __forceinline__ void slab_load8(const T* slab, int r, T (&v)[NB])
{
    constexpr int VW = 16 / (int)sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(VW)));
    const vec_t* sp = reinterpret_cast<const vec_t*>(slab + (size_t)r * SLD);
    SFOR(jv, 0, SB / VW, {
        const vec_t x = sp[jv];
        SFOR(el, 0, VW, { v[OFF + jv * VW + el] = x[el]; });
    });
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t blk_rsrc(u64* scratch)   // boundary records: [parity of the sub-panel][pivot t]
{
    return __builtin_amdgcn_make_buffer_rsrc(scratch + PX_OFFSET_WORDS, 0, 2 * SB * PB_SLOT_BYTES, 0x00020000);
}

// =====================================================================================================================
// The chain wave
// =====================================================================================================================
// Its row array e[q * SB + j] (row slot q of the lane, column j of the sub-panel) is ONLY ever written by unconditional
// straight-line multiply-adds: no branch, switch arm or predicate touches it.  (With 56 doubles per lane live across every
// step, each conditional write costs a copy of the whole array at the join, and identical arms of a row-slot switch are
// merged into one access through a selected pointer, which sends the array to scratch memory.)  What makes that possible:
//   * rows that are not active any more (retired pivots, rows beyond m) keep being "eliminated" like everybody else; what
//     their registers hold is never stored and never searched -- amask[q] (0 / 0x7fffffff per row) takes them out of the
//     search, the pivot row's final values go to sh->pivsave the moment it retires, and a row retired in an earlier
//     sub-panel is final in its update lane's registers (cnt -1: that lane does not read the slab);
//   * a timeout ("dead") does not skip anything either: the polls are skipped, the arithmetic runs on garbage, the stores
//     at the end are suppressed by the flag.
template <typename T>
struct ChainCand {   // this workgroup's candidate for the next column, wave-uniform
    unsigned cp;     // position (POS_NONE: the workgroup has no active row)
    int wl, wq;      // lane and row slot of the chain wave that hold it
    T a1, a2, l;     // a_c (complete), a_{c+1} (misses the last elimination), l of the last elimination
};

template <typename T>
struct ChainPiv {    // the pivot of a column as every workgroup sees it, wave-uniform
    unsigned gp;     // position (POS_NONE: no active row anywhere)
    int wg;          // workgroup that owns it
    T ga, gu, gl;    // pivot value, u_{c,c+1} (complete), the pivot row's l of the last elimination
    T scale;         // 1 / pivot (1 for a zero pivot)
    T p1, p2;        // P_{c-1}[c+1], P_{c-1}[c+2]
};
constexpr int BLK_DELAY_MAX = 4000;

// READ entries of row slot q of the row array for a wave-uniform q: scalar branches, no dynamic register index.  The arms end
// in DIFFERENT (empty) asm statements so that they are not merged.
#define BLK_ARM(UWv, n, ...) { constexpr int Q = (n) < (UWv) ? (n) : 0; __VA_ARGS__; asm volatile("; row slot " #n ::: "memory"); } break
#define BLK_SWITCH_Q(UWv, qv, ...)                   \
    do {                                             \
        switch (qv) {                                \
            case 0: BLK_ARM(UWv, 0, __VA_ARGS__);           \
            case 1: BLK_ARM(UWv, 1, __VA_ARGS__);           \
            case 2: BLK_ARM(UWv, 2, __VA_ARGS__);           \
            case 3: BLK_ARM(UWv, 3, __VA_ARGS__);           \
            case 4: BLK_ARM(UWv, 4, __VA_ARGS__);           \
            case 5: BLK_ARM(UWv, 5, __VA_ARGS__);           \
            case 6: BLK_ARM(UWv, 6, __VA_ARGS__);           \
            default: BLK_ARM(UWv, 7, __VA_ARGS__);          \
        }                                            \
    } while (0)

template <typename T>
__device__ __forceinline__ unsigned hi_word(T v);
template <>
__device__ __forceinline__ unsigned hi_word<double>(double v) { return (unsigned)((u64)__double_as_longlong(v) >> 32); }
template <>
__device__ __forceinline__ unsigned hi_word<float>(float v) { return __float_as_uint(v); }

__device__ __forceinline__ unsigned umax3(unsigned a, unsigned b, unsigned c)
{
    unsigned r;
    asm("v_max3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// Search of local column I1 among the active rows of the chain wave (e[q * SB + I1] complete): the workgroup's candidate.
// Usual case: the high words of |a| (sign-free exponent + 20 mantissa bits for Float64, the whole value for Float32) of the
// active rows -- one v_and per row with amask[q] -- have ONE maximum over all rows of all lanes, which is then the strict
// maximum of |a| whatever the low words are: three v_max3 per lane, one DPP reduction, the lane from a ballot, the row slot
// from 7 compares.  Anything else -- maximum zero, an infinity or NaN, two lanes or two rows with the same high word -- takes the
// general path: exact ties go to the lowest position, zero / NaN entries carry key 0 and still take part (an all-zero
// column's pivot is its first row, src/lu.jl:298-305).  Rows q >= 1 of a lane sit at positions rbase + 64 q + lane, ascending in
// q; row 0 may have been displaced to any position (pos0), so the general path compares it last, with its position.
template <typename T, int UW, int I1>
__device__ __forceinline__ void chain_search(const T (&e)[NB], const unsigned (&amask)[8], unsigned actbits, unsigned pos0, int rbase,
                                             int lane, ChainCand<T>& cd)
{
    typedef typename BKey<T>::key_t key_t;
    unsigned hq[8];
    SFOR(q, 0, 8, { hq[q] = q < UW ? (hi_word<T>(e[(q < UW ? q : 0) * SB + I1]) & amask[q]) : 0u; });
    const unsigned mhi = umax3(umax3(hq[0], hq[1], hq[2]), umax3(hq[3], hq[4], hq[5]), max(hq[6], hq[7]));
    const unsigned mh = wave_max_b(mhi);
    const u64 hit = __ballot(mhi == mh);
    constexpr unsigned INF_HI = sizeof(T) == 8 ? 0x7ff00000u : 0x7f800000u;
    bool slow = mh == 0u || mh >= INF_HI || __popcll(hit) != 1;
    int wl = 0, wq = 0;
    unsigned cp = POS_NONE;
    if (!slow) {
        wl = __ffsll((long long)hit) - 1;
        unsigned qml = 0u;   // per lane: the row slots that hold the lane's maximum
        SFOR(q, 0, UW, { qml |= hq[q] == mhi ? (1u << q) : 0u; });
        const unsigned qm = (unsigned)__builtin_amdgcn_readlane((int)qml, wl);
        if (__popc(qm) != 1) slow = true;
        else {
            wq = __ffs((int)qm) - 1;
            cp = wq == 0 ? (unsigned)__builtin_amdgcn_readlane((int)pos0, wl) : (unsigned)(rbase + wq * 64 + wl);
        }
    }
    if (slow) {
        key_t bkey = 0;
        int bq = -1;
        SFOR(q, 1, UW, {
            const key_t k = BKey<T>::of(e[q * SB + I1]);
            if ((actbits & (1u << q)) && k > bkey) { bkey = k; bq = q; }
        });
        if (bq < 0 && (actbits & ~1u) != 0) bq = __ffs((int)(actbits & ~1u)) - 1;   // no positive key: the lowest active row of q >= 1
        unsigned bpos = bq < 0 ? POS_NONE : (unsigned)(rbase + bq * 64 + lane);
        if (actbits & 1u) {
            const key_t k0 = BKey<T>::of(e[I1]);
            if (bq < 0 || k0 > bkey || (k0 == bkey && pos0 < bpos)) { bkey = k0; bq = 0; bpos = pos0; }
        }
        unsigned hi = BKey<T>::hi(bkey), lo = BKey<T>::lo(bkey);
        cp = bpos;
        if (bq < 0) { hi = 0u; lo = 0u; }
        wl = wave_argmax_i<IKey<T>::TWO>(hi, lo, cp);
        wq = __builtin_amdgcn_readlane(bq < 0 ? 0 : bq, wl);
    }
    cd.cp = cp;
    cd.wl = wl;
    cd.wq = wq;
    T a1 = T(0), a2 = T(0), l = T(0);
    BLK_SWITCH_Q(UW, wq, {
        a1 = e[Q * SB + I1];
        if constexpr (I1 + 1 < SB) a2 = e[Q * SB + I1 + 1];
        if constexpr (I1 >= 1) l = e[Q * SB + I1 - 1];
    });
    cd.a1 = readlane_val(a1, wl);
    cd.a2 = readlane_val(a2, wl);
    cd.l = readlane_val(l, wl);
}

// the candidate row's entries j >= I1 + 2 of the sub-panel, one per lane j (lanes 0..7), for the row record: out of the one
// lane that holds them through LDS (a read under a row-slot switch, three 16-byte writes, one read: off the chain)
template <typename T, int UW, int I1>
__device__ __forceinline__ T chain_row_vector(BLds<T, UW>* sh, const T (&e)[NB], const ChainCand<T>& cd, int lane)
{
    T rv = T(0);
    if constexpr (I1 + 2 < SB) {
        if (lane == cd.wl) {
            T x[SB];
            SFOR(j, 0, SB, { x[j] = T(0); });
            BLK_SWITCH_Q(UW, cd.wq, { SFOR(j, I1 + 2, SB, { x[j] = e[Q * SB + j]; }); });
            SFOR(j, I1 + 2, SB, { sh->rowtmp[j] = x[j]; });
        }
        asm volatile("" ::: "memory");
        rv = sh->rowtmp[lane & (SB - 1)];
        asm volatile("" ::: "memory");
    }
    return rv;
}

// Row record Rw(c) of workgroup g: 8 granules.  A workgroup publishes Rw(c + 2) right after it has decided column c + 1, while
// a slower peer may still be polling for column c + 1 -- and for Rw(c), which it reads in that same round: records of columns
// two apart must not share a slot (the headers may: H(c + 2) leaves only after every peer's H(c + 1), i.e. after every peer
// has read H(c)).  Eight slots per workgroup, in its 1 KiB row area of the first record buffer.
__device__ __forceinline__ unsigned row_record_off(int c, int g)
{
    return PS_HDR_REGION + (unsigned)g * PS_ROW_BYTES + (unsigned)(c & 7) * (unsigned)(SB * PS_VAL_BYTES);
}

template <typename T, int AUX>
__device__ __forceinline__ void chain_publish_hdr(u64* scratch, unsigned epoch, int c1, int g, int lane, const ChainCand<T>& cd)
{
    if (lane == 0)
        Hdr4<T>::template store<AUX>(scratch_rsrc(scratch), (unsigned)(c1 & 1) * PS_BUF_BYTES + (unsigned)g * PS_HDR_BYTES,
                                     epoch + (unsigned)c1, cd.cp, cd.a1, cd.a2, cd.l);
}
template <typename T, int AUX>
__device__ __forceinline__ void chain_publish_row(u64* scratch, unsigned epoch, int c1, int i1, int g, int lane, T rv)
{
    if (lane >= i1 + 2 && lane < SB) {
        Gran<T>::template store<AUX>(scratch_rsrc(scratch), row_record_off(c1, g) + (unsigned)lane * PS_VAL_BYTES, epoch + (unsigned)c1, rv);
    }
}

// Decide column c1 = 8 k + I: ONE poll round for the G headers H(c1) and, for I >= 1, the row record Rw(c1 - 1) of the previous
// pivot (its entries j >= I + 1 of the sub-panel), `delay` clocks after this workgroup's own header left.
//   pc (in/out): lane j holds P_{I-2}[j] on entry (I >= 2) and P_{I-1}[j] (j >= I + 1) on exit.
//   prev_wg / prev_l: owner of the previous pivot and its l in the elimination before (from its header).
//   dead (in/out): a peer has timed out: nothing is polled any more.
template <typename T, int I, bool SINGLE>
__device__ __forceinline__ void chain_decide(u64* scratch, unsigned epoch, int G, int c1, int g, int lane, const ChainCand<T>& cd, T rv_own,
                                             int prev_wg, T prev_l, T& pc, int& delay, bool& dead, ChainPiv<T>& pv)
{
    unsigned xp = POS_NONE;
    T xa = T(0), xa1 = T(0), xl = T(0), xv = T(0);
    if constexpr (SINGLE) {
        xp = cd.cp;
        xa = cd.a1;
        xa1 = cd.a2;
        xl = cd.l;
        xv = rv_own;   // published one step ago by this very wave: lane j holds entry j
    } else {
        const __amdgpu_buffer_rsrc_t rs = scratch_rsrc(scratch);
        const bool want_h = lane < G;
        const bool want_r = I >= 1 && prev_wg >= 0 && lane >= I + 1 && lane < SB;
        const unsigned hoff = (unsigned)(c1 & 1) * PS_BUF_BYTES + (unsigned)(want_h ? lane : 0) * PS_HDR_BYTES;
        const unsigned roff = row_record_off(c1 - 1, prev_wg >= 0 ? prev_wg : 0) + (unsigned)(lane & (SB - 1)) * PS_VAL_BYTES;
        if (!dead) {
            if (delay > 0) {
                const long long t0 = clock64();
                while (clock64() - t0 < delay) __builtin_amdgcn_s_sleep(1);
            }
            bool ok_h = !want_h, ok_r = !want_r;
            int spins = 0;
            bool first = true, stale_first = false;
            for (;;) {
                asm volatile("" ::: "memory");   // plain buffer intrinsics: keep the loads inside the loop
                bool got_h = ok_h, got_r = ok_r;
                if (!ok_h) got_h = Hdr4<T>::load(rs, hoff, epoch + (unsigned)c1, xp, xa, xa1, xl);
                if (!ok_r) got_r = Gran<T>::load(rs, roff, epoch + (unsigned)(c1 - 1), xv);
                ok_h = got_h;
                ok_r = got_r;
                if (!__any(!ok_h || !ok_r)) break;
                if (first) stale_first = true;
                first = false;
                if (++spins > SPIN_LIMIT) { dead = true; break; }
            }
            delay = stale_first ? min(delay + 64, BLK_DELAY_MAX) : max(delay - 8, 0);
            if (!ok_h) xp = POS_NONE;
        }
    }
    // P_{I-1}[j], j >= I + 1: the record misses elimination I-2 of its (then candidate) row
    if constexpr (I >= 2) xv -= prev_l * pc;
    if constexpr (I >= 1) pc = xv;
    unsigned hi, lo, gp = xp;
    IKey<T>::split(xa, xp != POS_NONE, hi, lo);
    const T xinv = (xa != T(0)) ? T(1) / xa : T(1);   // every lane divides for ITS header while the reduction runs
    int wl = 0;
    if constexpr (!SINGLE) wl = wave_argmax_i<IKey<T>::TWO>(hi, lo, gp);
    pv.ga = readlane_val(xa, wl);
    const T ga1 = readlane_val(xa1, wl);
    pv.gl = readlane_val(xl, wl);
    pv.scale = readlane_val(xinv, wl);
    pv.gu = ga1;
    pv.p1 = T(0);
    pv.p2 = T(0);
    if constexpr (I >= 1) {
        if constexpr (I + 1 < SB) {
            pv.p1 = readlane_val(pc, I + 1);
            pv.gu = ga1 - pv.gl * pv.p1;   // u_{c1,c1+1}: the header's entry misses elimination I-1
        }
        if constexpr (I + 2 < SB) pv.p2 = readlane_val(pc, I + 2);
    }
    pv.wg = SINGLE ? 0 : wl;
    pv.gp = gp;
}

struct ChainState {
    unsigned amask[8];   // per row slot: 0x7fffffff while the row is active, 0 otherwise
    unsigned actbits;    // the same as bits (general search path, bookkeeping)
    unsigned pos0;       // position of row slot 0
};

// Step I of sub-panel k (column c1 = 8 k + I): decide the pivot, eliminate, search column c1 + 1 and publish its candidate.
template <typename T, int I, int AUX, bool SINGLE, int UW>
__device__ __forceinline__ void chain_step(const BlkArgs<T>& la, BLds<T, UW>* sh, T (&e)[NB], ChainState& st, ChainCand<T>& cd, T& rv, T& rvp,
                                           int& prev_wg, T& prev_l, T& prev_ga, T& prev_gu, T& pc, int& delay, bool& dead, int k, int g, int rbase, int lane)
{
    const PanelArgs<T>& p = la.p;
    const int c1 = k * SB + I;
    ChainPiv<T> pv;
    RFLU_STAMP(p.scratch, c1, 0, g, lane);
    // rv: row record of this workgroup's candidate for column c1, rvp: for column c1 - 1 (a lone workgroup reads its own records)
    chain_decide<T, I, SINGLE>(p.scratch, p.epoch, p.G, c1, g, lane, cd, rvp, prev_wg, prev_l, pc, delay, dead, pv);
    RFLU_STAMP(p.scratch, c1, 1, g, lane);
    // The pivot row of column c1 - 1, complete, for its update lane: the pivot and u_{c1-1,c1} as every workgroup saw them, the
    // entries behind them as every workgroup has just finished them in pc (the same multiply-adds, with the same operands, its
    // owner applies) -- nobody has to dig them out of the row array.
    if constexpr (I >= 1) {
        if (lane == 0) {
            sh->pivsave[k & 1][I - 1][I - 1] = prev_ga;
            sh->pivsave[k & 1][I - 1][I] = prev_gu;
        }
        if (lane >= I + 1 && lane < SB) sh->pivsave[k & 1][I - 1][lane] = pc;
    }
    const unsigned kpos = (unsigned)(p.r0 + c1);
    const bool any = pv.gp != POS_NONE && !dead;
    const bool retire = any && pv.wg == g;   // this workgroup's candidate (lane cd.wl, row slot cd.wq) is the pivot
    const int pwl = cd.wl, pwq = cd.wq;
    const int par = k & 1;
    if (lane == 0) {
        sh->pivloc[par][I] = retire ? pwq * 64 + pwl : -1;
        if (g == 0 && any) {
            sh->piv[c1] = pv.gp;
            if (pv.ga == T(0) && sh->zinfo == 0) sh->zinfo = p.r0 + c1 + 1;
        }
    }
    // the row at position kpos (always one of workgroup 0's rows 0..63, i.e. a row slot 0) takes the pivot's old position
    if (g == 0 && any && (st.actbits & 1u) && st.pos0 == kpos) st.pos0 = pv.gp;
    // the two entries of elimination I-1 the next record needs, then elimination I: every row slot, unconditionally
    SFOR(q, 0, UW, {
        if constexpr (I >= 1) {
            if constexpr (I + 1 < SB) e[q * SB + I + 1] -= e[q * SB + I - 1] * pv.p1;
            if constexpr (I + 2 < SB) e[q * SB + I + 2] -= e[q * SB + I - 1] * pv.p2;
        }
        const T l = e[q * SB + I] * pv.scale;   // reciprocal-multiply (src/lu.jl:317-320); scale == 1 after a zero pivot
        e[q * SB + I] = l;
        if constexpr (I + 1 < SB) e[q * SB + I + 1] -= l * pv.gu;
        sh->lcol[par][I][q * 64 + lane] = l;   // what the update lanes need of this elimination
    });
    if (lane == 0) lds_post(&sh->step, c1 + 1);
    RFLU_STAMP(p.scratch, c1, 2, g, lane);
    // the pivot row retires: out of the search, final position kpos
    {
        const bool me = retire && lane == pwl;
        SFOR(q, 0, UW, { st.amask[q] = (me && pwq == q) ? 0u : st.amask[q]; });
        if (me) {
            st.actbits &= ~(1u << pwq);
            sh->fpos[pwq * 64 + lane] = kpos;
        }
    }
    RFLU_STAMP(p.scratch, c1, 3, g, lane);
    if constexpr (I + 1 < SB) {
        chain_search<T, UW, I + 1>(e, st.amask, st.actbits, st.pos0, rbase, lane, cd);
        if constexpr (!SINGLE) chain_publish_hdr<T, AUX>(p.scratch, p.epoch, c1 + 1, g, lane, cd);
    }
    RFLU_STAMP(p.scratch, c1, 4, g, lane);
    // the rest of elimination I-1 (entries j >= I + 3), next to the exchange
    if constexpr (I >= 1 && I + 3 < SB) {
        T P[SB];
        SFOR(j, I + 3, SB, { P[j] = readlane_val(pc, j); });
        SFOR(q, 0, UW, { SFOR(j, I + 3, SB, { e[q * SB + j] -= e[q * SB + I - 1] * P[j]; }); });
    }
    RFLU_STAMP(p.scratch, c1, 5, g, lane);
    RFLU_STAMP(p.scratch, c1, 6, g, lane);
    rvp = rv;
    if constexpr (I + 1 < SB) {
        rv = chain_row_vector<T, UW, I + 1>(sh, e, cd, lane);
        if constexpr (!SINGLE) chain_publish_row<T, AUX>(p.scratch, p.epoch, c1 + 1, I + 1, g, lane, rv);
    }
    RFLU_STAMP(p.scratch, c1, 7, g, lane);
    prev_wg = any ? pv.wg : -1;
    prev_l = pv.gl;
    prev_ga = pv.ga;
    prev_gu = pv.gu;
    if constexpr (I == SB - 1) {
        if (lane == 0) sh->pivsave[k & 1][I][I] = pv.ga;   // the sub-panel's last pivot row ends with the pivot
    }
}

template <typename T, int AUX, bool SINGLE, int UW>
__device__ __forceinline__ void chain_main(const BlkArgs<T>& la, BLds<T, UW>* sh, int g, int rbase, int lane)
{
    typedef BLds<T, UW> L;
    const PanelArgs<T>& p = la.p;
    T e[NB];
    ChainState st;
    st.actbits = 0u;
    SFOR(q, 0, 8, { st.amask[q] = 0u; });
    SFOR(q, 0, UW, {
        const int row = rbase + q * 64 + lane;
        if (row < p.m) { st.actbits |= 1u << q; st.amask[q] = 0x7fffffffu; }
        sh->fpos[q * 64 + lane] = row < p.m ? (unsigned)row : POS_NONE;
    });
    st.pos0 = (st.actbits & 1u) ? (unsigned)(rbase + lane) : POS_NONE;
    int delay = (SINGLE || AUX == 0) ? 0 : la.poll_delay;   // XCD-local records (plain stores) land at once
    bool dead = false;
    barrier_lds_only();   // the one workgroup barrier: counters are zero, sub-panel 0 of every row is in the slab
    for (int k = 0; k < NSUB; ++k) {
        // the update lanes have applied every earlier elimination to this sub-panel's columns
        if (k > 0 && !lds_wait(&sh->next_cnt, UW * k, &sh->dead)) dead = true;
        SFOR(q, 0, UW, { slab_load8<T, L::SLD, q * SB>(sh->slab, q * 64 + lane, e); });
        ChainCand<T> cd;
        T rv, rvp = T(0), pc = T(0), prev_l = T(0), prev_ga = T(0), prev_gu = T(0);
        int prev_wg = -1;
        // prologue of the sub-panel: candidates of its first column, complete entries
        chain_search<T, UW, 0>(e, st.amask, st.actbits, st.pos0, rbase, lane, cd);
        if constexpr (!SINGLE) chain_publish_hdr<T, AUX>(p.scratch, p.epoch, k * SB, g, lane, cd);
        rv = chain_row_vector<T, UW, 0>(sh, e, cd, lane);
        if constexpr (!SINGLE) chain_publish_row<T, AUX>(p.scratch, p.epoch, k * SB, 0, g, lane, rv);
        int d = delay;
        chain_step<T, 0, AUX, SINGLE, UW>(la, sh, e, st, cd, rv, rvp, prev_wg, prev_l, prev_ga, prev_gu, pc, d, dead, k, g, rbase, lane);
        chain_step<T, 1, AUX, SINGLE, UW>(la, sh, e, st, cd, rv, rvp, prev_wg, prev_l, prev_ga, prev_gu, pc, d, dead, k, g, rbase, lane);
        chain_step<T, 2, AUX, SINGLE, UW>(la, sh, e, st, cd, rv, rvp, prev_wg, prev_l, prev_ga, prev_gu, pc, d, dead, k, g, rbase, lane);
        chain_step<T, 3, AUX, SINGLE, UW>(la, sh, e, st, cd, rv, rvp, prev_wg, prev_l, prev_ga, prev_gu, pc, d, dead, k, g, rbase, lane);
        chain_step<T, 4, AUX, SINGLE, UW>(la, sh, e, st, cd, rv, rvp, prev_wg, prev_l, prev_ga, prev_gu, pc, d, dead, k, g, rbase, lane);
        chain_step<T, 5, AUX, SINGLE, UW>(la, sh, e, st, cd, rv, rvp, prev_wg, prev_l, prev_ga, prev_gu, pc, d, dead, k, g, rbase, lane);
        chain_step<T, 6, AUX, SINGLE, UW>(la, sh, e, st, cd, rv, rvp, prev_wg, prev_l, prev_ga, prev_gu, pc, d, dead, k, g, rbase, lane);
        chain_step<T, 7, AUX, SINGLE, UW>(la, sh, e, st, cd, rv, rvp, prev_wg, prev_l, prev_ga, prev_gu, pc, d, dead, k, g, rbase, lane);
        if (la.poll_adapt) delay = d;
        if (dead && lane == 0) {
            sh->dead = 1;
            __hip_atomic_fetch_or((u64*)(p.info + 1), (u64)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) lds_post(&sh->subdone, k + 1);   // every pivsave entry of the sub-panel is written
        if (lds_peek(&sh->dead) != 0) dead = true;
    }
    if (st.actbits & 1u) sh->fpos[lane] = st.pos0;
    if (lane == 0) lds_post(&sh->done, 1);
}

// =====================================================================================================================
// The update waves
// =====================================================================================================================
// They follow the chain wave column by column through LDS counters (no workgroup barrier after the first): as soon as column
// c = 8 K + i is decided and eliminated in the chain wave (sh->step), the pivot's update lane -- in whichever workgroup it
// lives -- publishes the row's entries behind the sub-panel as it holds them, together with its multipliers of the sub-panel's
// earlier eliminations; in every workgroup update wave i % UW collects that record, finishes the row (u_i = raw_i - sum_{s<i}
// l_is u_s: the eliminations its owner would have applied) and posts it in LDS (sh->useq); every update lane then applies
// elimination c to the NEXT sub-panel's 8 columns of its row.  When the chain wave finishes the sub-panel only the last of
// those rounds is still in flight; the columns go to the slab, and the lanes catch up on the columns further right (and
// assemble their row's final values of the sub-panel) while the chain wave is already at work on the next sub-panel.
template <typename T, int K, int AUX, bool SINGLE, int UW>
__device__ __forceinline__ void upd_subpanel(const BlkArgs<T>& la, BLds<T, UW>* sh, T (&a)[NB], bool& alive, bool& dead, PermState& perm, int g,
                                             int tid)
{
    typedef BLds<T, UW> L;
    const PanelArgs<T>& p = la.p;
    const int lane = tid & 63, wave = uni(tid >> 6);
    constexpr int par = K & 1;
    constexpr int NT = NB - (K + 1) * SB;   // columns behind the sub-panel
    constexpr int NX = NT < SB ? NT : SB;   // ... of which the next sub-panel's ("near"; the others are "far")
    unsigned part = 0u;                      // eliminations of the sub-panel this row takes part in
    int myt = -1;                            // the step at which this row is the pivot
    for (int i = 0; i < SB; ++i) {
        const int c = K * SB + i;
        if (!dead && !lds_wait(&sh->step, c + 1, &sh->dead)) dead = true;
        if (lds_peek(&sh->dead) != 0) dead = true;
        const bool piv_here = !dead && sh->pivloc[par][i] == tid;
        if constexpr (NT > 0) {
            const unsigned tag = p.epoch + (unsigned)c;
            const unsigned slot = (unsigned)(par * SB + i) * PB_SLOT_BYTES;
            // ---- the pivot row leaves.  Its NEAR entries (next sub-panel) have received every elimination up to c - 1 in its
            // update lane, step by step (below): they are complete and go out first, by themselves -- the chain wave waits for what
            // they trigger.  Its FAR entries still miss the sub-panel's eliminations s < i: they follow with the row's multipliers.
            if (__ballot(piv_here) != 0) {
                if (piv_here) {
                    if constexpr (SINGLE) {
                        SFOR(x, 0, NX, { sh->U[par][i][x] = a[(K + 1) * SB + x]; });
                        lds_post(&sh->nearseq, c + 1);
                    } else {
                        SFOR(x, 0, NX, {
                            Gran<T>::template store<AUX>(blk_rsrc(p.scratch), slot + (unsigned)x * PS_VAL_BYTES, tag, a[(K + 1) * SB + x]);
                        });
                    }
                }
                if constexpr (NT > NX) {
                    T* dst_raw = SINGLE ? &sh->raw[i][0] : &sh->stage[wave][0];
                    T* dst_lp = SINGLE ? &sh->lp[i][0] : &sh->stage[wave][NB - SB];
                    if (piv_here) {
                        SFOR(x, NX, NT, { dst_raw[x] = a[(K + 1) * SB + x]; });
                        for (int s2 = 0; s2 < i; ++s2) dst_lp[s2] = sh->lcol[par][s2][tid];
                    }
                    if constexpr (SINGLE) {
                        if (piv_here) lds_post(&sh->rawseq, c + 1);
                    } else {
                        asm volatile("" ::: "memory");
                        if ((lane >= NX && lane < NT) || (lane >= NB - SB && lane < NB - SB + i)) {
                            const T v = sh->stage[wave][lane];
                            Gran<T>::template store<AUX>(blk_rsrc(p.scratch), slot + (unsigned)lane * PS_VAL_BYTES, tag, v);
                        }
                        asm volatile("" ::: "memory");
                    }
                }
            }
            // ---- the near entries arrive: update wave i % UW collects them for the workgroup (one poller per workgroup: with every
            // wave polling, the 32 x 7 pollers slow the chain waves' own exchange down: 2400 -> 4800 clocks per column)
            if constexpr (!SINGLE) {
                if (wave == i % UW && !dead) {
                    T un = T(0);
                    const bool want = lane < NX;
                    bool ok = !want;
                    int spins = 0;
                    for (;;) {
                        asm volatile("" ::: "memory");
                        if (!ok) ok = Gran<T>::load(blk_rsrc(p.scratch), slot + (unsigned)(lane & (SB - 1)) * PS_VAL_BYTES, tag, un);
                        if (!__any(!ok)) break;
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > SPIN_LIMIT || (((spins & 255) == 0) && lds_peek(&sh->dead) != 0)) { dead = true; break; }
                    }
                    if (want) sh->U[par][i][lane] = un;
                    if (dead && lane == 0) {
                        sh->dead = 1;
                        __hip_atomic_fetch_or((u64*)(p.info + 1), (u64)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    if (lane == 0) lds_post(&sh->nearseq, c + 1);
                }
            }
            // ---- elimination c on the next sub-panel's columns of this row
            if (!dead && !lds_wait(&sh->nearseq, c + 1, &sh->dead)) dead = true;
            if (alive && !piv_here) {
                part |= 1u << i;
                const T li = sh->lcol[par][i][tid];
                SFOR(x, 0, NX, { a[(K + 1) * SB + x] -= li * sh->U[par][i][x]; });
            }
            // ---- the far entries arrive and are finished (u_i = raw_i - sum_{s<i} l_is u_s): update wave i % UW, nobody waits
            if constexpr (NT > NX) {
                if (wave == i % UW && !dead) {
                    const bool want = (lane >= NX && lane < NT) || (lane >= NB - SB && lane < NB - SB + i);
                    T v = T(0);
                    if constexpr (SINGLE) {
                        if (!lds_wait(&sh->rawseq, c + 1, &sh->dead)) dead = true;
                        if (want) v = lane < NT ? sh->raw[i][lane] : sh->lp[i][lane - (NB - SB)];
                    } else {
                        bool ok = !want;
                        int spins = 0;
                        for (;;) {
                            asm volatile("" ::: "memory");
                            if (!ok) ok = Gran<T>::load(blk_rsrc(p.scratch), slot + (unsigned)lane * PS_VAL_BYTES, tag, v);
                            if (!__any(!ok)) break;
                            __builtin_amdgcn_s_sleep(1);
                            if (++spins > SPIN_LIMIT || (((spins & 255) == 0) && lds_peek(&sh->dead) != 0)) { dead = true; break; }
                        }
                    }
                    if (!lds_wait(&sh->useq, c, &sh->dead)) dead = true;   // the earlier pivot rows of the sub-panel are complete
                    T u = v;
                    for (int s2 = 0; s2 < i; ++s2) u -= readlane_val(v, NB - SB + s2) * sh->U[par][s2][(lane >= NX && lane < NT) ? lane : NX];
                    if (lane >= NX && lane < NT) sh->U[par][i][lane] = u;
                    if (dead && lane == 0) {
                        sh->dead = 1;
                        __hip_atomic_fetch_or((u64*)(p.info + 1), (u64)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    if (lane == 0) lds_post(&sh->useq, c + 1);
                }
            }
        } else {
            if (alive && !piv_here) part |= 1u << i;
        }
        if (piv_here) { myt = i; alive = false; }
        // interchange bookkeeping (last update wave of workgroup 0), off everybody's chain
        if (g == 0 && wave == UW - 1 && !dead) {
            const unsigned w = sh->piv[c];
            if (w != POS_NONE) perm_state_step(perm, p.r0, c, __builtin_amdgcn_readfirstlane((int)w), lane);
        }
    }
    // ---- the next sub-panel to the chain wave: rows that are retired by now hand it zeros
    if constexpr (K + 1 < NSUB) {
        if (alive) slab_store8<T, L::SLD, (K + 1) * SB>(sh->slab, tid, a);
        else {
            T z[NB];
            SFOR(x, 0, SB, { z[x] = T(0); });
            slab_store8<T, L::SLD, 0>(sh->slab, tid, z);
        }
        asm volatile("" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(&sh->next_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // ---- the row's final values of this sub-panel: its multipliers, and from its own pivot step on the pivot row as saved
    if (!dead && !lds_wait(&sh->subdone, K + 1, &sh->dead)) dead = true;
    SFOR(i2, 0, SB, { if (part & (1u << i2)) a[K * SB + i2] = sh->lcol[par][i2][tid]; });
    if (myt >= 0) {
        SFOR(j, 0, SB, { if (j >= myt) a[K * SB + j] = sh->pivsave[par][myt][j]; });
    }
    // ---- the sub-panel's eliminations on the far columns
    if constexpr (NT > SB) {
        if (!dead && !lds_wait(&sh->useq, K * SB + SB, &sh->dead)) dead = true;
        SFOR(t, 0, SB, {
            if (part & (1u << t)) { SFOR(x, SB, NT, { a[(K + 1) * SB + x] -= a[K * SB + t] * sh->U[par][t][x]; }); }
        });
    }
}

template <typename T, int K0, int AUX, bool SINGLE, int UW>
struct UpdBlocks {
    static __device__ __forceinline__ void run(const BlkArgs<T>& la, BLds<T, UW>* sh, T (&a)[NB], bool& alive, bool& dead, PermState& perm, int g,
                                               int tid)
    {
        if constexpr (K0 < NSUB) {
            upd_subpanel<T, K0, AUX, SINGLE, UW>(la, sh, a, alive, dead, perm, g, tid);
            UpdBlocks<T, K0 + 1, AUX, SINGLE, UW>::run(la, sh, a, alive, dead, perm, g, tid);
        }
    }
};

template <typename T, int AUX, bool SINGLE, int UW>
__device__ __forceinline__ void upd_main(const BlkArgs<T>& la, BLds<T, UW>* sh, int g, int rbase, int tid)
{
    typedef BLds<T, UW> L;
    const PanelArgs<T>& p = la.p;
    const int lane = tid & 63, wave = uni(tid >> 6);
    const int row = rbase + tid;
    const bool valid = row < p.m;
    T a[NB];
    load_row_direct<T>(p.R, p.ld, row, valid, p.c0, NB, a);
    if (g == 0 && tid < NB) sh->piv[tid] = POS_NONE;
    if (tid == 0) {
        sh->zinfo = 0; sh->dead = 0; sh->step = 0; sh->subdone = 0; sh->nearseq = 0; sh->rawseq = 0; sh->useq = 0; sh->next_cnt = 0; sh->done = 0;
    }
    slab_store8<T, L::SLD, 0>(sh->slab, tid, a);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the row is in registers (see panel_single.hip)
    PermState perm = perm_state_init(lane);
    barrier_lds_only();   // the one workgroup barrier
    bool alive = valid, dead = false;
    UpdBlocks<T, 0, AUX, SINGLE, UW>::run(la, sh, a, alive, dead, perm, g, tid);
    if (!dead && !lds_wait(&sh->done, 1, &sh->dead)) dead = true;
    if (dead || lds_peek(&sh->dead) != 0) return;
    const unsigned fp = sh->fpos[tid];
    store_row_direct<T>(p.R, p.ld, fp, p.c0, NB, a);
    if (g == 0) {
        if (wave == 0) {
            const unsigned gp = sh->piv[lane];
            if (gp != POS_NONE) p.ipiv[p.r0 + lane] = (int64_t)gp + 1;
            if (lane == 0 && sh->zinfo != 0 && p.info[0] == 0) p.info[0] = (int64_t)sh->zinfo;
        }
        if (wave == UW - 1) {
            const int chunk = p.r0 / NB;
            perm_state_finish(perm, p.r0, lane, sh->rows, p.pm_cnt + chunk, p.pm_dst + (size_t)chunk * 2 * NB,
                              p.pm_src + (size_t)chunk * 2 * NB);
        }
    }
}

// UW update waves (64 UW rows, one per thread) + the chain wave.  SINGLE: one workgroup, nothing leaves the CU.
template <typename T, bool LOCAL, bool SINGLE, int UW>
__global__ void __launch_bounds__(UW * 64 + 64) panel_blk_kernel(BlkArgs<T> la)
{
    constexpr int AUX = LOCAL ? 0 : AUX_SC1;
    if constexpr (!SINGLE) {
        if (LOCAL ? ((int)hw_xcc_id() != la.want_xcc) : ((int)(blockIdx.x % (unsigned)la.stride) != la.sel)) return;
    }
    __shared__ BLds<T, UW> s_lds;
    BLds<T, UW>* const sh = &s_lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    const int g = SINGLE ? 0 : (int)(blockIdx.x / (unsigned)la.stride);
    const int rbase = la.p.r0 + g * (UW * 64);
    if (wave == UW) {
        __builtin_amdgcn_s_setprio(3);
        chain_main<T, AUX, SINGLE, UW>(la, sh, g, rbase, lane);
    } else {
        upd_main<T, AUX, SINGLE, UW>(la, sh, g, rbase, tid);
    }
}

// ---- host side.  Translation units: {Float64, Float32} (RFLU_PB_F32)
#if defined(RFLU_PB_F32)
typedef float pb_t;
#else
typedef double pb_t;
#endif

constexpr int PB_UW = (sizeof(pb_t) == 8 ? PANEL_BLOCKED_ROWS_F64 : PANEL_BLOCKED_ROWS_F32) / 64;   // Float64: 7 update waves + the chain wave = 2 waves per SIMD (256 VGPRs: the chain wave holds 7 x 8 doubles per lane); with 8 the 9 waves get 168 VGPRs each and the row array spills

template <>
int launch_panel_blocked<pb_t>(Handle* h, const PanelArgs<pb_t>& p0, int local)
{
    typedef pb_t T;
    if (p0.w != NB) { set_error("launch_panel_blocked: full leaves only (w = %d)", p0.w); return RFLU_ERR_ARG; }
    BlkArgs<T> la;
    la.p = p0;
    const int64_t rows = (int64_t)p0.m - p0.r0;
    constexpr int ROWS = PB_UW * 64;
    la.p.G = (int)((rows + ROWS - 1) / ROWS);
    la.poll_delay = h->tune.poll_delay;
    la.poll_adapt = h->tune.poll_adapt;
    la.stride = 1;
    la.sel = 0;
    la.want_xcc = -1;
    const dim3 block(PB_UW * 64 + 64);
    int grid_g = la.p.G;
    // fault injection (tests): this launch polls for one participant more than it has -- the bounded spins end it with the timeout flag
    const bool ghost = h->tune.debug_ghost_leaf >= 0 && h->coop_leaf_seq == h->tune.debug_ghost_leaf && la.p.G < 63;
    h->coop_leaf_seq++;
    if (la.p.G == 1 && !ghost) {
        hipLaunchKernelGGL((panel_blk_kernel<T, false, true, PB_UW>), dim3(1), block, 0, h->stream, la);
    } else {
        if (ghost) la.p.G += 1;
        if (local) {
            la.stride = 8;
            la.sel = h->panel_xcc;
            la.want_xcc = h->panel_xcc;
            hipLaunchKernelGGL((panel_blk_kernel<T, true, false, PB_UW>), dim3((unsigned)(grid_g * 8)), block, 0, h->stream, la);
        } else {
            hipLaunchKernelGGL((panel_blk_kernel<T, false, false, PB_UW>), dim3((unsigned)grid_g), block, 0, h->stream, la);
        }
    }
    RFLU_HIP(hipGetLastError());
    return RFLU_OK;
}

#if defined(RFLU_PB_F32)
int panel_blocked_resident_limit_f32(int num_cus)
#else
int panel_blocked_resident_limit_f64(int num_cus)
#endif
{
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(&panel_blk_kernel<pb_t, false, false, PB_UW>),
                                                     PB_UW * 64 + 64, 0) != hipSuccess) {
        (void)hipGetLastError();
        nb = 0;
    }
    return nb * num_cus;
}


}  // namespace rflu
