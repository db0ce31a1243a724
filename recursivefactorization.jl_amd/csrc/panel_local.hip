// panel_local.hip -- XCD-local pivoted leaf panel: the cooperative leaf of panel.hip with every workgroup on ONE XCD.
//
// Same semantics as _generic_lufact! (/root/reference/src/lu.jl:290-338) and the same arithmetic per entry as
// panel_pivot_pipe_kernel (panel.hip): argmax |a_ik| with strict '>' from 0 and lowest position on ties (:298-305),
// interchange, reciprocal-multiply scaling (:317-320), zero pivot -> info once, keep updating (:321-334).
//
// What changes is WHERE the workgroups sit and HOW a column's candidates travel (scripts/probes/xcdlocal.hip, MI355X):
//   * A launch of 8*G workgroups puts block b on XCD b % 8; only the blocks of ONE residue class take part, the others
//     exit at once.  All G participants then share one 4 MiB L2.
//   * Producer stores are PLAIN (the line stays in that L2), consumer loads are sc1 (bypass the reader's L1, served by the
//     L2): one hop costs ~0.34 us instead of 0.56 (sc1/sc1, same XCD) .. 0.75 us (across XCDs).  The records are the same
//     data-tagged granules as in panel.hip, so no fence or flag orders anything.
//   * Only wave 0 of a workgroup polls the headers and hands the winner to the other waves through LDS: with all 8 waves of
//     32 workgroups polling, the 256 pollers hammer a handful of L2 lines (all-to-all step 1.3 us vs 0.54 us with one
//     polling wave per workgroup).
// Placement is checked, never assumed: every participant compares HW_REG_XCC_ID with the XCC the host expects (found by a
// census launch at handle creation); a mismatch raises the placement flag in info[1] and the host reports
// RFLU_ERR_PLACEMENT instead of returning factors that may have been computed from stale records.
// With LOCAL = false the same kernel runs with sc1 stores on any placement (stride 1).
//
// Roofline: latency -- w x (one L2 hop + two workgroup barriers + one division); work reported to the timers: m*w^2 flops.
#include <algorithm>

#include "panel_common.hpp"
#include "panel_xchg.hpp"

namespace rflu {

template <typename T>
struct LocalArgs {
    PanelArgs<T> p;
    int stride;     // !LOCAL: participants are the blocks with blockIdx % stride == sel
    int sel;
    int want_xcc;   // LOCAL: participants are the blocks running on this XCC
    int poll_delay; // !LOCAL: clocks between a workgroup's header publish and its poll round (first step; then adapted)
    int poll_adapt; // !LOCAL: adapt the delay step by step (x_w0_exchange)
    int grid_g;     // participants actually launched (== p.G except under fault injection)
};


// =====================================================================================================================
// The pipeline.  Column c's pivot needs a full all-to-all exchange; everything else is arranged so that between two
// exchanges the threads execute as few dependent instructions as possible (two waves share a SIMD: every wave instruction on
// the chain costs ~8 clocks, and the old kernels ran ~850 of them per column on every wave):
//   * H(c), the header a workgroup publishes for column c, = {position, a_c, a_{c+1}, l_{c-1}} of its candidate row, where
//     a_c has received eliminations 0..c-1, a_{c+1} only 0..c-2, and l_{c-1} is the row's multiplier of elimination c-1.  The
//     reader completes the lagging entry itself: u_{c,c+1} = a_{c+1} - l_{c-1} * P_{c-1}[c+1] (P_{c-1} = pivot row c-1, which
//     every workgroup holds) -- the same multiply-add, with the same operands, the owner applies to its own register later.
//     So a header never waits for the previous pivot row.
//   * Rw(c), the candidate's row record, = entries j >= c+2 with a_{c+2} through elimination c-1 and a_{j>=c+3} through c-2;
//     the reader finishes them the same way: P_c[j] = Rw(c)[j] - (j >= c+3 ? l_{c-1} * P_{c-1}[j] : 0).  The owner learns
//     that it is its workgroup's candidate one barrier after the header left and publishes the row then (one coalesced
//     store of its wave, staged through LDS); nobody waits for it before the next barrier A.
//   * per step, all threads:  barrier A(c) -> read {pivot position, 1/pivot, u_{c,c+1}} from LDS -> deferred multiply-adds
//     of elimination c-1 on the two entries the next record needs -> interchange bookkeeping, l_c, a_{c+1} -> wave argmax
//     (integer keys, DPP) -> the wave's record {key, position, a_{c+1}, a_{c+2}, l_c} to LDS -> barrier B(c) -> the rest of
//     elimination c-1 (off the chain, next to the exchange).
//   * per step, wave 0 only (alone on the chain, raised priority): after B(c) combine the 8 wave records, publish H(c+1),
//     finish P_c, poll the G headers H(c+1), reduce, divide once, hand over through LDS -> barrier A(c+1).
// Every entry receives exactly the multiply-adds of the unblocked algorithm in the same order: results are bit-identical
// to panel_pivot_kernel / panel_pivot_pipe_kernel.
// =====================================================================================================================

template <typename T>
struct alignas(16) XHand {   // what the communication wave hands to the row waves at barrier A(c): read as a whole right
    T scale;            // after the barrier (one LDS round trip for everything).  1 / pivot (1 when the pivot is exactly zero)
    T wu;               // u_{c,c+1}
    T p1, p2;           // P_{c-1}[c+1], P_{c-1}[c+2]: what the two chain entries still miss
    unsigned win;       // the pivot's row position (POS_NONE: no candidate anywhere, POS_DEAD: a peer timed out)
    unsigned cpos;      // position of THIS workgroup's candidate for column c (its owner publishes Rw(c))
    unsigned pad[sizeof(T) == 8 ? 2 : 2];
};

template <typename T>
struct alignas(16) XRec {      // a wave's candidate for column c+1: three 16-byte LDS accesses for the lane that leaves it
    unsigned hi, lo, pos, pad;
    T a1, a2;                  // a_{c+1} (complete), a_{c+2} (misses elimination c)
    T l, pad2;                 // l_c of that row
};

template <typename T>
struct XLds {
    T prow[2][NB];             // P_c by parity of c (entries j >= c+2)
    T crow[NB];                // staging of the candidate row for the coalesced publish
    XHand<T> hand[2];          // by parity of c
    XRec<T> rec[PANEL_WAVES];  // per-wave records of the search for column c+1
    T w0_lh[2];                // wave 0: l^H of the winner of column c (finishes P_c), by parity of c
    int w0_wl[2];              // wave 0: workgroup of the winner of column c
    int rows[NB];
};

// All waves, between the bookkeeping of column c and barrier B(c): search of column c+1 inside the wave; the winning lane
// leaves the wave's record {key, position, a_{c+1}, a_{c+2}, l_c}.
template <typename T, int PW>
__device__ __forceinline__ void x_record(XLds<T>* sh, int tid, T a1, T a2, T l, unsigned pos, bool act)
{
    const int lane = tid & 63, wave = uni(tid >> 6);
    if (wave < PW) {
        unsigned hi, lo, p = act ? pos : POS_NONE;
        IKey<T>::split(a1, act, hi, lo);
        unsigned mh = hi, ml = lo;
        const int wl = wave_argmax_i<IKey<T>::TWO>(mh, ml, p);
        if (lane == wl) {   // the winning lane leaves the record itself (lane 0 an empty one if the wave has no candidate)
            XRec<T>* r = &sh->rec[wave];
            r->hi = mh;
            r->lo = ml;
            r->pos = p;
            r->a1 = a1;
            r->a2 = a2;
            r->l = l;
        }
    }
    barrier_lds_only();   // B
}

// Communication wave after barrier B(c): the workgroup's candidate for column c+1 from the PW wave records; its header leaves at once.
template <typename T, int AUX, int PW>
__device__ __forceinline__ void x_w0_publish(XLds<T>* sh, u64* scratch, unsigned epoch, int c1, int g, int lane)
{
    scratch = uni(scratch);
    epoch = uni(epoch);
    c1 = uni(c1);
    g = uni(g);
    const bool has = lane < PW;   // lanes 0..PW-1 hold the wave records
    const int r = has ? lane : 0;
    const XRec<T>* rc = &sh->rec[r];
    unsigned hi = rc->hi, lo = rc->lo, cp = rc->pos;
    const T a1 = rc->a1, a2 = rc->a2, l = rc->l;
    if (!has) { hi = 0u; lo = 0u; cp = POS_NONE; }
    const int wl = (PW == 1) ? 0 : wave_argmax_i<IKey<T>::TWO, true>(hi, lo, cp);
    if (PW == 1) cp = (unsigned)__builtin_amdgcn_readlane((int)cp, 0);
    if (lane == wl)   // the lane that holds the winning record (an empty header if there is no candidate: cp == POS_NONE)
        Hdr4<T>::template store<AUX>(scratch_rsrc(scratch), (unsigned)(c1 & 1) * PS_BUF_BYTES + (unsigned)g * PS_HDR_BYTES,
                                     epoch + (unsigned)c1, cp, a1, a2, l);
    if (lane == 0) sh->hand[c1 & 1].cpos = cp;
    if (c1 >= 1) RFLU_STAMP(scratch, c1 - 1, 6, g, lane);
}

// Communication wave before barrier A(c1): ONE poll round for the winner's row record of column c = c1-1 (finishes P_c) and the
// G headers of column c1 -- issued `delay` clocks after this workgroup's own header left, then reduce, complete u_{c1,c1+1},
// divide, hand over.
//   The delay: a poll that reaches the memory side before a peer's write-through store has landed comes back stale and costs a
//   whole second round trip; a pause of about a third of a microsecond between the publish and the first poll lets the
//   slowest header land first (scripts/probes/xchg.hip: all-to-all of 32 workgroups 1.57 us polling at once, 1.22 us with an
//   800-clock pause; the round-3 kernel got that pause by accident, from requesting the row record first and the headers only
//   after it had arrived: two dependent round trips).  It is adapted per workgroup: longer after a stale first poll, slowly
//   shorter after a fresh one.
// Returns true when a peer timed out: the hand-over then carries POS_DEAD as the pivot position, which is how the row waves learn of
// it (one LDS read per step for everything, instead of a separate flag word read -- and waited for -- ahead of the hand-over).
constexpr unsigned POS_DEAD = 0x7ffffffeu;
constexpr int POLL_DELAY_MAX = 4000;
template <typename T>
__device__ __forceinline__ bool x_w0_exchange(XLds<T>* sh, u64* scratch, int64_t* info, int64_t* ipiv, unsigned epoch, int G,
                                           int c1, int r0, int g, int lane, int& delay)
{
    scratch = uni(scratch);
    info = uni(info);
    ipiv = uni(ipiv);
    epoch = uni(epoch);
    G = uni(G);
    c1 = uni(c1);
    r0 = uni(r0);
    g = uni(g);
    const __amdgpu_buffer_rsrc_t rs = scratch_rsrc(scratch);
    const int c = c1 - 1;
    bool timed_out = false;
    // ---- what this lane asks for: lane x < G the header of workgroup x (column c1); lane j in [c+2, NB) entry j of the row
    // record of column c's winner
    int wlc = -1;
    T lh = T(0);
    if (c >= 0) {
        wlc = uni(sh->w0_wl[c & 1]);
        lh = sh->w0_lh[c & 1];
    }
    const bool want_h = lane < G;
    const bool want_r = c >= 0 && wlc >= 0 && lane >= c + 2 && lane < NB;
    const unsigned hoff = (unsigned)(c1 & 1) * PS_BUF_BYTES + (unsigned)(want_h ? lane : 0) * PS_HDR_BYTES;
    const unsigned roff = (unsigned)(c & 1) * PS_BUF_BYTES + PS_HDR_REGION + (unsigned)(wlc >= 0 ? wlc : 0) * PS_ROW_BYTES +
                          (unsigned)lane * PS_VAL_BYTES;
    if (delay > 0) {
        const long long t0 = clock64();
        while (clock64() - t0 < delay) __builtin_amdgcn_s_sleep(1);
    }
    unsigned xp = POS_NONE;
    T xa = T(0), xa1 = T(0), xl = T(0), xv = T(0);
    bool ok_h = !want_h, ok_r = !want_r;
    int spins = 0;
    bool first = true, stale_first = false;
    for (;;) {
        asm volatile("" ::: "memory");  // plain buffer intrinsics: keep the loads inside the loop
        bool got_h = ok_h, got_r = ok_r;
        if (!ok_h) got_h = Hdr4<T>::load(rs, hoff, epoch + (unsigned)c1, xp, xa, xa1, xl);
        if (!ok_r) got_r = Gran<T>::load(rs, roff, epoch + (unsigned)c, xv);
        ok_h = got_h;
        ok_r = got_r;
        if (!__any(!ok_h || !ok_r)) break;
        if (first) stale_first = true;
        first = false;
        if (++spins > SPIN_LIMIT) { timed_out = true; break; }
    }
    delay = stale_first ? min(delay + 64, POLL_DELAY_MAX) : max(delay - 8, 0);
    if (!ok_h) xp = POS_NONE;
    T pc = T(0);   // P_c[lane]
    if (want_r) {
        // entries j >= c+3 still miss elimination c-1 of the (then) candidate row
        if (c >= 1 && lane >= c + 3) xv -= lh * sh->prow[(c - 1) & 1][lane];
        pc = xv;
        sh->prow[c & 1][lane] = xv;
    }
    unsigned hi, lo, gp = xp;
    IKey<T>::split(xa, xp != POS_NONE, hi, lo);
    // every lane divides for ITS header while the reduction runs (independent instruction streams: the quotient's ~12 dependent
    // operations fill the wait states of the DPP chain); the winner's quotient is picked afterwards -- the same IEEE quotient
    const T xinv = (xa != T(0)) ? T(1) / xa : T(1);
    const int wl = wave_argmax_i<IKey<T>::TWO>(hi, lo, gp);   // the winner's lane is its workgroup index
    const T ga = readlane_val(xa, wl), ga1 = readlane_val(xa1, wl), gl = readlane_val(xl, wl);
    T gu = ga1;
    if (c >= 0 && c1 + 1 < NB) gu = ga1 - gl * readlane_val(pc, c1 + 1);   // u_{c1,c1+1}: the entry misses elimination c
    const bool dead = __any(timed_out);
    if (dead) {
        gp = POS_DEAD;
        if (lane == 0) __hip_atomic_fetch_or((u64*)(info + 1), (u64)65, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const T sc = readlane_val(xinv, wl);   // 1 / pivot (1 when the pivot is exactly zero or there is no candidate)
    const T p1 = readlane_val(pc, (c1 + 1) & 63), p2 = readlane_val(pc, (c1 + 2) & 63);   // P_c[c1+1], P_c[c1+2]
    if (lane == 0) {
        XHand<T>* h = &sh->hand[c1 & 1];
        h->scale = sc;
        h->wu = gu;
        h->p1 = p1;
        h->p2 = p2;
        h->win = gp;
        sh->w0_lh[c1 & 1] = gl;
        sh->w0_wl[c1 & 1] = (gp != POS_NONE && !dead) ? wl : -1;
        if (g == 0 && gp != POS_NONE && !dead) {
            ipiv[r0 + c1] = (int64_t)gp + 1;
            if (ga == T(0) && info[0] == 0) info[0] = (int64_t)r0 + c1 + 1;
        }
    }
    if (c1 >= 1) RFLU_STAMP(scratch, c1 - 1, 7, g, lane);
    return dead;
}

// the candidate row staged in LDS (entries c+2..) leaves as ONE coalesced store of the owner's wave
template <typename T, int AUX>
__device__ __forceinline__ void x_publish_row(XLds<T>* sh, u64* scratch, unsigned epoch, int c, int g, int lane)
{
    scratch = uni(scratch);
    epoch = uni(epoch);
    c = uni(c);
    g = uni(g);
    if (lane >= c + 2 && lane < NB) {
        const T v = sh->crow[lane];
        const unsigned roff = (unsigned)(c & 1) * PS_BUF_BYTES + PS_HDR_REGION + (unsigned)g * PS_ROW_BYTES;
        Gran<T>::template store<AUX>(scratch_rsrc(scratch), roff + (unsigned)lane * PS_VAL_BYTES, epoch + (unsigned)c, v);
    }
}

struct XState {
    unsigned pos;
    bool act;
    bool updprev;   // elimination c-1 still has to reach this row's entries j >= c+1
    bool dead;
};

// Step C >= 0: column C.  C == -1 is the prologue: records of column 0, H(0), first exchange.
// FULL: the leaf has all NB columns -- no run-time column tests between the barriers.
template <typename T, int C, int AUX, int PW, bool FULL>
__device__ __forceinline__ void x_step(const PanelArgs<T>& p, XLds<T>* sh, T (&a)[NB], T& lprev, XState& st, PermState& perm,
                                       int g, int tid)
{
    if ((!FULL && C >= p.w) || st.dead) return;
    const int lane = tid & 63, wave = tid >> 6;
    bool owner = false;
    T l = T(0);
    bool upd = false;
    unsigned win_c = POS_NONE;
    if constexpr (C >= 0) RFLU_STAMP(p.scratch, C, 0, g, tid);
    if constexpr (C >= 0) {
        // the whole hand-over in one LDS round trip, requested before anything is tested
        // (a plain read: barrier A in front of it is an asm statement that clobbers memory, so nothing is carried over from an earlier
        // step.  Round 4 read it through a `volatile` pointer -- hipcc turned that into SIX flat loads (sc0 sc1) with an
        // s_waitcnt vmcnt(0) behind each: six dependent trips through the vector memory path to LDS, each also waiting for the
        // acknowledgement of the row-record stores in flight, at the top of every pivot step of every row wave)
        // ... and as ONE round trip: all 16-byte pieces requested together and pinned in registers by an empty asm statement, or hipcc
        // reads the winner first, branches, reads the next two pieces, branches, reads the third (three dependent LDS latencies)
        typedef unsigned xh_u4 __attribute__((ext_vector_type(4)));
        constexpr int XH_Q = (int)(sizeof(XHand<T>) / 16);
        static_assert(sizeof(XHand<T>) % 16 == 0, "XHand is read in 16-byte pieces");
        xh_u4 xq[XH_Q];
        {
            const xh_u4* hq = reinterpret_cast<const xh_u4*>(&sh->hand[C & 1]);
#pragma unroll
            for (int i = 0; i < XH_Q; ++i) xq[i] = hq[i];
#pragma unroll
            for (int i = 0; i < XH_Q; ++i) asm volatile("" : "+v"(xq[i]));
        }
        XHand<T> hcopy;
        __builtin_memcpy(&hcopy, xq, sizeof(hcopy));
        const XHand<T>* hv = &hcopy;
        const T h_scale = hv->scale, h_wu = hv->wu, h_p1 = hv->p1, h_p2 = hv->p2;
        // (the asm statement hides that the pieces came from a wave-uniform address: the two control words go back to scalar registers,
        // or every test below turns into exec-mask arithmetic)
        const unsigned h_win = (unsigned)__builtin_amdgcn_readfirstlane((int)hv->win), h_cpos = (unsigned)__builtin_amdgcn_readfirstlane((int)hv->cpos);
        if (h_win == POS_DEAD) { st.dead = true; return; }
        owner = st.pos == h_cpos && st.pos != POS_NONE;
        if constexpr (C >= 1) {
            if (st.updprev) {   // elimination C-1 on the two entries the next record needs
                if constexpr (C + 1 < NB) a[C + 1] -= lprev * h_p1;
                if constexpr (C + 2 < NB) a[C + 2] -= lprev * h_p2;
            }
        }
        const unsigned kpos = (unsigned)(p.r0 + C);
        const bool is_piv = st.act && st.pos == h_win;          // h_win != POS_NONE whenever an active row matches it
        upd = st.act && !is_piv && h_win != POS_NONE;
        if (upd) {
            if (st.pos == kpos) st.pos = h_win;   // displaced row takes the pivot's old position
            l = a[C] * h_scale;                   // reciprocal-multiply (src/lu.jl:317-320); scale == 1 after a zero pivot
            a[C] = l;
            if constexpr (C + 1 < NB) a[C + 1] -= l * h_wu;
        }
        if (is_piv) {
            st.pos = kpos;      // pivot row: final position r0+C, no further updates
            st.act = false;
        }
        win_c = h_win;
    }
    const bool more = FULL || C + 1 < p.w;   // workgroup-uniform
    if constexpr (C + 1 < NB) {
        if (more) {
            T a2 = T(0);
            if constexpr (C + 2 < NB) a2 = a[C + 2];
            if constexpr (C >= 0) RFLU_STAMP(p.scratch, C, 1, g, tid);
            x_record<T, PW>(sh, tid, a[C + 1], a2, l, st.pos, st.act);   // ends with barrier B
            if constexpr (C >= 0) RFLU_STAMP(p.scratch, C, 2, g, tid);
            if constexpr (C >= 0) RFLU_STAMP(p.scratch, C, 3, g, tid);
        }
    }
    if constexpr (C >= 0 && C + 2 < NB) {
        if (owner) {   // Rw(C): a[C+2] has elimination C-1, the rest C-2 (the deferred loop below has not run yet)
#pragma unroll
            for (int j = C + 2; j < NB; ++j) sh->crow[j] = a[j];
        }
        if (__ballot(owner) != 0) x_publish_row<T, AUX>(sh, p.scratch, p.epoch, C, g, lane);
    }
    // interchange bookkeeping of column C (one wave of workgroup 0): AFTER barrier B, next to the exchange -- before it, that
    // wave reached the barrier ~40 instructions after everybody else, every step, and every workgroup waits for the slowest one
    if constexpr (C >= 0) {
        if (g == 0 && wave == PW - 1 && win_c != POS_NONE)
            perm_state_step(perm, p.r0, C, __builtin_amdgcn_readfirstlane((int)win_c), lane);
    }
    if constexpr (C >= 1 && C + 3 < NB) {
        if (st.updprev) {   // the rest of elimination C-1
            const T* P = sh->prow[(C - 1) & 1];
#pragma unroll
            for (int j = C + 3; j < NB; ++j) a[j] -= lprev * P[j];
        }
    }
    lprev = l;
    st.updprev = upd;
    if constexpr (C >= 0) RFLU_STAMP(p.scratch, C, 4, g, tid);
    if constexpr (C + 1 < NB) {
        if (more) {
            if constexpr (C >= 0) RFLU_STAMP(p.scratch, C, 5, g, tid);
            barrier_lds_only();   // A(C+1)
        }
    }
}

template <typename T, int C0, int C1, int AUX, int PW, bool FULL>
struct XSteps {
    static __device__ __forceinline__ void run(const PanelArgs<T>& p, XLds<T>* sh, T (&a)[NB], T& lprev, XState& st,
                                               PermState& perm, int g, int tid)
    {
        if constexpr (C0 < C1) {
            x_step<T, C0, AUX, PW, FULL>(p, sh, a, lprev, st, perm, g, tid);
            XSteps<T, C0 + 1, C1, AUX, PW, FULL>::run(p, sh, a, lprev, st, perm, g, tid);
        }
    }
};

// PW = row waves per workgroup: 8 (512 rows, two row waves per SIMD) for tall panels; 4 (256 rows, ONE row wave per SIMD: the
// dependent instruction chain between the barriers runs without a second wave sharing the issue slots) while the panel's
// rows still fit 32 such workgroups.  Full leaves only (w == NB; launch_panel sends a narrower last leaf to the kernels of
// panel.hip / panel_single.hip): the 64 steps are straight-line code without run-time column tests.
// SP (PW = 6): the communication wave gets a SIMD for itself.  The waves of a workgroup go round the four SIMDs in a fixed cycle
// (waves w, w + 4, w + 8 share one: scripts/probes/simdmap.hip), so in the 9-wave workgroup of PW = 8 the communication wave sits on
// a SIMD with two row waves whose multiply-adds fill it, and every one of its ~400 instructions per column waits for them (publish
// 900 clocks, poll-to-hand-over 3600+, against 450 / 2400 for a wave by itself).  Here the workgroup is launched with 9 waves of
// which waves 0 and 4 leave at once: wave 8 has their SIMD, the six row waves (384 rows) the other three.
template <typename T, bool LOCAL, int PW, int SP = 0>
__global__ void __launch_bounds__((SP ? PW + 3 : PW + 1) * 64) panel_pivot_local_kernel(LocalArgs<T> la)
{
    constexpr int AUX = LOCAL ? 0 : AUX_SC1;
    // LOCAL: the participants are the blocks that RUN on the chosen XCD.  A launch spreads its blocks round-robin over the
    // 8 XCDs (block b -> XCD (b + offset) % 8, the offset varies from launch to launch), so exactly one block of every
    // group of 8 qualifies and blockIdx / 8 numbers the participants 0..G-1.  Should the hardware ever place blocks
    // differently a participant is missing and the bounded spins end the launch with the timeout flag.
    if (LOCAL ? ((int)hw_xcc_id() != la.want_xcc) : ((int)(blockIdx.x % (unsigned)la.stride) != la.sel)) return;
    const PanelArgs<T>& p = la.p;
    __shared__ XLds<T> s_lds;
    XLds<T>* const sh = &s_lds;
    const int lane = threadIdx.x & 63;
    int wave = (int)(threadIdx.x >> 6);
    if constexpr (SP != 0) {
        static_assert(SP == 0 || PW == 6, "the spare-SIMD variant is laid out for 6 row waves in a 9-wave launch");
        if (wave == 0 || wave == 4) return;
        wave = wave == 8 ? PW : wave - 1 - (wave > 4 ? 1 : 0);
    }
    const int tid = wave * 64 + lane;
    const int g = (int)(blockIdx.x / (unsigned)la.stride);
    const int row = p.r0 + g * (PW * 64) + tid;
    // waves 0..7 own one matrix row per thread; wave 8 is the communication wave (no rows): it alone runs the exchange, at
    // raised priority, so none of that sits on a wave that also has row work to do
    if (wave == PW) __builtin_amdgcn_s_setprio(3);
    XState st;
    st.act = row < p.m && wave < PW;
    st.pos = st.act ? (unsigned)row : POS_NONE;
    st.updprev = false;
    st.dead = false;
    if (tid == 0) sh->w0_wl[0] = sh->w0_wl[1] = -1;
    T a[NB];
    load_row_direct<T>(p.R, p.ld, row, st.act, p.c0, NB, a);
    // the rows are in registers before the first step: a real s_waitcnt (one the compiler's wait-count pass sees), so that no
    // conservative vmcnt(0) -- a wait for the acknowledgement of the row-record store -- is left at the merges inside the steps
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) (gfx9 encoding: expcnt 7, lgkmcnt 15 = no wait)
    __syncthreads();
    T lprev = T(0);
    PermState perm = perm_state_init(lane);
    if (wave == PW) {
        // communication wave: a run-time loop (no row registers, no static indices), everything inline -- a non-inlined
        // call would wait for the acknowledgement of the stores just issued (s_waitcnt vmcnt(0) at every call boundary)
        int delay = LOCAL ? 0 : la.poll_delay;   // clocks between the publish and the poll round (adapted, see x_w0_exchange)
        for (int c1 = 0; c1 < NB; ++c1) {
            barrier_lds_only();   // B(c1 - 1): the wave records of column c1 are in LDS
            x_w0_publish<T, AUX, PW>(sh, p.scratch, p.epoch, c1, g, lane);
            int d = delay;
            const bool dead = x_w0_exchange<T>(sh, p.scratch, p.info, p.ipiv, p.epoch, p.G, c1, p.r0, g, lane, d);
            if (!LOCAL && la.poll_adapt) delay = d;
            barrier_lds_only();   // A(c1)
            if (dead) break;
        }
    } else {
        XSteps<T, -1, NB, AUX, PW, true>::run(p, sh, a, lprev, st, perm, g, tid);
        store_row_direct<T>(p.R, p.ld, st.pos, p.c0, NB, a);
    }
    __syncthreads();
    if (g == 0 && wave == PW - 1) {
        const int chunk = p.r0 / NB;
        perm_state_finish(perm, p.r0, lane, sh->rows, p.pm_cnt + chunk, p.pm_dst + (size_t)chunk * 2 * NB,
                          p.pm_src + (size_t)chunk * 2 * NB);
    }
}

// ---- host side.  Four translation units (parallel compile of the 64-step kernels): {Float64, Float32} x {any placement,
// XCD-local}; each holds the launches of its four workgroup sizes (launch_panel_local_variant), the Float64 / any-placement
// one also the choice between them.
#if defined(RFLU_PL_F32)
typedef float pl_t;
#else
typedef double pl_t;
#endif
#if defined(RFLU_PL_XCD)
constexpr bool PL_LOCAL = true;
#else
constexpr bool PL_LOCAL = false;
#endif

template <typename T, bool LOCAL>
int launch_panel_local_variant(Handle* h, const LocalArgs<T>& la, int rpw, int ballast);
template <typename T, bool LOCAL>
int panel_local_resident_limit_variant(int num_cus);

template <>
int launch_panel_local_variant<pl_t, PL_LOCAL>(Handle* h, const LocalArgs<pl_t>& la, int rpw, int ballast)
{
    typedef pl_t T;
    constexpr bool LOCAL = PL_LOCAL;
    const dim3 grid((unsigned)(la.grid_g * la.stride));
    if (h->coop_launch && !LOCAL) {   // launch-time residency check by the runtime (opt-in: +15-19 us per launch)
        LocalArgs<T> lc = la;
        void* kargs[] = {&lc};
        RFLU_HIP(hipLaunchCooperativeKernel(reinterpret_cast<const void*>(&panel_pivot_local_kernel<T, LOCAL, 8>), grid,
                                            dim3(8 * 64 + 64), kargs, 0, h->stream));
        return RFLU_OK;
    }
    if (rpw <= 128) {
        // LDS ballast: a two- or three-wave workgroup fits next to the update GEMM's workgroups on a shared CU and is then
        // slowed by them (N=4096: 13.3 ms without, 12.7 ms with); asking for more LDS than a CU with a GEMM workgroup (70 KB
        // each) has left sends it to an empty CU -- one of those the update stream's mask keeps free.
        // (the XCD-local variants too: without it N=4096 11.12-11.15 ms, with it 10.74-10.82)
        bool& attr_set = h->panel_attr_set[LOCAL ? 1 : 0][sizeof(T) == 8 ? 0 : 1][rpw == 64 ? 0 : 1];   // per handle = per device
        const void* fn = rpw == 64 ? reinterpret_cast<const void*>(&panel_pivot_local_kernel<T, LOCAL, 1>)
                                   : reinterpret_cast<const void*>(&panel_pivot_local_kernel<T, LOCAL, 2>);
        if (!attr_set && ballast > 0) {
            RFLU_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, ballast));
            attr_set = true;
        }
        if (rpw == 64) hipLaunchKernelGGL((panel_pivot_local_kernel<T, LOCAL, 1>), grid, dim3(1 * 64 + 64), (size_t)ballast, h->stream, la);
        else hipLaunchKernelGGL((panel_pivot_local_kernel<T, LOCAL, 2>), grid, dim3(2 * 64 + 64), (size_t)ballast, h->stream, la);
    } else if (rpw == 256) {
        hipLaunchKernelGGL((panel_pivot_local_kernel<T, LOCAL, 4>), grid, dim3(4 * 64 + 64), 0, h->stream, la);
    } else if (rpw == 384) {
        hipLaunchKernelGGL((panel_pivot_local_kernel<T, LOCAL, 6, 1>), grid, dim3(9 * 64), 0, h->stream, la);
    } else {
        hipLaunchKernelGGL((panel_pivot_local_kernel<T, LOCAL, 8>), grid, dim3(8 * 64 + 64), 0, h->stream, la);
    }
    RFLU_HIP(hipGetLastError());
    return RFLU_OK;
}

template <>
int panel_local_resident_limit_variant<pl_t, PL_LOCAL>(int num_cus)
{
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(&panel_pivot_local_kernel<pl_t, PL_LOCAL, 8>),
                                                     8 * 64 + 64, 0) != hipSuccess) {
        (void)hipGetLastError();
        nb = 0;
    }
    return nb * num_cus;
}

#if !defined(RFLU_PL_XCD)
// the XCD-local variants live in panel_local_xcd*.hip
template <>
int launch_panel_local_variant<pl_t, true>(Handle* h, const LocalArgs<pl_t>& la, int rpw, int ballast);
template <>
int panel_local_resident_limit_variant<pl_t, true>(int num_cus);

#if !defined(RFLU_PL_F32)
// Rows per workgroup of a pivoted leaf of `rows` rows.  The fewer row waves a workgroup has, the cheaper its per-column
// argmax / barrier (scripts/panel_bench.py).  One polling wave reads at most 64 headers and the lookahead schedule keeps 32 or
// 64 CUs free, so a panel takes the smallest workgroup that keeps it at <= h->tune.panel_maxg workgroups; h->tune.panel_pw
// (RFLU_PANEL_PW=1|2|4|8) sets a floor.
int panel_local_rows_per_wg(const Handle* h, int64_t rows, size_t esize)
{
    if (h->coop_launch) return 512;
    const int floor_pw = h->tune.panel_pw > 0 ? h->tune.panel_pw : 1;
    const int max_g = h->tune.panel_maxg;
    if (floor_pw <= 1 && (rows + 63) / 64 <= max_g) return 64;
    if (floor_pw <= 2 && (rows + 127) / 128 <= max_g) return 128;
    if (h->tune.panel_rpw > 0) return h->tune.panel_rpw;   // RFLU_PANEL_RPW: experiments
    // 6 row waves, the communication wave on a SIMD of its own: Float64 panels of 8193..12288 rows (Float32, whose row waves have half
    // the arithmetic, measures 2 % slower with them in the schedule: N=16384 61.4-61.6 -> 62.5-63.3 ms)
    // ... and of 16385..24576 rows (43..64 workgroups: the same 64-CU reservation the 512-row workgroups need there)
    const int64_t g384 = (rows + 383) / 384, g512 = (rows + 511) / 512;
    const bool spare = h->tune.panel_spare && esize == 8 && rows > h->tune.panel_spare_min && floor_pw <= 6 && g384 <= 64 &&
                       (g384 + 31) / 32 == (g512 + 31) / 32;
    if (floor_pw <= 4 && (rows + 255) / 256 <= 32 && !spare) return 256;
    if (spare) return 384;
    return 512;
}
#else
int panel_local_rows_per_wg(const Handle* h, int64_t rows, size_t esize);
#endif

// Launch the leaf on the blocks b with b % stride == sel of a grid of G*stride workgroups.  local != 0: plain-store
// records (all participants must share an XCD: stride 8); local == 0: sc1 records, any placement.
template <typename T>
int launch_panel_local(Handle* h, const PanelArgs<T>& p0, int stride, int sel, int want_xcc, int local)
{
    LocalArgs<T> la;
    la.p = p0;
    la.stride = stride;
    la.sel = sel;
    la.want_xcc = want_xcc;
    la.poll_delay = h->tune.poll_delay;
    la.poll_adapt = h->tune.poll_adapt;
    if (p0.w != NB) { set_error("launch_panel_local: full leaves only (w = %d)", p0.w); return RFLU_ERR_ARG; }
    const int64_t rows = (int64_t)p0.m - p0.r0;
    int rpw = local ? (((rows + 255) / 256 <= 32 && !h->coop_launch) ? 256 : 512) : panel_local_rows_per_wg(h, rows, sizeof(T));
    // XCD-local: at most 16 participants (the other 7/8 of the launch have to find a home too), so the short workgroups whose
    // per-column chain is shorter serve panels of <= 1024 / 2048 rows (N=2048 5.38 -> 5.11 ms, N=4096 11.01 -> 10.74, N=8192 25.47 -> 25.2)
    if (local && !h->coop_launch) {
        if ((rows + 63) / 64 <= 16) rpw = 64;
        else if ((rows + 127) / 128 <= 16) rpw = 128;
        else if (h->tune.panel_local_pw8_rows > 0 && rows > h->tune.panel_local_pw8_rows) rpw = 512;
    }
    la.p.G = (int)((rows + rpw - 1) / rpw);
    la.grid_g = la.p.G;
    // fault injection (tests): this launch polls for one participant more than it has -- the bounded spins end it with the timeout flag
    if (h->tune.debug_ghost_leaf >= 0 && h->coop_leaf_seq == h->tune.debug_ghost_leaf && la.p.G < 63) la.p.G += 1;
    h->coop_leaf_seq++;
    const int ballast = h->tune.panel_ballast;
    if (local && !h->coop_launch) return launch_panel_local_variant<T, true>(h, la, rpw, ballast);
    la.stride = 1;
    la.sel = 0;
    return launch_panel_local_variant<T, false>(h, la, rpw, ballast);
}

template int launch_panel_local<pl_t>(Handle*, const PanelArgs<pl_t>&, int, int, int, int);
#if defined(RFLU_PL_F32)
int panel_local_resident_limit_f32(int num_cus)
{
    return std::min(panel_local_resident_limit_variant<float, false>(num_cus), panel_local_resident_limit_variant<float, true>(num_cus));
}
#else
int panel_local_resident_limit_f64(int num_cus)
{
    return std::min(panel_local_resident_limit_variant<double, false>(num_cus), panel_local_resident_limit_variant<double, true>(num_cus));
}
#endif
#endif  // !RFLU_PL_XCD

}  // namespace rflu
