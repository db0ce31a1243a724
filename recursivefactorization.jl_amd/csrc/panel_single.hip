// panel_single.hip -- pivoted leaf panel of at most 512 rows in ONE workgroup: nothing leaves the CU between the first
// load and the last store.
//
// Same semantics as _generic_lufact! (/root/reference/src/lu.jl:290-338) and the same arithmetic per entry as the
// cooperative leaves (panel.hip, panel_local.hip): argmax |a_ik| with strict '>' from 0 and lowest position on ties
// (:298-305), interchange by position renaming, reciprocal-multiply scaling (:317-320), zero pivot -> info once, keep
// updating (:321-334).  Every entry receives the multiply-adds of the unblocked algorithm in the same order, so factors and
// pivots are bit-identical to those kernels.
//
// Why a kernel of its own: the cooperative leaf publishes every workgroup's candidate through global memory and polls
// it back (a write-through store, two dependent round trips and two workgroup barriers per column) even when there is no
// peer: 1.9 us per column for a lone workgroup.  Here a column costs ONE workgroup barrier:
//   * no communication wave: after barrier(c) every wave reads the PW wave records of column c from LDS and reduces them
//     redundantly (3 DPP stages for <= 8 records), divides once per wave, and goes on;
//   * the same one-elimination lag as panel_local.hip keeps the pivot row off the chain: a wave record carries
//     {key, position, a_c, a_{c+1} lagging one elimination, l_{c-1}}, every wave finishes u_{c,c+1} itself, and the
//     pivot row P_c (entries >= c+2) is written to LDS by its owner AFTER its record for column c+1 has left, to be read
//     one barrier later.
// Roofline: latency -- w x (one barrier + one LDS round trip + ~100 dependent wave instructions); m*w^2 flops reported.
#include "panel_common.hpp"

namespace rflu {

namespace {

__device__ __forceinline__ int s_uni(int x) { return __builtin_amdgcn_readfirstlane(x); }

template <typename T>
struct SKey;
template <>
struct SKey<double> {
    static constexpr bool TWO = true;
    static __device__ __forceinline__ void split(double v, bool cand, unsigned& hi, unsigned& lo)
    {
        const u64 b = (u64)__double_as_longlong(v);
        const bool ok = cand && (__builtin_fabs(v) > 0.0);   // false for 0 and NaN
        hi = ok ? ((unsigned)(b >> 32) & 0x7fffffffu) : 0u;
        lo = ok ? (unsigned)b : 0u;
    }
};
template <>
struct SKey<float> {
    static constexpr bool TWO = false;
    static __device__ __forceinline__ void split(float v, bool cand, unsigned& hi, unsigned& lo)
    {
        const bool ok = cand && (__builtin_fabsf(v) > 0.0f);
        hi = ok ? (__float_as_uint(v) & 0x7fffffffu) : 0u;
        lo = 0u;
    }
};

// max / min over the whole wave (6 DPP stages, result in lane 63) and over lanes 0..7 (3 stages, result in lanes 0..7)
__device__ __forceinline__ unsigned s_max64(unsigned v)
{
    asm("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v));
    asm("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(v));
    asm("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(v));
    asm("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(v));
    asm("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(v));
    asm("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(v));
    asm("s_nop 1" : "+v"(v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned s_min64(unsigned v)
{
    asm("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v));
    asm("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(v));
    asm("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(v));
    asm("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(v));
    asm("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(v));
    asm("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(v));
    asm("s_nop 1" : "+v"(v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned s_max8(unsigned v)
{
    asm("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v));
    asm("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(v));
    asm("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(v));
    asm("s_nop 1" : "+v"(v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 0);
}
__device__ __forceinline__ unsigned s_min8(unsigned v)
{
    asm("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v));
    asm("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(v));
    asm("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(v));
    asm("s_nop 1" : "+v"(v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 0);
}

// Argmax of (hi, lo) descending, pos ascending over the lanes with pos != POS_NONE (their hi / lo must be 0 otherwise).
// EIGHT: only lanes 0..7 carry entries (the others pass hi = lo = 0, pos = POS_NONE).  Returns the winning lane (0 if there
// is no candidate), wave-uniform; hi / lo / pos become the winner's.  One reduction in the common case (the high words of the
// maximum almost never coincide), the low words and the positions only on collisions.
template <bool TWO, bool EIGHT>
__device__ __forceinline__ int s_argmax(unsigned& hi, unsigned& lo, unsigned& pos)
{
    const unsigned mh = EIGHT ? s_max8(hi) : s_max64(hi);
    bool hit = (hi == mh) && (pos != POS_NONE);
    u64 mask = __ballot(hit);
    unsigned p = POS_NONE, ml = 0u;
    int wl = 0;
    if (mask != 0) {
        if (__popcll(mask) != 1) {
            if (TWO) {
                ml = EIGHT ? s_max8(hit ? lo : 0u) : s_max64(hit ? lo : 0u);
                hit = hit && (lo == ml);
                mask = __ballot(hit);
            }
            if (__popcll(mask) != 1) {   // exact ties: the lowest position among the lanes holding the maximum
                p = EIGHT ? s_min8(hit ? pos : POS_NONE) : s_min64(hit ? pos : POS_NONE);
                mask = __ballot(hit && pos == p);
            }
        }
        wl = __ffsll((long long)mask) - 1;
        p = (unsigned)__builtin_amdgcn_readlane((int)pos, wl);
        ml = (unsigned)__builtin_amdgcn_readlane((int)lo, wl);
    }
    hi = mh;
    lo = ml;
    pos = p;
    return wl;
}

template <typename T>
struct SRec {              // a wave's candidate for one column, 16-byte pieces (three LDS accesses of the owning lane)
    unsigned hi, lo, pos, pad;
    T a1, a2;              // a_c (complete), a_{c+1} (misses elimination c-1)
    T l, rinv;             // l_{c-1} of that row; 1 / a_c (1 for a zero entry): every lane divides for its own entry while the
};                         // wave's search runs, so the quotient never sits on the chain

template <typename T, int PW>
struct SLds {
    T prow[2][NB];         // P_c by parity of c (entries j >= c+2), written by the pivot's owner after its next record
    SRec<T> rec[2][PW];    // by parity of the column
    int rows[NB];
    unsigned piv[NB];      // the pivots' rows (flushed to ipiv by one coalesced store at the end: a global store per step on
    int zinfo;             // a wave of the chain would make that wave wait for its acknowledgement at the next vmcnt wait)
};

struct SState {
    unsigned pos;
    bool act;
    bool updprev;          // elimination c-1 still has to reach this row's entries j >= c+1
};

template <typename T, int PW>
__device__ __forceinline__ void s_record(SLds<T, PW>* sh, int par, int wave, int lane, T a1, T a2, T l, unsigned pos, bool act)
{
    unsigned hi, lo, p = act ? pos : POS_NONE;
    SKey<T>::split(a1, act, hi, lo);
    const T rinv = (a1 != T(0)) ? T(1) / a1 : T(1);
    unsigned mh = hi, ml = lo;
    const int wl = s_argmax<SKey<T>::TWO, false>(mh, ml, p);
    if (lane == wl) {   // the winning lane leaves the record itself (lane 0 an empty one if the wave has no candidate)
        SRec<T>* r = &sh->rec[par][wave];
        r->hi = mh;
        r->lo = ml;
        r->pos = p;
        r->a1 = a1;
        r->a2 = a2;
        r->l = l;
        r->rinv = rinv;
    }
}

// Step C >= 0: after barrier(C) the wave records of column C are in LDS.  C == -1 is the prologue (records of column 0).
// FULL: the leaf has all NB columns (every leaf but the last one of a matrix whose width is no multiple of NB): no run-time
// column tests, so the 64 unrolled steps are straight-line code between their barriers (with the tests the compiler lays the
// steps out as far-apart blocks joined by s_setpc jumps).
template <typename T, int C, int PW, bool FULL>
__device__ __forceinline__ void s_step(const PanelArgs<T>& p, SLds<T, PW>* sh, T (&a)[NB], T& lprev, SState& st,
                                       PermState& perm, int tid)
{
    if (!FULL && C >= p.w) return;   // workgroup-uniform
    const int lane = tid & 63, wave = s_uni(tid >> 6);
    T l = T(0);
    bool upd = false, owner = false;
    unsigned win = POS_NONE;
    if constexpr (C >= 0) RFLU_STAMP(p.scratch, C, 0, 0, tid);
    if constexpr (C >= 0) {
        // ---- every wave: the winner among the PW wave records of column C
        const int r = lane & (PW - 1);
        const bool has = lane < PW;
        const SRec<T>* rc = &sh->rec[C & 1][r];
        unsigned hi = rc->hi, lo = rc->lo, gp = rc->pos;
        const T ra1 = rc->a1, ra2 = rc->a2, rl = rc->l, rri = rc->rinv;
        T p1 = T(0), p2 = T(0);
        if constexpr (C >= 1) {
            if constexpr (C + 1 < NB) p1 = sh->prow[(C - 1) & 1][C + 1];
            if constexpr (C + 2 < NB) p2 = sh->prow[(C - 1) & 1][C + 2];
        }
        if (!has) { hi = 0u; lo = 0u; gp = POS_NONE; }
        const int wl = (PW == 1) ? 0 : s_argmax<SKey<T>::TWO, true>(hi, lo, gp);
        if (PW == 1) gp = (unsigned)__builtin_amdgcn_readlane((int)gp, 0);
        const T ga = readlane_val(ra1, wl), ga1 = readlane_val(ra2, wl), gl = readlane_val(rl, wl);
        T gu = ga1;
        if constexpr (C >= 1 && C + 1 < NB) gu = ga1 - gl * p1;   // u_{C,C+1}: the record's entry misses elimination C-1
        if constexpr (C >= 0) RFLU_STAMP(p.scratch, C, 1, 0, tid);
        const T sc = readlane_val(rri, wl);   // 1 / pivot (1 when the pivot is exactly zero)
        win = gp;
        if (tid == 0) {
            sh->piv[C] = gp;
            if (gp != POS_NONE && ga == T(0) && sh->zinfo == 0) sh->zinfo = p.r0 + C + 1;
        }
        // ---- own row: the two entries of elimination C-1 the next record needs, then the interchange and elimination C
        if constexpr (C >= 1) {
            if (st.updprev) {
                if constexpr (C + 1 < NB) a[C + 1] -= lprev * p1;
                if constexpr (C + 2 < NB) a[C + 2] -= lprev * p2;
            }
        }
        if (gp != POS_NONE && st.act) {
            const unsigned kpos = (unsigned)(p.r0 + C);
            if (st.pos == gp) {
                st.pos = kpos;      // pivot row: final position r0+C, no further updates
                st.act = false;
                owner = true;
            } else {
                if (st.pos == kpos) st.pos = gp;   // displaced row takes the pivot's old position
                upd = true;
                l = a[C] * sc;      // reciprocal-multiply (src/lu.jl:317-320); sc == 1 after a zero pivot
                a[C] = l;
                if constexpr (C + 1 < NB) a[C + 1] -= l * gu;
            }
        }
    }
    const bool more = FULL || C + 1 < p.w;   // workgroup-uniform
    if constexpr (C >= 0) RFLU_STAMP(p.scratch, C, 2, 0, tid);
    if constexpr (C + 1 < NB) {
        if (more) {
            T a2 = T(0);
            if constexpr (C + 2 < NB) a2 = a[C + 2];
            s_record<T, PW>(sh, (C + 1) & 1, wave, lane, a[C + 1], a2, l, st.pos, st.act);
        }
    }
    if constexpr (C >= 0) RFLU_STAMP(p.scratch, C, 3, 0, tid);
    // interchange bookkeeping of column C (the last wave), off the other waves' chain
    if constexpr (C >= 0) {
        if (wave == PW - 1 && win != POS_NONE)
            perm_state_step(perm, p.r0, C, __builtin_amdgcn_readfirstlane((int)win), lane);
    }
    if constexpr (C >= 0) RFLU_STAMP(p.scratch, C, 4, 0, tid);
    if constexpr (C >= 1 && C + 3 < NB) {
        if (st.updprev) {   // the rest of elimination C-1
            const T* P = sh->prow[(C - 1) & 1];
#pragma unroll
            for (int j = C + 3; j < NB; ++j) a[j] -= lprev * P[j];
        }
    }
    if constexpr (C >= 0) RFLU_STAMP(p.scratch, C, 5, 0, tid);
    if constexpr (C >= 0 && C + 2 < NB) {
        if (owner) {   // P_C, complete through elimination C-1, for the steps behind the next barrier
            T* P = sh->prow[C & 1];
#pragma unroll
            for (int j = C + 2; j < NB; ++j) P[j] = a[j];
        }
    }
    lprev = l;
    st.updprev = upd;
    if constexpr (C >= 0) RFLU_STAMP(p.scratch, C, 6, 0, tid);
    if constexpr (C + 1 < NB) {
        if (more) barrier_lds_only();   // barrier(C+1)
    }
}

template <typename T, int C0, int C1, int PW, bool FULL>
struct SSteps {
    static __device__ __forceinline__ void run(const PanelArgs<T>& p, SLds<T, PW>* sh, T (&a)[NB], T& lprev, SState& st,
                                               PermState& perm, int tid)
    {
        if constexpr (C0 < C1) {
            s_step<T, C0, PW, FULL>(p, sh, a, lprev, st, perm, tid);
            SSteps<T, C0 + 1, C1, PW, FULL>::run(p, sh, a, lprev, st, perm, tid);
        }
    }
};

}  // namespace

// PW = row waves: 64 * PW rows, one matrix row per thread in registers
template <typename T, int PW, bool FULL>
__global__ void __launch_bounds__(PW * 64) panel_single_kernel(PanelArgs<T> p)
{
    __shared__ SLds<T, PW> s_lds;
    SLds<T, PW>* const sh = &s_lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = p.r0 + tid;
    SState st;
    st.act = row < p.m;
    st.pos = st.act ? (unsigned)row : POS_NONE;
    st.updprev = false;
    T a[NB];
    load_row_direct<T>(p.R, p.ld, row, st.act, p.c0, p.w, a);
    if (tid == 0) sh->zinfo = 0;
    // the row is in registers before the first step: a real s_waitcnt (which the compiler's wait-count pass sees), so that no
    // conservative vmcnt(0) is left at the control-flow merges inside the steps
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) only (gfx9 encoding: expcnt 7, lgkmcnt 15 = no wait)
    T lprev = T(0);
    PermState perm = perm_state_init(lane);
    SSteps<T, -1, NB, PW, FULL>::run(p, sh, a, lprev, st, perm, tid);
    store_row_direct<T>(p.R, p.ld, st.pos, p.c0, p.w, a);
    if (wave == 0) {   // wave 0 wrote piv / zinfo itself (LDS operations of one wave stay in order)
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
        if (lane < p.w) {
            const unsigned gp = sh->piv[lane];
            if (gp != POS_NONE) p.ipiv[p.r0 + lane] = (int64_t)gp + 1;
        }
        if (lane == 0 && sh->zinfo != 0 && p.info[0] == 0) p.info[0] = (int64_t)sh->zinfo;
    }
    if (wave == PW - 1) {
        const int chunk = p.r0 / NB;
        perm_state_finish(perm, p.r0, lane, sh->rows, p.pm_cnt + chunk, p.pm_dst + (size_t)chunk * 2 * NB,
                          p.pm_src + (size_t)chunk * 2 * NB);
    }
}

template <typename T>
int launch_panel_single(Handle* h, const PanelArgs<T>& p)
{
    const int64_t rows = (int64_t)p.m - p.r0;
    if (p.w != NB) hipLaunchKernelGGL((panel_single_kernel<T, 8, false>), dim3(1), dim3(512), 0, h->stream, p);   // partial leaf (rare)
    else if (rows <= 64) hipLaunchKernelGGL((panel_single_kernel<T, 1, true>), dim3(1), dim3(64), 0, h->stream, p);
    else if (rows <= 128) hipLaunchKernelGGL((panel_single_kernel<T, 2, true>), dim3(1), dim3(128), 0, h->stream, p);
    else if (rows <= 256) hipLaunchKernelGGL((panel_single_kernel<T, 4, true>), dim3(1), dim3(256), 0, h->stream, p);
    else hipLaunchKernelGGL((panel_single_kernel<T, 8, true>), dim3(1), dim3(512), 0, h->stream, p);
    RFLU_HIP(hipGetLastError());
    return RFLU_OK;
}

#ifdef RFLU_PANEL_F32_TU
template int launch_panel_single<float>(Handle*, const PanelArgs<float>&);
#else
template int launch_panel_single<double>(Handle*, const PanelArgs<double>&);
#endif

}  // namespace rflu
