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
#include "panel_common.hpp"

namespace rflu {

template <typename T>
struct LocalLds {
    T prow[2][NB];             // pivot rows of the last two steps (columns k+2.. valid), by step parity
    T crow[NB];                // staging of this workgroup's candidate row for the coalesced publish
    unsigned whi[PANEL_WAVES]; // per-wave candidate: integer key (high / low word) and row position
    unsigned wlo[PANEL_WAVES];
    unsigned wpos[PANEL_WAVES];
    unsigned win[2];           // pivot position, by step parity
    T scale[2];                // 1 / pivot (1 when the pivot is exactly zero)
    T wu[2];                   // the pivot row's entry in column k+1
    int dead;
    int rows[NB];
};

template <typename T>
struct LocalArgs {
    PanelArgs<T> p;
    int stride;     // !LOCAL: participants are the blocks with blockIdx % stride == sel
    int sel;
    int want_xcc;   // LOCAL: participants are the blocks running on this XCC
};

__device__ __forceinline__ unsigned hw_xcc_id()
{
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 0xfu;
}

// ---- wave-uniform copies of values that arrive in vector registers (arguments of a non-inlined function): addresses
// built from them stay in scalar registers, so buffer descriptors need no waterfall loop and loop/branch conditions are
// scalar
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ unsigned uni(unsigned x) { return (unsigned)__builtin_amdgcn_readfirstlane((int)x); }
template <typename P>
__device__ __forceinline__ P* uni(P* p)
{
    const u64 v = (u64)p;
    const unsigned lo = uni((unsigned)v), hi = uni((unsigned)(v >> 32));
    return (P*)(((u64)hi << 32) | (u64)lo);
}

// ---- pivot search on INTEGER keys.  For finite x > 0 (and +inf) the IEEE bit pattern of |x| orders like |x|, so the
// argmax of src/lu.jl:298-305 (strict '>' from 0: zeros and NaNs never beat anything, ties keep the lowest position) is a
// max over (bits, -position) with key 0 for zero / NaN entries.  Integer max has single-instruction DPP forms
// (v_max_u32_dpp) where the Float64 version needs two moves and a v_max_f64 per stage.
template <typename T>
struct IKey;
template <>
struct IKey<double> {
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
struct IKey<float> {
    static constexpr bool TWO = false;
    static __device__ __forceinline__ void split(float v, bool cand, unsigned& hi, unsigned& lo)
    {
        const bool ok = cand && (__builtin_fabsf(v) > 0.0f);
        hi = ok ? (__float_as_uint(v) & 0x7fffffffu) : 0u;
        lo = 0u;
    }
};

// 64-lane reductions with the gfx9 row-broadcast DPP modes: after the four intra-row stages row_bcast15 / row_bcast31
// carry the row results upward; lane 63 holds the total
template <int CTRL, int ROWMASK>
__device__ __forceinline__ unsigned dpp_keep(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROWMASK, 0xF, false);
}
__device__ __forceinline__ unsigned wave_max_b(unsigned v)
{
    v = max(v, dpp_keep<0xB1, 0xF>(v));
    v = max(v, dpp_keep<0x4E, 0xF>(v));
    v = max(v, dpp_keep<0x141, 0xF>(v));
    v = max(v, dpp_keep<0x140, 0xF>(v));
    v = max(v, dpp_keep<0x142, 0xA>(v));
    v = max(v, dpp_keep<0x143, 0xC>(v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_min_b(unsigned v)
{
    v = min(v, dpp_keep<0xB1, 0xF>(v));
    v = min(v, dpp_keep<0x4E, 0xF>(v));
    v = min(v, dpp_keep<0x141, 0xF>(v));
    v = min(v, dpp_keep<0x140, 0xF>(v));
    v = min(v, dpp_keep<0x142, 0xA>(v));
    v = min(v, dpp_keep<0x143, 0xC>(v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// Every lane passes (hi, lo, pos); pos == POS_NONE marks a lane without a candidate (its hi / lo must be 0).
// Returns wave-uniform: the best key in (hi, lo), its position in pos (POS_NONE: no candidate at all) and the lane that
// holds it (0 if none).
template <bool TWO>
__device__ __forceinline__ int wave_argmax_i(unsigned& hi, unsigned& lo, unsigned& pos)
{
    const unsigned mh = wave_max_b(hi);
    bool hit = (hi == mh) && (pos != POS_NONE);
    unsigned ml = 0u;
    if (TWO) {
        ml = wave_max_b((hi == mh) ? lo : 0u);
        hit = hit && (lo == ml);
    }
    u64 mask = __ballot(hit);
    unsigned p = POS_NONE;
    int wl = 0;
    if (mask != 0) {
        if (__popcll(mask) != 1) {   // exact ties: the lowest position among the lanes holding the maximum
            p = wave_min_b(hit ? pos : POS_NONE);
            mask = __ballot(hit && pos == p);
        }
        wl = __ffsll((long long)mask) - 1;
        p = (unsigned)__builtin_amdgcn_readlane((int)pos, wl);
    }
    hi = mh;
    lo = ml;
    pos = p;
    return wl;
}

// wave-level part of the column search: this wave's best (key, pos) goes to LDS
template <typename T>
__device__ __forceinline__ void local_front_wave(LocalLds<T>* sh, T aval, unsigned pos, bool act, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    unsigned hi, lo, p = act ? pos : POS_NONE;
    IKey<T>::split(aval, act, hi, lo);
    wave_argmax_i<IKey<T>::TWO>(hi, lo, p);
    if (lane == 0) { sh->whi[wave] = hi; sh->wlo[wave] = lo; sh->wpos[wave] = p; }
}

// after the barrier: 1 = this thread owns the workgroup's candidate row, 2 = (thread 0) no candidate at all, 0 otherwise.
// Lane l looks at the record of wave l & 7 (duplicates change nothing), one more wave reduction picks the winner.
template <typename T>
__device__ __forceinline__ int local_front_combine(LocalLds<T>* sh, unsigned pos, bool act, int tid)
{
    const int r = tid & (PANEL_WAVES - 1);
    unsigned hi = sh->whi[r], lo = sh->wlo[r], cp = sh->wpos[r];
    wave_argmax_i<IKey<T>::TWO>(hi, lo, cp);
    if (act && pos == cp) return 1;
    if (cp == POS_NONE && tid == 0) return 2;
    return 0;
}

// the candidate row staged in LDS (columns kc+2..) leaves as ONE coalesced store of the candidate's wave
template <typename T, int AUX>
__device__ __noinline__ void local_publish_row(LocalLds<T>* sh, u64* scratch, unsigned epoch, int kc, int g, int lane)
{
    scratch = uni(scratch);
    epoch = uni(epoch);
    kc = uni(kc);
    g = uni(g);
    if (lane >= kc + 2 && lane < NB) {
        const T v = sh->crow[lane];
        const unsigned roff = (unsigned)(kc & 1) * PS_BUF_BYTES + PS_HDR_REGION + (unsigned)g * PS_ROW_BYTES;
        Gran<T>::template store<AUX>(scratch_rsrc(scratch), roff + (unsigned)lane * PS_VAL_BYTES, epoch + (unsigned)kc, v);
    }
}

// header of column kc: {position, a[kc], a[kc+1]} of the workgroup's candidate, or an empty header
template <typename T, int AUX>
__device__ __noinline__ void local_publish_header(u64* scratch, unsigned epoch, int kc, int g, unsigned pos, T a0, T a1)
{
    scratch = uni(scratch);
    epoch = uni(epoch);
    kc = uni(kc);
    g = uni(g);
    Gran<T>::template store_hdr3<AUX>(scratch_rsrc(scratch), (unsigned)(kc & 1) * PS_BUF_BYTES + (unsigned)g * PS_HDR_BYTES,
                                      epoch + (unsigned)kc, pos, a0, a1);
}

// One column, everything that needs no static register index.
//   wave 0: poll the G headers of column k (lane x = header x), reduce them to the pivot, request the pivot row, divide,
//           hand {pos, 1/pivot, u} over in LDS
//   barrier A
//   all   : bookkeeping for the thread's row, column k+1 brought up to date, wave-level search of column k+1
//   wave 0: the pivot row (columns k+2..) lands in LDS
//   barrier B
//   all   : workgroup candidate of column k+1
// flags: bit0 apply the update to this row, bit1 row still active, bit2 give up, bit3 this thread owns the workgroup's
// candidate row for the NEXT column, bit4 (thread 0) the workgroup has no candidate for the next column
template <typename T>
__device__ __noinline__ MidOut<T> local_mid(LocalLds<T>* sh, u64* scratch, int64_t* info, int64_t* ipiv, unsigned epoch,
                                            int G, int k, int w, int r0, int g, int tid, T ak, T ak1, unsigned pos, bool act)
{
    scratch = uni(scratch);
    info = uni(info);
    ipiv = uni(ipiv);
    epoch = uni(epoch);
    G = uni(G);
    k = uni(k);
    w = uni(w);
    r0 = uni(r0);
    g = uni(g);
    const int lane = tid & 63, wave = uni(tid >> 6);
    const int par = k & 1;
    const __amdgpu_buffer_rsrc_t rs = scratch_rsrc(scratch);
    const unsigned tag = epoch + (unsigned)k;
    const unsigned base = (unsigned)par * PS_BUF_BYTES;
    RFLU_STAMP(scratch, k, 0, g, tid);
    typename Gran<T>::raw_t raw;
    bool want_row = false;
    unsigned roff = 0;
    if (wave == 0) {
        bool timed_out = false;
        unsigned xp = POS_NONE;
        T xa = T(0), xu = T(0);
        if (lane < G) {
            int spins = 0;
            for (;;) {
                asm volatile("" ::: "memory");  // plain buffer intrinsics: keep the loads inside the loop
                if (Gran<T>::load_hdr3(rs, base + (unsigned)lane * PS_HDR_BYTES, tag, xp, xa, xu)) break;
                if (++spins > SPIN_LIMIT) { timed_out = true; xp = POS_NONE; break; }
            }
        }
        unsigned hi, lo, gp = xp;
        IKey<T>::split(xa, xp != POS_NONE, hi, lo);
        const int wl = wave_argmax_i<IKey<T>::TWO>(hi, lo, gp);   // the winner's lane is its workgroup index
        const T ga = readlane_val(xa, wl), gu = readlane_val(xu, wl);
        if (__any(timed_out)) {
            gp = POS_NONE;
            if (lane == 0) {
                __hip_atomic_fetch_or((u64*)(info + 1), (u64)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sh->dead = 1;
            }
        }
        if (gp != POS_NONE && lane >= k + 2 && lane < NB) {  // the pivot row: requested now, looked at after the search
            roff = base + PS_HDR_REGION + (unsigned)wl * PS_ROW_BYTES + (unsigned)lane * PS_VAL_BYTES;
            raw = Gran<T>::load_raw(rs, roff);
            want_row = true;
        }
        const T sc = (ga != T(0)) ? T(1) / ga : T(1);   // once per workgroup instead of once per thread
        if (lane == 0) {
            sh->win[par] = gp;
            sh->scale[par] = sc;
            sh->wu[par] = gu;
            if (g == 0 && gp != POS_NONE) {
                ipiv[r0 + k] = (int64_t)gp + 1;
                if (ga == T(0) && info[0] == 0) info[0] = (int64_t)r0 + k + 1;
            }
        }
    }
    barrier_lds_only();   // A: the pivot is known to every wave (wave 0's row request stays in flight)
    RFLU_STAMP(scratch, k, 1, g, tid);
    const unsigned win_pos = sh->win[par];
    MidOut<T> o;
    o.scale = sh->scale[par];
    o.pos = pos;
    o.flags = act ? 2u : 0u;
    T a1 = ak1;
    if (win_pos != POS_NONE) {
        const unsigned kpos = (unsigned)(r0 + k);
        if (act) {
            if (pos == win_pos) {
                o.pos = kpos;      // pivot row: final position r0+k, no further updates
                o.flags &= ~2u;
            } else {
                if (pos == kpos) o.pos = win_pos;  // displaced row takes the pivot's old position
                o.flags |= 1u;
                a1 = ak1 - (ak * o.scale) * sh->wu[par];   // column k+1 is current before the row arrives
            }
        }
    }
    const bool more = k + 1 < w;   // workgroup-uniform
    if (more) local_front_wave<T>(sh, a1, o.pos, (o.flags & 2u) != 0, tid);
    RFLU_STAMP(scratch, k, 2, g, tid);
    if (wave == 0 && want_row) {
        T xv = T(0);
        if (!Gran<T>::unpack(raw, tag, xv)) {
            int spins = 0;
            for (;;) {
                asm volatile("" ::: "memory");
                if (Gran<T>::load(rs, roff, tag, xv)) break;
                if (++spins > SPIN_LIMIT) {
                    __hip_atomic_fetch_or((u64*)(info + 1), (u64)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    sh->dead = 1;
                    break;
                }
            }
        }
        sh->prow[par][lane] = xv;
    }
    barrier_lds_only();   // B: wave records of column k+1 and the pivot row of column k are in LDS
    RFLU_STAMP(scratch, k, 3, g, tid);
    if (more) {
        const int f = local_front_combine<T>(sh, o.pos, (o.flags & 2u) != 0, tid);
        if (f == 1) o.flags |= 8u;
        if (f == 2) o.flags |= 16u;
    }
    if (sh->dead) o.flags |= 4u;
    RFLU_STAMP(scratch, k, 4, g, tid);
    return o;
}

template <typename T, int K, int AUX>
__device__ __forceinline__ void local_step(const PanelArgs<T>& p, LocalLds<T>* sh, T (&a)[NB], unsigned& pos, bool& act,
                                           bool& dead, PermState& perm, int g, int tid)
{
    if (K >= p.w || dead) return;
    T ak1 = T(0);
    if constexpr (K + 1 < NB) ak1 = a[K + 1];
    const MidOut<T> o = local_mid<T>(sh, p.scratch, p.info, p.ipiv, p.epoch, p.G, K, p.w, p.r0, g, tid, a[K], ak1, pos, act);
    pos = o.pos;
    act = (o.flags & 2u) != 0;
    dead = (o.flags & 4u) != 0;
    if (dead) return;
    if (g == 0 && (tid >> 6) == PANEL_WAVES - 1) {
        const unsigned wp = sh->win[K & 1];
        if (wp != POS_NONE) perm_state_step(perm, p.r0, K, __builtin_amdgcn_readfirstlane((int)wp), tid & 63);
    }
    const T* prow = sh->prow[K & 1];
    const bool upd = (o.flags & 1u) != 0;
    const bool more = K + 1 < p.w;                      // workgroup-uniform
    const bool cand = more && (o.flags & 8u) != 0;      // this row is the workgroup's candidate for column K+1
    T l = T(0);
    if (upd) {
        l = a[K] * o.scale;  // reciprocal-multiply (src/lu.jl:317-320); scale == 1 after a zero pivot
        a[K] = l;
        if constexpr (K + 1 < NB) a[K + 1] -= l * sh->wu[K & 1];
        if constexpr (K + 2 < NB) a[K + 2] -= l * prow[K + 2];
    }
    if constexpr (K + 1 < NB) {
        // the header of column K+1 leaves before the long update loop
        T un = T(0);
        if constexpr (K + 2 < NB) un = a[K + 2];
        if (cand) local_publish_header<T, AUX>(p.scratch, p.epoch, K + 1, g, pos, a[K + 1], un);
        else if (more && (o.flags & 16u)) local_publish_header<T, AUX>(p.scratch, p.epoch, K + 1, g, POS_NONE, T(0), T(0));
        RFLU_STAMP(p.scratch, K, 6, g, tid);
    }
    if constexpr (K + 3 < NB) {
        if (upd) {
#pragma unroll
            for (int j = K + 3; j < NB; ++j) a[j] -= l * prow[j];
        }
        if (cand) {
#pragma unroll
            for (int j = K + 3; j < NB; ++j) sh->crow[j] = a[j];
        }
        if (__ballot(cand) != 0) local_publish_row<T, AUX>(sh, p.scratch, p.epoch, K + 1, g, tid & 63);
    }
    RFLU_STAMP(p.scratch, K, 5, g, tid);
}

template <typename T, int K0, int K1, int AUX>
struct LocalSteps {
    static __device__ __forceinline__ void run(const PanelArgs<T>& p, LocalLds<T>* sh, T (&a)[NB], unsigned& pos,
                                               bool& act, bool& dead, PermState& perm, int g, int tid)
    {
        if constexpr (K0 < K1) {
            local_step<T, K0, AUX>(p, sh, a, pos, act, dead, perm, g, tid);
            LocalSteps<T, K0 + 1, K1, AUX>::run(p, sh, a, pos, act, dead, perm, g, tid);
        }
    }
};

template <typename T, bool LOCAL>
__global__ void __launch_bounds__(PANEL_THREADS) panel_pivot_local_kernel(LocalArgs<T> la)
{
    constexpr int AUX = LOCAL ? 0 : AUX_SC1;
    // LOCAL: the participants are the blocks that RUN on the chosen XCD.  A launch spreads its blocks round-robin over the
    // 8 XCDs (block b -> XCD (b + offset) % 8, the offset varies from launch to launch), so exactly one block of every
    // group of 8 qualifies and blockIdx / 8 numbers the participants 0..G-1.  Should the hardware ever place blocks
    // differently a participant is missing and the bounded spins end the launch with the timeout flag.
    if (LOCAL ? ((int)hw_xcc_id() != la.want_xcc) : ((int)(blockIdx.x % (unsigned)la.stride) != la.sel)) return;
    const PanelArgs<T>& p = la.p;
    __shared__ LocalLds<T> s_lds;
    LocalLds<T>* const sh = &s_lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = (int)(blockIdx.x / (unsigned)la.stride);
    const int row = p.r0 + g * PANEL_THREADS + tid;
    bool act = row < p.m;
    unsigned pos = act ? (unsigned)row : POS_NONE;
    if (tid == 0) sh->dead = 0;
    T a[NB];
    load_row_direct<T>(p.R, p.ld, row, act, p.c0, p.w, a);
    {   // column 0: search and publish
        local_front_wave<T>(sh, a[0], pos, act, tid);
        __syncthreads();
        const int f = local_front_combine<T>(sh, pos, act, tid);
        if (f == 1) {
            local_publish_header<T, AUX>(p.scratch, p.epoch, 0, g, pos, a[0], a[1]);
#pragma unroll
            for (int j = 2; j < NB; ++j) sh->crow[j] = a[j];
        }
        if (f == 2) local_publish_header<T, AUX>(p.scratch, p.epoch, 0, g, POS_NONE, T(0), T(0));
        if (__ballot(f == 1) != 0) local_publish_row<T, AUX>(sh, p.scratch, p.epoch, 0, g, lane);
    }
    bool dead = false;
    PermState perm = perm_state_init(lane);
    LocalSteps<T, 0, NB, AUX>::run(p, sh, a, pos, act, dead, perm, g, tid);
    store_row_direct<T>(p.R, p.ld, pos, p.c0, p.w, a);
    __syncthreads();
    if (g == 0 && wave == PANEL_WAVES - 1) {
        const int chunk = p.r0 / NB;
        perm_state_finish(perm, p.r0, lane, sh->rows, p.pm_cnt + chunk, p.pm_dst + (size_t)chunk * 2 * NB,
                          p.pm_src + (size_t)chunk * 2 * NB);
    }
}

// Launch the leaf on the blocks b with b % stride == sel of a grid of G*stride workgroups.  local != 0: plain-store
// records (all participants must share an XCD: stride 8); local == 0: sc1 records, any placement.
template <typename T>
int launch_panel_local(Handle* h, const PanelArgs<T>& p, int stride, int sel, int want_xcc, int local)
{
    LocalArgs<T> la;
    la.p = p;
    la.stride = stride;
    la.sel = sel;
    la.want_xcc = want_xcc;
    const dim3 grid((unsigned)(p.G * stride));
    if (local) hipLaunchKernelGGL((panel_pivot_local_kernel<T, true>), grid, dim3(PANEL_THREADS), 0, h->stream, la);
    else hipLaunchKernelGGL((panel_pivot_local_kernel<T, false>), grid, dim3(PANEL_THREADS), 0, h->stream, la);
    RFLU_HIP(hipGetLastError());
    return RFLU_OK;
}

#ifdef RFLU_PANEL_F32_TU
template int launch_panel_local<float>(Handle*, const PanelArgs<float>&, int, int, int, int);
#else
template int launch_panel_local<double>(Handle*, const PanelArgs<double>&, int, int, int, int);
#endif

}  // namespace rflu
