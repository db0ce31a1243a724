// panel_common.hpp -- pieces shared by the leaf panel kernels (panel.hip, panel_local.hip): tagged-granule records,
// row <-> register transfer, interchange bookkeeping, wave-wide argmax.  See panel.hip for the design notes.
#pragma once
#include "rflu_internal.hpp"

namespace rflu {


typedef unsigned long long u64;
constexpr unsigned POS_NONE = 0x7fffffffu;
constexpr int SPIN_LIMIT = 1 << 20;
constexpr int TILE_LD = NB + 1;
constexpr int PANEL_WAVES = PANEL_THREADS / 64;
static_assert(PANEL_WAVES == 8, "the workgroup-winner trees in step_a are written for 8 waves");

// ---- cross-workgroup records: data-tagged granules ------------------------------------------------------------------
// A granule is 8 bytes {payload dword, tag dword}; a Float64 value travels as two granules written by ONE 16-byte
// write-through store (buffer_store_dwordx4 ... sc1) and read by one 16-byte sc1 load; the reader accepts a value only
// when both tags equal the step's tag, so no flag, fence or barrier orders the exchange
// (cdna_hip_programming.md Guideline 16, form R2).
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
constexpr int AUX_SC1 = 16;

constexpr unsigned PS_HDR_BYTES = 64;                       // per workgroup: {pos,tag | a_pk granule(s) | next-column granule}
constexpr unsigned PS_VAL_BYTES = 16;                       // per row value (Float32 uses the first 8)
constexpr unsigned PS_ROW_BYTES = NB * PS_VAL_BYTES;        // per workgroup candidate row
constexpr unsigned PS_HDR_REGION = MAX_PANEL_WGS * PS_HDR_BYTES;
constexpr unsigned PS_BUF_BYTES = PS_HDR_REGION + MAX_PANEL_WGS * PS_ROW_BYTES;  // one parity buffer
constexpr size_t PS_TOTAL_WORDS = 2 * (size_t)PS_BUF_BYTES / 8;
constexpr size_t PS_TRACE_WORDS = 8 * NB + 16;              // RFLU_PANEL_TRACE stamps live right after the records
// pair leaf (two leaves in one launch): slot k = {L11 row k | raw leaf-B row of pivot k}, 2*NB granules each
constexpr unsigned PX_SLOT_BYTES = 2 * NB * PS_VAL_BYTES;
constexpr unsigned PX_BYTES = NB * PX_SLOT_BYTES;
constexpr size_t PX_OFFSET_WORDS = PS_TOTAL_WORDS + PS_TRACE_WORDS;
constexpr int PX_UL = NB + 1;                                          // LDS row stride of U12 (bank spread)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t scratch_rsrc(u64* scratch)
{
    return __builtin_amdgcn_make_buffer_rsrc(scratch, 0, 2 * PS_BUF_BYTES, 0x00020000);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t pair_rsrc(u64* scratch)
{
    return __builtin_amdgcn_make_buffer_rsrc(scratch + PX_OFFSET_WORDS, 0, PX_BYTES, 0x00020000);
}

template <typename T>
struct Gran;
template <>
struct Gran<double> {
    template <int AUX = AUX_SC1>
    static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned tag, double v) {
        const u64 b = (u64)__double_as_longlong(v);
        const u4v x = {(unsigned)(b >> 32), tag, (unsigned)b, tag};
        __builtin_amdgcn_raw_buffer_store_b128(x, r, off, 0, AUX);
    }
    static __device__ __forceinline__ bool load(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned tag, double& v) {
        const u4v x = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, AUX_SC1);
        v = __longlong_as_double((long long)(((u64)x[0] << 32) | (u64)x[2]));
        return x[1] == tag && x[3] == tag;
    }
    // header of the pipelined kernel: position, pivot candidate a and the candidate row's NEXT-column entry u
    template <int AUX = AUX_SC1>
    static __device__ __forceinline__ void store_hdr3(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned tag, unsigned pos, double a, double u) {
        const u64 b = (u64)__double_as_longlong(a), c = (u64)__double_as_longlong(u);
        const u4v x = {pos, tag, (unsigned)(b >> 32), tag};
        const u4v y = {(unsigned)b, tag, (unsigned)(c >> 32), tag};
        const u4v z = {(unsigned)c, tag, 0u, 0u};
        __builtin_amdgcn_raw_buffer_store_b128(x, r, off, 0, AUX);
        __builtin_amdgcn_raw_buffer_store_b128(y, r, off + 16, 0, AUX);
        __builtin_amdgcn_raw_buffer_store_b128(z, r, off + 32, 0, AUX);
    }
    static __device__ __forceinline__ bool load_hdr3(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned tag, unsigned& pos, double& a, double& u) {
        const u4v x = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, AUX_SC1);
        const u4v y = __builtin_amdgcn_raw_buffer_load_b128(r, off + 16, 0, AUX_SC1);
        const u4v z = __builtin_amdgcn_raw_buffer_load_b128(r, off + 32, 0, AUX_SC1);
        pos = x[0];
        a = __longlong_as_double((long long)(((u64)x[2] << 32) | (u64)y[0]));
        u = __longlong_as_double((long long)(((u64)y[2] << 32) | (u64)z[0]));
        return x[1] == tag && x[3] == tag && y[1] == tag && y[3] == tag && z[1] == tag;
    }
    typedef u4v raw_t;
    static __device__ __forceinline__ raw_t load_raw(__amdgpu_buffer_rsrc_t r, unsigned off) {
        return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, AUX_SC1);
    }
    static __device__ __forceinline__ bool unpack(raw_t x, unsigned tag, double& v) {
        v = __longlong_as_double((long long)(((u64)x[0] << 32) | (u64)x[2]));
        return x[1] == tag && x[3] == tag;
    }
    static __device__ __forceinline__ void store_hdr(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned tag, unsigned pos, double a) {
        const u64 b = (u64)__double_as_longlong(a);
        const u4v x = {pos, tag, (unsigned)(b >> 32), tag};
        const u4v y = {(unsigned)b, tag, 0u, 0u};
        __builtin_amdgcn_raw_buffer_store_b128(x, r, off, 0, AUX_SC1);
        __builtin_amdgcn_raw_buffer_store_b128(y, r, off + 16, 0, AUX_SC1);
    }
    static __device__ __forceinline__ bool load_hdr(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned tag, unsigned& pos, double& a) {
        const u4v x = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, AUX_SC1);
        const u4v y = __builtin_amdgcn_raw_buffer_load_b128(r, off + 16, 0, AUX_SC1);
        pos = x[0];
        a = __longlong_as_double((long long)(((u64)x[2] << 32) | (u64)y[0]));
        return x[1] == tag && x[3] == tag && y[1] == tag;
    }
};
template <>
struct Gran<float> {
    template <int AUX = AUX_SC1>
    static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned tag, float v) {
        unsigned bits = __float_as_uint(v);
        // opaque to the optimiser: otherwise {a[j], tag} is built from a <2 x i32> load of (a[j], a[j+1]) out of the row
        // array, and those overlapping vector loads keep the whole row in scratch memory instead of registers
        asm volatile("" : "+v"(bits));
        const u2v x = {bits, tag};
        __builtin_amdgcn_raw_buffer_store_b64(x, r, off, 0, AUX);
    }
    static __device__ __forceinline__ bool load(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned tag, float& v) {
        const u2v x = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, AUX_SC1);
        v = __uint_as_float(x[0]);
        return x[1] == tag;
    }
    template <int AUX = AUX_SC1>
    static __device__ __forceinline__ void store_hdr3(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned tag, unsigned pos, float a, float u) {
        const u4v x = {pos, tag, __float_as_uint(a), tag};
        const u4v y = {__float_as_uint(u), tag, 0u, 0u};
        __builtin_amdgcn_raw_buffer_store_b128(x, r, off, 0, AUX);
        __builtin_amdgcn_raw_buffer_store_b128(y, r, off + 16, 0, AUX);
    }
    static __device__ __forceinline__ bool load_hdr3(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned tag, unsigned& pos, float& a, float& u) {
        const u4v x = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, AUX_SC1);
        const u4v y = __builtin_amdgcn_raw_buffer_load_b128(r, off + 16, 0, AUX_SC1);
        pos = x[0];
        a = __uint_as_float(x[2]);
        u = __uint_as_float(y[0]);
        return x[1] == tag && x[3] == tag && y[1] == tag;
    }
    typedef u2v raw_t;
    static __device__ __forceinline__ raw_t load_raw(__amdgpu_buffer_rsrc_t r, unsigned off) {
        return __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, AUX_SC1);
    }
    static __device__ __forceinline__ bool unpack(raw_t x, unsigned tag, float& v) {
        v = __uint_as_float(x[0]);
        return x[1] == tag;
    }
    static __device__ __forceinline__ void store_hdr(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned tag, unsigned pos, float a) {
        const u4v x = {pos, tag, __float_as_uint(a), tag};
        __builtin_amdgcn_raw_buffer_store_b128(x, r, off, 0, AUX_SC1);
    }
    static __device__ __forceinline__ bool load_hdr(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned tag, unsigned& pos, float& a) {
        const u4v x = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, AUX_SC1);
        pos = x[0];
        a = __uint_as_float(x[2]);
        return x[1] == tag && x[3] == tag;
    }
};

// Optional step tracing (compile with -DRFLU_PANEL_TRACE; experiment builds only): thread 0 of workgroup 0 stores
// clock64() stamps into the panel scratch area past the granule records.
#if defined(RFLU_PANEL_TRACE_ALL)
// every workgroup's thread 0 (first 32 workgroups), wall clock (100 MHz, common to all CUs): skew between workgroups
#define RFLU_TRACE_ALL_WORDS (32 * (NB + 1) * 8)
#define RFLU_STAMP(scratch, k, i, g, tid) do { if ((g) < 32 && (tid) == 0) ((long long*)((scratch) + PX_OFFSET_WORDS + PX_BYTES / 8))[((g) * (NB + 1) + (k)) * 8 + (i)] = (long long)wall_clock64(); } while (0)
#define RFLU_STAMP_ANY(scratch, k, i, g) do { if ((g) < 32) ((long long*)((scratch) + PX_OFFSET_WORDS + PX_BYTES / 8))[((g) * (NB + 1) + (k)) * 8 + (i)] = (long long)wall_clock64(); } while (0)
#elif defined(RFLU_PANEL_TRACE)
#define RFLU_TRACE_ALL_WORDS 0
#define RFLU_STAMP_ANY(scratch, k, i, g) do { } while (0)
#define RFLU_STAMP(scratch, k, i, g, tid) do { if ((g) == 0 && (tid) == 0) ((long long*)((scratch) + PS_TOTAL_WORDS))[(k) * 8 + (i)] = clock64(); } while (0)
#else
#define RFLU_TRACE_ALL_WORDS 0
#define RFLU_STAMP_ANY(scratch, k, i, g) do { } while (0)
#define RFLU_STAMP(scratch, k, i, g, tid) do { } while (0)
#endif

template <typename T>
struct PanelArgs {
    T* R;
    int64_t ld;
    int m, r0, c0, w;
    int64_t* ipiv;   // global, 1-based entries written at [r0, r0+w)
    int64_t* info;   // [0] info, [1] error flag
    u64* scratch;
    unsigned epoch;  // first tag of this launch (w consecutive tags are used)
    int G;
    int* pm_cnt;     // chunk bookkeeping outputs for chunk r0/NB
    int* pm_dst;
    int* pm_src;
};

__device__ __forceinline__ double tabs(double x) { return __builtin_fabs(x); }
__device__ __forceinline__ float tabs(float x) { return __builtin_fabsf(x); }

template <typename T>
__device__ __forceinline__ bool better(T ov, unsigned op, T bv, unsigned bp)
{
    return ov > bv || (ov == bv && op < bp);
}

// Turn w sequential interchanges (position base+i <-> piv[i], piv global 0-based) into an equivalent list of row moves
// new[dst] = old[src].  One wave; rows/content are LDS scratch of 2*NB ints each.
__device__ void perm_build_wave(const int* piv, int base, int w, int lane, int* rows, int* content, int* out_cnt,
                                int* out_dst, int* out_src)
{
    rows[lane] = base + lane;
    content[lane] = lane;
    rows[NB + lane] = -1;
    content[NB + lane] = NB + lane;
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    int nextra = 0;
    for (int i = 0; i < w; ++i) {
        const int p = piv[i];
        int sp;
        if (p < base + NB) {
            sp = p - base;
        } else {
            const bool hit = (lane < nextra) && (rows[NB + lane] == p);
            const u64 mask = __ballot(hit);
            if (mask) {
                sp = NB + (__ffsll((long long)mask) - 1);
            } else {
                sp = NB + nextra;
                if (lane == 0) rows[sp] = p;
                ++nextra;
            }
        }
        __threadfence_block();
        if (lane == 0 && sp != i) {
            const int t = content[i];
            content[i] = content[sp];
            content[sp] = t;
        }
        __threadfence_block();
        __builtin_amdgcn_wave_barrier();
    }
    int total = 0;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int s = half * NB + lane;
        const bool moved = (half == 0 || lane < nextra) && (content[s] != s);
        const u64 mask = __ballot(moved);
        if (moved) {
            const int off = total + __popcll(mask & ((1ull << lane) - 1ull));
            out_dst[off] = rows[s];
            out_src[off] = rows[content[s]];
        }
        total += __popcll(mask);
    }
    if (lane == 0) *out_cnt = total;
}


// ---- row <-> register staging through an LDS transpose (coalesced 512-byte row segments on the memory side) -------
template <typename T, int RT>
__device__ __forceinline__ void load_rows(const T* R, int64_t ld, int row_base, int m, int c0, int w, T (&a)[RT][NB],
                                          T* tile, int wave, int lane)
{
#pragma unroll
    for (int q = 0; q < RT; ++q) {
#pragma unroll
        for (int chunk = 0; chunk < PANEL_WAVES; ++chunk) {
            const int rb = row_base + q * PANEL_THREADS + chunk * 64;
            for (int rr = wave; rr < 64; rr += PANEL_WAVES) {
                const int grow = rb + rr;
                T v = T(0);
                if (grow < m && lane < w) v = R[(int64_t)grow * ld + c0 + lane];
                tile[rr * TILE_LD + lane] = v;
            }
            __syncthreads();
            if (wave == chunk) {
#pragma unroll
                for (int j = 0; j < NB; ++j) a[q][j] = tile[lane * TILE_LD + j];
            }
            __syncthreads();
        }
    }
}

template <typename T, int RT>
__device__ __forceinline__ void store_rows(T* R, int64_t ld, int c0, int w, const T (&a)[RT][NB],
                                           const unsigned (&pos)[RT], T* tile, int* spos, int wave, int lane)
{
#pragma unroll
    for (int q = 0; q < RT; ++q) {
#pragma unroll
        for (int chunk = 0; chunk < PANEL_WAVES; ++chunk) {
            if (wave == chunk) {
#pragma unroll
                for (int j = 0; j < NB; ++j) tile[lane * TILE_LD + j] = a[q][j];
                spos[lane] = (pos[q] == POS_NONE) ? -1 : (int)pos[q];
            }
            __syncthreads();
            for (int rr = wave; rr < 64; rr += PANEL_WAVES) {
                const int p = spos[rr];
                if (p >= 0 && lane < w) R[(int64_t)p * ld + c0 + lane] = tile[rr * TILE_LD + lane];
            }
            __syncthreads();
        }
    }
}

// ---- direct row <-> register transfer: every thread moves its own row with 16-byte accesses, all in flight at once
// (one memory latency for the whole slab instead of 8 staged LDS round trips); each lane touches whole 128-byte lines.
template <typename T>
__device__ __forceinline__ void load_row_direct(const T* __restrict__ R, int64_t ld, int row, bool valid, int c0, int w,
                                                T (&a)[NB])
{
    constexpr int VW = 16 / (int)sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(VW)));
    const T* p = R + (int64_t)row * ld + c0;
    const bool vec_ok = ((reinterpret_cast<uintptr_t>(p) & 15) == 0);
    if (valid && vec_ok && w == NB) {
#pragma unroll
        for (int j = 0; j < NB; j += VW) {
            const vec_t x = *reinterpret_cast<const vec_t*>(p + j);
#pragma unroll
            for (int e = 0; e < VW; ++e) a[j + e] = x[e];
        }
    } else {
#pragma unroll
        for (int j = 0; j < NB; ++j) a[j] = (valid && j < w) ? p[j] : T(0);
    }
}

template <typename T>
__device__ __forceinline__ void store_row_direct(T* __restrict__ R, int64_t ld, unsigned pos, int c0, int w,
                                                 const T (&a)[NB])
{
    if (pos == POS_NONE) return;
    constexpr int VW = 16 / (int)sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(VW)));
    T* p = R + (int64_t)pos * ld + c0;
    const bool vec_ok = ((reinterpret_cast<uintptr_t>(p) & 15) == 0);
    if (vec_ok && w == NB) {
#pragma unroll
        for (int j = 0; j < NB; j += VW) {
            vec_t x;
#pragma unroll
            for (int e = 0; e < VW; ++e) x[e] = a[j + e];
            *reinterpret_cast<vec_t*>(p + j) = x;
        }
    } else {
#pragma unroll
        for (int j = 0; j < NB; ++j)
            if (j < w) p[j] = a[j];
    }
}

// ---- the same bookkeeping, incrementally and in registers -----------------------------------------------------------
// perm_build_wave costs ~64 dependent LDS round trips (15-20 us) when run after the last pivot step.  The cooperative kernels
// instead let ONE wave (the last wave of workgroup 0) carry the slot contents across the pivot steps in registers: lane s
// holds content[s] and content[NB+s], lane x holds the row of extra slot x; a step is a few readlane / select operations.
struct PermState {
    int c_lo;    // content[lane]
    int c_hi;    // content[NB + lane]
    int r_hi;    // rows[NB + lane] (row index of extra slot `lane`, -1 = unused)
    int nextra;  // number of extra slots in use (wave-uniform)
};

__device__ __forceinline__ PermState perm_state_init(int lane)
{
    PermState ps;
    ps.c_lo = lane;
    ps.c_hi = NB + lane;
    ps.r_hi = -1;
    ps.nextra = 0;
    return ps;
}

// interchange  position base+i <-> row p  (p wave-uniform)
__device__ __forceinline__ void perm_state_step(PermState& ps, int base, int i, int p, int lane)
{
    int sp;
    if (p < base + NB) {
        sp = p - base;
    } else {
        const u64 mask = __ballot(lane < ps.nextra && ps.r_hi == p);
        if (mask) {
            sp = NB + (__ffsll((long long)mask) - 1);
        } else {
            sp = NB + ps.nextra;
            if (lane == ps.nextra) ps.r_hi = p;
            ++ps.nextra;
        }
    }
    sp = __builtin_amdgcn_readfirstlane(sp);
    if (sp != i) {
        const int ci = __builtin_amdgcn_readlane(ps.c_lo, i);
        const int cs = (sp < NB) ? __builtin_amdgcn_readlane(ps.c_lo, sp & (NB - 1))
                                 : __builtin_amdgcn_readlane(ps.c_hi, sp & (NB - 1));
        if (lane == i) ps.c_lo = cs;
        if (sp < NB) { if (lane == sp) ps.c_lo = ci; }
        else { if (lane == sp - NB) ps.c_hi = ci; }
    }
}

// compact the slots whose content changed into the move list of the chunk (rows_tmp: NB ints of LDS)
__device__ __forceinline__ void perm_state_finish(const PermState& ps, int base, int lane, int* rows_tmp, int* out_cnt,
                                                  int* out_dst, int* out_src)
{
    rows_tmp[lane] = ps.r_hi;
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    int total = 0;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int s = half * NB + lane;
        const int c = half ? ps.c_hi : ps.c_lo;
        const bool moved = (half == 0 || lane < ps.nextra) && (c != s);
        const u64 mask = __ballot(moved);
        if (moved) {
            const int off = total + __popcll(mask & ((1ull << lane) - 1ull));
            out_dst[off] = half ? ps.r_hi : base + lane;
            out_src[off] = (c < NB) ? base + c : rows_tmp[c - NB];
        }
        total += __popcll(mask);
    }
    if (lane == 0) *out_cnt = total;
}

// ---- wave-wide argmax of (key, pos): DPP butterflies inside each row of 16 lanes, then 4 row results via readlane ----
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ double dpp_val(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = dpp_u32<CTRL>((unsigned)b), hi = dpp_u32<CTRL>((unsigned)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <int CTRL>
__device__ __forceinline__ float dpp_val(float v)
{
    return __uint_as_float(dpp_u32<CTRL>(__float_as_uint(v)));
}
__device__ __forceinline__ double readlane_val(double v, int l)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ float readlane_val(float v, int l)
{
    return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), l));
}

__device__ __forceinline__ double tmax(double a, double b) { return __builtin_fmax(a, b); }
__device__ __forceinline__ float tmax(float a, float b) { return __builtin_fmaxf(a, b); }

// max of a non-negative, NaN-free key over the wave; every lane gets the result
template <typename T>
__device__ __forceinline__ T wave_max(T v)
{
    v = tmax(v, dpp_val<0xB1>(v));   // quad_perm [1,0,3,2]
    v = tmax(v, dpp_val<0x4E>(v));   // quad_perm [2,3,0,1]
    v = tmax(v, dpp_val<0x141>(v));  // row_half_mirror
    v = tmax(v, dpp_val<0x140>(v));  // row_mirror  -> every lane of a 16-lane row holds the row max
    const T r0 = readlane_val(v, 0), r1 = readlane_val(v, 16), r2 = readlane_val(v, 32), r3 = readlane_val(v, 48);
    return tmax(tmax(r0, r1), tmax(r2, r3));
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
    v = min(v, dpp_u32<0xB1>(v));
    v = min(v, dpp_u32<0x4E>(v));
    v = min(v, dpp_u32<0x141>(v));
    v = min(v, dpp_u32<0x140>(v));
    const unsigned r0 = (unsigned)__builtin_amdgcn_readlane((int)v, 0), r1 = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
    const unsigned r2 = (unsigned)__builtin_amdgcn_readlane((int)v, 32), r3 = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    return min(min(r0, r1), min(r2, r3));
}

// Wave-wide argmax of (key desc, pos asc): two cheap passes (v_max on the key, then v_min on the positions of the lanes
// that hold the maximum) instead of a compare-select butterfly on (key,pos) pairs.  Keys are >= 0 and never NaN.
template <typename T>
__device__ __forceinline__ void wave_argmax(T& v, unsigned& p)
{
    const T m = wave_max<T>(v);
    const bool hit = (v == m) && (p != POS_NONE);
    const u64 mask = __ballot(hit);
    if (__popcll(mask) == 1) {  // the usual case: one lane holds the maximum -> its position by one readlane
        p = (unsigned)__builtin_amdgcn_readlane((int)p, __ffsll((long long)mask) - 1);
    } else {                     // exact ties (or no candidate at all): lowest position among the lanes holding the max
        p = wave_min_u32(hit ? p : POS_NONE);
    }
    v = m;
}

template <typename T>
struct MidOut {
    T scale;         // 1/pivot (or 1 when the pivot is exactly zero)
    unsigned pos;    // this thread's row position after the interchange
    unsigned flags;  // bit0: apply the update to this row, bit1: row still active, bit2: give up (timeout)
};                   // 16 bytes: returned in registers (a larger struct goes through scratch memory on every step)

// Workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() also drains the vector-memory counter,
// i.e. it would make wave 0 sit out the round trip of the row request it has just issued -- the very latency the pipelined
// kernel wants to hide behind the search (scripts/panel_skew_trace.py: 400 ns at barrier 1 with __syncthreads()).
__device__ __forceinline__ void barrier_lds_only()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// panel_local.hip: the leaf on the blocks b % stride == sel of a G*stride grid (local: plain-store records, one XCD)
template <typename T>
int launch_panel_local(Handle* h, const PanelArgs<T>& p, int stride, int sel, int want_xcc, int local);

// panel_single.hip: a leaf of at most 512 rows in one workgroup (no cross-workgroup records at all)
template <typename T>
int launch_panel_single(Handle* h, const PanelArgs<T>& p);

// panel_blocked.hip: a full leaf as 8 sub-panels of 8 columns, one chain wave per workgroup (any number of workgroups up to 64)
constexpr int PANEL_BLOCKED_ROWS_F64 = 7 * 64, PANEL_BLOCKED_ROWS_F32 = 8 * 64;   // rows per workgroup
template <typename T>
int launch_panel_blocked(Handle* h, const PanelArgs<T>& p, int local);
int panel_blocked_resident_limit_f64(int num_cus);
int panel_blocked_resident_limit_f32(int num_cus);

int panel_resident_limit_f64(int num_cus);
int panel_resident_limit_f32(int num_cus);
int panel_local_resident_limit_f64(int num_cus);
int panel_local_resident_limit_f32(int num_cus);

}  // namespace rflu
