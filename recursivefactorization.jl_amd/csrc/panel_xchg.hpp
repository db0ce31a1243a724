// panel_xchg.hpp -- pieces shared by the cooperative leaves that exchange candidates through tagged records (panel_local.hip,
// panel_blocked.hip): XCC id, wave-uniform copies, the pivot search on integer keys with single-instruction DPP stages, the
// 64-byte candidate header.
#pragma once
#include "panel_common.hpp"

namespace rflu {

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
// (hand-written: hipcc emits v_mov / s_nop / v_mov_dpp / v_max per stage where ONE v_max_u32_dpp does the stage; the s_nop 1
// covers the two wait states between a VALU write and a DPP read of the same register)
__device__ __forceinline__ unsigned wave_max_b(unsigned v)
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
__device__ __forceinline__ unsigned wave_min_b(unsigned v)
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

// the same over lanes 0..7 only (the <= 8 wave records of a workgroup): three stages, result in lanes 0..7
__device__ __forceinline__ unsigned wave_max8_b(unsigned v)
{
    asm("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v));
    asm("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(v));
    asm("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(v));
    asm("s_nop 1" : "+v"(v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 0);
}
__device__ __forceinline__ unsigned wave_min8_b(unsigned v)
{
    asm("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v));
    asm("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(v));
    asm("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(v));
    asm("s_nop 1" : "+v"(v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 0);
}

// Every lane passes (hi, lo, pos); pos == POS_NONE marks a lane without a candidate (its hi / lo must be 0).
// Returns wave-uniform: the best key in (hi, lo), its position in pos (POS_NONE: no candidate at all) and the lane that
// holds it (0 if none).
// The common case needs ONE reduction: the high words (sign-free exponent + 20 mantissa bits for Float64) of two rows of a
// wave almost never coincide at the maximum; only then the low words and, for exact ties, the positions are reduced too.
// EIGHT: only lanes 0..7 carry entries (the others pass hi = lo = 0, pos = POS_NONE): three DPP stages instead of six.
template <bool TWO, bool EIGHT = false>
__device__ __forceinline__ int wave_argmax_i(unsigned& hi, unsigned& lo, unsigned& pos)
{
    const unsigned mh = EIGHT ? wave_max8_b(hi) : wave_max_b(hi);
    bool hit = (hi == mh) && (pos != POS_NONE);
    u64 mask = __ballot(hit);
    unsigned p = POS_NONE, ml = 0u;
    int wl = 0;
    if (mask != 0) {
        if (__popcll(mask) != 1) {
            if (TWO) {
                ml = EIGHT ? wave_max8_b(hit ? lo : 0u) : wave_max_b(hit ? lo : 0u);
                hit = hit && (lo == ml);
                mask = __ballot(hit);
            }
            if (__popcll(mask) != 1) {   // exact ties: the lowest position among the lanes holding the maximum
                p = EIGHT ? wave_min8_b(hit ? pos : POS_NONE) : wave_min_b(hit ? pos : POS_NONE);
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
struct Hdr4;
template <>
struct Hdr4<double> {
    template <int AUX>
    static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned tag, unsigned pos, double a,
                                                 double a1, double l)
    {
        const u64 A = (u64)__double_as_longlong(a), B = (u64)__double_as_longlong(a1), L = (u64)__double_as_longlong(l);
        const u4v g0 = {pos, tag, (unsigned)(A >> 32), tag};
        const u4v g1 = {(unsigned)A, tag, (unsigned)(B >> 32), tag};
        const u4v g2 = {(unsigned)B, tag, (unsigned)(L >> 32), tag};
        const u4v g3 = {(unsigned)L, tag, 0u, tag};
        __builtin_amdgcn_raw_buffer_store_b128(g0, r, off, 0, AUX);
        __builtin_amdgcn_raw_buffer_store_b128(g1, r, off + 16, 0, AUX);
        __builtin_amdgcn_raw_buffer_store_b128(g2, r, off + 32, 0, AUX);
        __builtin_amdgcn_raw_buffer_store_b128(g3, r, off + 48, 0, AUX);
    }
    static __device__ __forceinline__ bool load(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned tag, unsigned& pos, double& a,
                                                double& a1, double& l)
    {
        const u4v g0 = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, AUX_SC1);
        const u4v g1 = __builtin_amdgcn_raw_buffer_load_b128(r, off + 16, 0, AUX_SC1);
        const u4v g2 = __builtin_amdgcn_raw_buffer_load_b128(r, off + 32, 0, AUX_SC1);
        const u4v g3 = __builtin_amdgcn_raw_buffer_load_b128(r, off + 48, 0, AUX_SC1);
        pos = g0[0];
        a = __longlong_as_double((long long)(((u64)g0[2] << 32) | (u64)g1[0]));
        a1 = __longlong_as_double((long long)(((u64)g1[2] << 32) | (u64)g2[0]));
        l = __longlong_as_double((long long)(((u64)g2[2] << 32) | (u64)g3[0]));
        const unsigned ok = (g0[1] ^ tag) | (g0[3] ^ tag) | (g1[1] ^ tag) | (g1[3] ^ tag) | (g2[1] ^ tag) | (g2[3] ^ tag) |
                            (g3[1] ^ tag) | (g3[3] ^ tag);
        return ok == 0u;
    }
};
template <>
struct Hdr4<float> {
    template <int AUX>
    static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned tag, unsigned pos, float a,
                                                 float a1, float l)
    {
        const u4v g0 = {pos, tag, __float_as_uint(a), tag};
        const u4v g1 = {__float_as_uint(a1), tag, __float_as_uint(l), tag};
        __builtin_amdgcn_raw_buffer_store_b128(g0, r, off, 0, AUX);
        __builtin_amdgcn_raw_buffer_store_b128(g1, r, off + 16, 0, AUX);
    }
    static __device__ __forceinline__ bool load(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned tag, unsigned& pos, float& a,
                                                float& a1, float& l)
    {
        const u4v g0 = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, AUX_SC1);
        const u4v g1 = __builtin_amdgcn_raw_buffer_load_b128(r, off + 16, 0, AUX_SC1);
        pos = g0[0];
        a = __uint_as_float(g0[2]);
        a1 = __uint_as_float(g1[0]);
        l = __uint_as_float(g1[2]);
        return ((g0[1] ^ tag) | (g0[3] ^ tag) | (g1[1] ^ tag) | (g1[3] ^ tag)) == 0u;
    }
};

}  // namespace rflu
