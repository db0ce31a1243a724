// laswp_strip.hpp -- one wave applies the row moves of a run of pivot chunks to its strip of columns: the body shared by
// laswp_kernel (laswp.hip) and the persistent update engine (engine.hip).  Geometry and the reasons for it: laswp.hip.
#pragma once

#include <stdint.h>

#include "rflu_internal.hpp"

namespace rflu {

template <typename T, int VW, int LW_LPR>
__device__ __forceinline__ void laswp_strip(T* __restrict__ R, int64_t ld, int64_t c0, int64_t ncolsA, int64_t c1,
                                            int64_t ncolsB, int64_t c2, int64_t ncolsC, const int* __restrict__ pm_cnt,
                                            const int* __restrict__ pm_dst, const int* __restrict__ pm_src, int chunk0,
                                            int chunk1, int64_t strip)
{
    constexpr int LW_RS = 64 / LW_LPR;        // row slots per wave
    constexpr int LW_NI = (2 * NB) / LW_RS;   // loads per lane and chunk (at most)
    constexpr int SC = LW_LPR * VW;           // columns per strip
    typedef T vec_t __attribute__((ext_vector_type(VW)));
    const int lane = threadIdx.x & 63;
    const int64_t stripsA = (ncolsA + SC - 1) / SC, stripsB = (ncolsB + SC - 1) / SC;
    const int cl = lane & (LW_LPR - 1), rsub = lane / LW_LPR;
    int64_t col, off, lim;
    int first = chunk0;
    if (strip < stripsA) {
        off = strip * SC + cl * VW; col = c0 + off; lim = ncolsA;
    } else if (strip < stripsA + stripsB) {
        off = (strip - stripsA) * SC + cl * VW; col = c1 + off; lim = ncolsB;
    } else {
        off = (strip - stripsA - stripsB) * SC + cl * VW; col = c2 + off; lim = ncolsC;
        first = chunk0 + 1;
    }
    const bool active = off < lim;   // whole vectors only: the launcher picks VW = 1 unless every range is a multiple of VW
    if (first >= chunk1) return;
    // (atomic = a vector load: a resident kernel -- engine.hip -- reads lists written while it runs, and the scalar cache a plain
    //  uniform load would go through is not refreshed by an agent-scope acquire)
    int cnt = __builtin_amdgcn_readfirstlane(__hip_atomic_load(pm_cnt + first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    int s0 = pm_src[(size_t)first * 2 * NB + lane], s1 = pm_src[(size_t)first * 2 * NB + NB + lane];
    int d0 = pm_dst[(size_t)first * 2 * NB + lane], d1 = pm_dst[(size_t)first * 2 * NB + NB + lane];
    for (int t = first; t < chunk1; ++t) {
        vec_t v[LW_NI];
        int dst[LW_NI];
#pragma unroll
        for (int i = 0; i < LW_NI; ++i) {
            const int e = i * LW_RS + rsub;   // < 128; the same for the 8 lanes of a row slot
            const int src = (i * LW_RS < NB) ? __shfl(s0, e & 63) : __shfl(s1, e & 63);
            dst[i] = (i * LW_RS < NB) ? __shfl(d0, e & 63) : __shfl(d1, e & 63);
            if (i * LW_RS < cnt) {   // wave-uniform
                if (e < cnt && active) v[i] = *reinterpret_cast<const vec_t*>(R + (int64_t)src * ld + col);
            }
        }
        const int cur = cnt;
        if (t + 1 < chunk1) {   // the next chunk's move list travels while this chunk's rows are still arriving
            cnt = __builtin_amdgcn_readfirstlane(__hip_atomic_load(pm_cnt + t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            s0 = pm_src[(size_t)(t + 1) * 2 * NB + lane];
            s1 = pm_src[(size_t)(t + 1) * 2 * NB + NB + lane];
            d0 = pm_dst[(size_t)(t + 1) * 2 * NB + lane];
            d1 = pm_dst[(size_t)(t + 1) * 2 * NB + NB + lane];
        }
#pragma unroll
        for (int i = 0; i < LW_NI; ++i) {
            const int e = i * LW_RS + rsub;
            if (i * LW_RS < cur) {
                if (e < cur && active) *reinterpret_cast<vec_t*>(R + (int64_t)dst[i] * ld + col) = v[i];
            }
        }
        // a row written in this chunk may be read in the next one by ANOTHER lane of the wave: have every store acknowledged
        // first (the memory pipeline keeps one wave's accesses in order anyway -- every parity test passes without this wait --
        // but the guarantee is per lane; the wait costs 2 % of the wide launches: 3.1 -> 3.03 TB/s)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

}  // namespace rflu
