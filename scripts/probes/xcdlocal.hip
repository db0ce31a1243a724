// XCD-local exchange probes (round 2).
//   1. ping-pong of a 16-byte tagged granule between two workgroups on the SAME XCD for every store/load cache policy:
//      does a plain (L2-resident) store + L1-bypassing load give an L2-latency hop?
//   2. the panel's per-column skeleton: G workgroups, each publishes one header per step and polls all G headers
//      (all-to-all), with a wave argmax + LDS barrier in between -- per-step time same-XCD vs spread over the XCDs.
// hipcc --offload-arch=gfx950 -O3 xcdlocal.hip -o xcdlocal && ./xcdlocal
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned u4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned xcc_id()
{
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 0xf;
}

template <int AUX>
__device__ __forceinline__ void st16(__amdgpu_buffer_rsrc_t r, unsigned off, u4v x)
{
    __builtin_amdgcn_raw_buffer_store_b128(x, r, off, 0, AUX);
}
template <int AUX>
__device__ __forceinline__ u4v ld16(__amdgpu_buffer_rsrc_t r, unsigned off)
{
    return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, AUX);
}

// ---- 1. ping-pong -----------------------------------------------------------------------------------------------------
template <int SA, int LA>
__global__ void pingpong(unsigned* buf, long long* out, int p, int rounds)
{
    const int me = blockIdx.x;
    if (me != 0 && me != p) return;
    if (threadIdx.x != 0) return;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 1 << 20, 0x00020000);
    const unsigned mine = me == 0 ? 0u : 4096u, theirs = me == 0 ? 4096u : 0u;
    out[me == 0 ? 2 : 3] = xcc_id();
    long long bad = 0;
    const long long w0 = wall_clock64();
    for (int i = 1; i <= rounds; ++i) {
        const u4v x = {(unsigned)i, (unsigned)i, 0u, (unsigned)i};
        if (me == 0) st16<SA>(r, mine, x);
        int sp = 0;
        for (; sp < 100000; ++sp) {
            asm volatile("" ::: "memory");
            const u4v y = ld16<LA>(r, theirs);
            if (y[1] == (unsigned)i && y[3] == (unsigned)i) break;
        }
        if (sp >= 100000) { ++bad; if (bad > 3) break; }
        if (me != 0) st16<SA>(r, mine, x);
    }
    if (me == 0) { out[0] = wall_clock64() - w0; out[1] = bad; }
}

// ---- 2. all-to-all header step ----------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v)
{
    v = max(v, dpp_u32<0xB1>(v));
    v = max(v, dpp_u32<0x4E>(v));
    v = max(v, dpp_u32<0x141>(v));
    v = max(v, dpp_u32<0x140>(v));
    const unsigned r0 = (unsigned)__builtin_amdgcn_readlane((int)v, 0), r1 = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
    const unsigned r2 = (unsigned)__builtin_amdgcn_readlane((int)v, 32), r3 = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    return max(max(r0, r1), max(r2, r3));
}

// grid = G * stride workgroups; participants are blocks b with b % stride == 0 (stride 8 => all on XCD 0)
template <int SA, int LA, int THREADS, int UNROLL = 1, int NOPS = 0>
__global__ void __launch_bounds__(THREADS) allstep(unsigned* buf, long long* out, int G, int stride, int steps, int local_work)
{
    if (blockIdx.x % stride != 0) return;
    const int g = blockIdx.x / stride;
    constexpr int WAVES = THREADS / 64;
    __shared__ unsigned s_w[16];
    __shared__ int s_bad;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 1 << 20, 0x00020000);
    if (tid == 0) { out[8 + g] = xcc_id(); s_bad = 0; }
    __syncthreads();
    unsigned acc = (unsigned)(g * 977 + tid);
    const long long w0 = wall_clock64();
    int bad = 0;
#pragma unroll UNROLL
    for (int s = 1; s <= steps; ++s) {
        if (NOPS == 300) asm volatile(".rept 300\n s_nop 0\n .endr" ::: "memory");   // code volume only (1.2 KB, ~300 clocks)
        // local search: wave argmax, LDS, barrier, combine
        unsigned key = acc * 2654435761u + (unsigned)s;
        if (local_work) {
            key = wave_max_u32(key);
            if (lane == 0) s_w[wave] = key;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            unsigned m = 0;
#pragma unroll
            for (int x = 0; x < WAVES; ++x) m = max(m, s_w[x]);
            key = m;
        }
        const unsigned base = (unsigned)(s & 1) * 32768u;
        // the "owner" thread of this step publishes the header (3 x 16 B as in the pipelined panel)
        if (tid == ((s * 37) & (THREADS - 1))) {
            const u4v x = {key, (unsigned)s, acc, (unsigned)s};
            st16<SA>(r, base + (unsigned)g * 64u, x);
            st16<SA>(r, base + (unsigned)g * 64u + 16u, x);
            st16<SA>(r, base + (unsigned)g * 64u + 32u, x);
        }
        // every wave polls all G headers, lane x = header x
        unsigned best = 0;
        for (int x = lane; x < G; x += 64) {
            int sp = 0;
            for (;;) {
                asm volatile("" ::: "memory");
                const u4v a = ld16<LA>(r, base + (unsigned)x * 64u);
                const u4v b = ld16<LA>(r, base + (unsigned)x * 64u + 16u);
                const u4v c = ld16<LA>(r, base + (unsigned)x * 64u + 32u);
                if (a[1] == (unsigned)s && a[3] == (unsigned)s && b[1] == (unsigned)s && b[3] == (unsigned)s && c[1] == (unsigned)s) {
                    best = max(best, a[0]);
                    break;
                }
                if (++sp > 200000) { bad = 1; break; }
            }
        }
        best = wave_max_u32(best);
        acc += best;
        if (bad) break;
    }
    if (bad) s_bad = 1;
    __syncthreads();
    if (tid == 0) {
        if (g == 0) { out[0] = wall_clock64() - w0; out[2] = acc; }
        if (s_bad) out[1] = 1;
    }
}

template <int SA, int LA>
static void run_pp(unsigned* buf, long long* out, const char* name, int p)
{
    const int rounds = 2000;
    hipMemset(buf, 0, 1 << 20);
    hipMemset(out, 0, 4096);
    pingpong<SA, LA><<<32, 64>>>(buf, out, p, rounds);
    hipError_t e = hipDeviceSynchronize();
    long long h[4];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("pingpong %-22s partner wg %2d (xcc %lld/%lld): round trip %7.1f ns  bad=%lld %s\n", name, p, h[2], h[3],
           (double)h[0] * 10.0 / rounds, h[1], e == hipSuccess ? "" : "ERROR");
    fflush(stdout);
}

template <int SA, int LA, int THREADS, int UNROLL = 1, int NOPS = 0>
static void run_all(unsigned* buf, long long* out, const char* name, int G, int stride, int local_work)
{
    const int steps = 4032;
    hipMemset(buf, 0, 1 << 20);
    hipMemset(out, 0, 4096);
    allstep<SA, LA, THREADS, UNROLL, NOPS><<<G * stride, THREADS>>>(buf, out, G, stride, steps, local_work);
    hipError_t e = hipDeviceSynchronize();
    std::vector<long long> h(8 + 64);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    int nx[16] = {0};
    for (int g = 0; g < G && g < 64; ++g) nx[h[8 + g] & 15]++;
    printf("allstep  %-22s T=%4d U=%2d nops=%3d G=%2d stride=%d local=%d: %7.1f ns/step  bad=%lld  xcc histogram:", name, THREADS, UNROLL, NOPS, G, stride,
           local_work, (double)h[0] * 10.0 / steps, h[1]);
    for (int x = 0; x < 8; ++x) printf(" %d", nx[x]);
    printf(" %s\n", e == hipSuccess ? "" : "ERROR");
    fflush(stdout);
}

int main(int argc, char** argv)
{
    unsigned* buf;
    long long* out;
    hipMalloc(&buf, 1 << 20);
    hipMalloc(&out, 4096);
    if (argc > 1) {   // instruction-fetch experiment: the same step looped (hot I-cache) vs 64 distinct copies (> 64 KB, cold)
        for (int rep = 0; rep < 2; ++rep) {
            run_all<0, 16, 512, 1, 0>(buf, out, "looped", 32, 8, 1);
            run_all<0, 16, 512, 1, 300>(buf, out, "looped + 300 nops", 32, 8, 1);
            run_all<0, 16, 512, 64, 0>(buf, out, "64 copies", 32, 8, 1);
            run_all<0, 16, 512, 64, 300>(buf, out, "64 copies + 300 nops", 32, 8, 1);
            run_all<0, 16, 512, 1, 300>(buf, out, "looped + 300 nops", 2, 8, 1);
            run_all<0, 16, 512, 64, 300>(buf, out, "64 copies + 300 nops", 2, 8, 1);
        }
        return 0;
    }
    for (int p : {8, 1}) {
        run_pp<16, 16>(buf, out, "st sc1 / ld sc1", p);
        run_pp<0, 16>(buf, out, "st plain / ld sc1", p);
        run_pp<1, 16>(buf, out, "st sc0 / ld sc1", p);
        run_pp<0, 17>(buf, out, "st plain / ld sc0sc1", p);
        run_pp<17, 17>(buf, out, "st sc0sc1 / ld sc0sc1", p);
        run_pp<0, 1>(buf, out, "st plain / ld sc0", p);
        run_pp<0, 2>(buf, out, "st plain / ld nt", p);
    }
    for (int local = 0; local < 2; ++local) {
        for (int G : {8, 16, 32}) {
            run_all<16, 16, 512>(buf, out, "st sc1 / ld sc1", G, 1, local);
            run_all<16, 16, 512>(buf, out, "st sc1 / ld sc1", G, 8, local);
            run_all<0, 16, 512>(buf, out, "st plain / ld sc1", G, 8, local);
            run_all<0, 2, 512>(buf, out, "st plain / ld nt", G, 8, local);
        }
    }
    run_all<0, 16, 1024>(buf, out, "st plain / ld sc1", 16, 8, 1);
    run_all<0, 16, 256>(buf, out, "st plain / ld sc1", 32, 8, 1);
    run_all<0, 16, 64>(buf, out, "st plain / ld sc1", 32, 8, 0);
    return 0;
}
