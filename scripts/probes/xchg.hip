// All-to-all header exchange probe (round 4): what does ONE polling wave per workgroup pay per step, by poll strategy?
//   G workgroups (one per CU, spread over the XCDs), per step: `work` clocks of local work, one lane publishes a 64-byte
//   tagged header (4 x 16 B, write-through), lanes 0..G-1 poll the G headers until all carry the step's tag.
//   MODE 0: the panel's loop -- issue the 4 loads, wait for them, check, repeat (a miss costs a full round trip)
//   MODE 1: K poll sets kept in flight (issued back to back, checked in order, re-issued after each check): the detection
//           granularity becomes round trip / K
//   HB = header bytes (64 or 32 or 16)
// hipcc --offload-arch=gfx950 -O3 xchg.hip -o xchg && ./xchg
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
__device__ __forceinline__ void st16(__amdgpu_buffer_rsrc_t r, unsigned off, u4v x) { __builtin_amdgcn_raw_buffer_store_b128(x, r, off, 0, AUX); }
template <int AUX>
__device__ __forceinline__ u4v ld16(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, AUX); }

template <int NG>
struct Set { u4v g[NG]; };

template <int NG, int LA>
__device__ __forceinline__ void issue(Set<NG>& s, __amdgpu_buffer_rsrc_t r, unsigned off)
{
    asm volatile("" ::: "memory");
#pragma unroll
    for (int k = 0; k < NG; ++k) s.g[k] = ld16<LA>(r, off + 16u * k);
}
template <int NG>
__device__ __forceinline__ bool good(const Set<NG>& s, unsigned tag)
{
    unsigned bad = 0;
#pragma unroll
    for (int k = 0; k < NG; ++k) bad |= (s.g[k][1] ^ tag) | (s.g[k][3] ^ tag);
    return bad == 0;
}

__device__ __forceinline__ void spin_clocks(int n)
{
    if (n <= 0) return;
    const long long t0 = clock64();
    while (clock64() - t0 < n) {}
}

// one wave per workgroup
template <int MODE, int K, int NG, int SA, int LA>
__global__ void __launch_bounds__(64) xchg(unsigned* buf, long long* out, int G, int stride, int steps, int work, int sleepn)
{
    if (blockIdx.x % stride != 0) return;
    const int g = blockIdx.x / stride;
    const int lane = threadIdx.x;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 1 << 20, 0x00020000);
    if (lane == 0) out[8 + g] = xcc_id();
    unsigned acc = (unsigned)(g * 977 + lane);
    long long polls = 0;
    int bad = 0;
    const long long w0 = wall_clock64();
    for (int s = 1; s <= steps && !bad; ++s) {
        spin_clocks(work);
        const unsigned base = (unsigned)(s & 1) * 32768u;
        if (lane == 0) {
            const u4v x = {acc, (unsigned)s, acc ^ 0x5555u, (unsigned)s};
#pragma unroll
            for (int k = 0; k < NG; ++k) st16<SA>(r, base + (unsigned)g * 64u + 16u * k, x);
        }
        const bool mine = lane < G;
        const unsigned off = base + (unsigned)(mine ? lane : 0) * 64u;
        bool done = !mine;
        unsigned val = 0;
        if (MODE == 2) {
            // K = delay before the first poll in units of 100 clocks, sleepn = pause between polls in units of 64 clocks
            spin_clocks(K * 100);
            int sp = 0;
            while (!done) {
                Set<NG> q;
                issue<NG, LA>(q, r, off);
                if (good<NG>(q, (unsigned)s)) { done = true; val = q.g[0][0]; }
                else if (sleepn) spin_clocks(sleepn * 64);
                if (lane == G - 1) ++polls;
                if (++sp > 200000) { bad = 1; break; }
            }
        } else if (MODE == 0) {
            int sp = 0;
            while (!done) {
                Set<NG> q;
                issue<NG, LA>(q, r, off);
                ++polls;
                if (good<NG>(q, (unsigned)s)) { done = true; val = q.g[0][0]; }
                else if (sleepn) __builtin_amdgcn_s_sleep(1);
                if (++sp > 200000) { bad = 1; break; }
            }
        } else {
            constexpr int KK = (MODE == 1 && K > 0) ? K : 1;
            Set<NG> q[KK];
#pragma unroll
            for (int k = 0; k < KK; ++k) {
                issue<NG, LA>(q[k], r, off);
                if (sleepn) spin_clocks(sleepn * 64);
            }
            int sp = 0;
            for (;;) {
                bool out_ = false;
#pragma unroll
                for (int k = 0; k < KK; ++k) {
                    if (!done && good<NG>(q[k], (unsigned)s)) { done = true; val = q[k].g[0][0]; }
                    ++polls;
                    if (!__any(!done)) { out_ = true; break; }
                    issue<NG, LA>(q[k], r, off);
                }
                if (out_) break;
                if (++sp > 200000) { bad = 1; break; }
            }
        }
        // winner = max over lanes (stands for the argmax)
        unsigned m = val;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
        acc = acc * 2654435761u + m;
    }
    if (lane == 0) {
        if (g == 0) { out[0] = wall_clock64() - w0; out[2] = acc; out[3] = polls; }
        if (bad) out[1] = 1;
    }
}

template <int MODE, int K, int NG, int SA, int LA>
static void run(unsigned* buf, long long* out, const char* name, int G, int stride, int work, int sleepn)
{
    const int steps = 4000;
    hipMemset(buf, 0, 1 << 20);
    hipMemset(out, 0, 4096);
    xchg<MODE, K, NG, SA, LA><<<G * stride, 64>>>(buf, out, G, stride, steps, work, sleepn);
    hipError_t e = hipDeviceSynchronize();
    std::vector<long long> h(8 + 64);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    int nx[16] = {0};
    for (int g = 0; g < G && g < 64; ++g) nx[h[8 + g] & 15]++;
    printf("%-26s K=%d hdr=%2dB G=%2d stride=%d work=%4d sleep=%d: %7.1f ns/step (minus work %7.1f)  polls/step %5.2f  bad=%lld  xcc:", name, K, NG * 16, G, stride, work, sleepn,
           (double)h[0] * 10.0 / steps, (double)h[0] * 10.0 / steps - work / 2.4, (double)h[3] / steps, h[1]);
    for (int x = 0; x < 8; ++x) printf(" %d", nx[x]);
    printf(" %s\n", e == hipSuccess ? "" : "ERROR");
    fflush(stdout);
}

int main(int argc, char** argv)
{
    unsigned* buf;
    long long* out;
    int uncached = argc > 1 && argv[1][0] == 'u';
    if (uncached) {
        if (hipExtMallocWithFlags((void**)&buf, 1 << 20, hipDeviceMallocUncached) != hipSuccess) { printf("uncached alloc failed\n"); return 1; }
        printf("== exchange buffer: hipDeviceMallocUncached\n");
    } else {
        hipMalloc(&buf, 1 << 20);
        printf("== exchange buffer: hipMalloc\n");
    }
    hipMalloc(&out, 4096);
    const bool only_delay = argc > 1 && argv[1][0] == 'd';
    for (int work : {0, 2400}) {
        if (only_delay) break;
        for (int G : {1, 2, 8, 16, 32}) {
            const int stride = 1;
            run<0, 1, 4, 16, 16>(buf, out, "sequential sc1/sc1", G, stride, work, 0);
            run<1, 2, 4, 16, 16>(buf, out, "pipelined sc1/sc1", G, stride, work, 0);
            run<1, 3, 4, 16, 16>(buf, out, "pipelined sc1/sc1", G, stride, work, 0);
            run<1, 4, 4, 16, 16>(buf, out, "pipelined sc1/sc1", G, stride, work, 0);
            run<1, 3, 4, 16, 16>(buf, out, "pipelined sc1/sc1", G, stride, work, 2);
            run<1, 3, 2, 16, 16>(buf, out, "pipelined sc1/sc1", G, stride, work, 0);
            run<1, 4, 1, 16, 16>(buf, out, "pipelined sc1/sc1", G, stride, work, 0);
            run<0, 1, 1, 16, 16>(buf, out, "sequential sc1/sc1", G, stride, work, 0);
            run<1, 3, 4, 17, 17>(buf, out, "pipelined sc0sc1/sc0sc1", G, stride, work, 0);
            if (uncached) {
                run<0, 1, 4, 0, 0>(buf, out, "sequential plain/plain", G, stride, work, 0);
                run<1, 3, 4, 0, 0>(buf, out, "pipelined plain/plain", G, stride, work, 0);
            }
        }
    }
    // delay sweep: does a pause between the publish and the first poll (and between polls) pay?
    for (int G : {8, 32}) {
        run<2, 0, 4, 16, 16>(buf, out, "delayed sc1/sc1", G, 1, 2400, 0);
        run<2, 2, 4, 16, 16>(buf, out, "delayed sc1/sc1", G, 1, 2400, 0);
        run<2, 4, 4, 16, 16>(buf, out, "delayed sc1/sc1", G, 1, 2400, 0);
        run<2, 6, 4, 16, 16>(buf, out, "delayed sc1/sc1", G, 1, 2400, 0);
        run<2, 8, 4, 16, 16>(buf, out, "delayed sc1/sc1", G, 1, 2400, 0);
        run<2, 10, 4, 16, 16>(buf, out, "delayed sc1/sc1", G, 1, 2400, 0);
        run<2, 14, 4, 16, 16>(buf, out, "delayed sc1/sc1", G, 1, 2400, 0);
        run<2, 18, 4, 16, 16>(buf, out, "delayed sc1/sc1", G, 1, 2400, 0);
        run<2, 0, 4, 16, 16>(buf, out, "delayed sc1/sc1", G, 1, 2400, 2);
        run<2, 0, 4, 16, 16>(buf, out, "delayed sc1/sc1", G, 1, 2400, 4);
        run<2, 6, 4, 16, 16>(buf, out, "delayed sc1/sc1", G, 1, 2400, 2);
        run<2, 10, 4, 16, 16>(buf, out, "delayed sc1/sc1", G, 1, 2400, 2);
        run<2, 6, 3, 16, 16>(buf, out, "delayed sc1/sc1", G, 1, 2400, 0);
        run<2, 10, 3, 16, 16>(buf, out, "delayed sc1/sc1", G, 1, 2400, 0);
        run<2, 6, 2, 16, 16>(buf, out, "delayed sc1/sc1", G, 1, 2400, 0);
        run<2, 10, 2, 16, 16>(buf, out, "delayed sc1/sc1", G, 1, 2400, 0);
    }
    for (int G : {8, 16}) {
        run<2, 0, 4, 0, 16>(buf, out, "delayed plain/sc1 (XCD)", G, 8, 2400, 0);
        run<2, 3, 4, 0, 16>(buf, out, "delayed plain/sc1 (XCD)", G, 8, 2400, 0);
        run<2, 6, 4, 0, 16>(buf, out, "delayed plain/sc1 (XCD)", G, 8, 2400, 0);
    }
    if (argc > 1 && argv[1][0] == 'd') return 0;
    // same XCD, plain stores (the XCD-local leaf)
    for (int G : {8, 16, 32}) {
        run<0, 1, 4, 0, 16>(buf, out, "sequential plain/sc1", G, 8, 2400, 0);
        run<1, 3, 4, 0, 16>(buf, out, "pipelined plain/sc1", G, 8, 2400, 0);
    }
    return 0;
}
