// Can a resident kernel leave ONE XCD empty, and is a launch of 8*G workgroups then still dealt one in eight to that XCD?
//   resident: a stream whose CU mask leaves out the first CU of every XCD, 496 workgroups of 256 threads + 64 KB LDS (two per CU);
//             the ones that find themselves on XCC 0 exit, the others spin until the host raises a flag
//   leaf    : on the null stream, 8*G workgroups of 320 threads; the ones on XCC 0 count themselves and wait (bounded) until G have arrived,
//             the others exit.  Reported: how many ran on each XCC, how many arrived, how long the launch took.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__device__ __forceinline__ unsigned xcc_id() { unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); return x & 0xf; }
__device__ __forceinline__ unsigned hw_id() { unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(x)); return x; }
__global__ void __launch_bounds__(256) resident(volatile int* stop, int* per_xcc, int vacate, unsigned* where)
{
    extern __shared__ char lds[];
    const unsigned x = xcc_id();
    if (threadIdx.x == 0) where[blockIdx.x] = 0x80000000u | (x << 16) | (hw_id() & 0xffffu);
    if (threadIdx.x == 0) atomicAdd(&per_xcc[x], 1);
    asm volatile("v_mov_b32 v127, 0" ::: "v127");   // (128 VGPRs, like the engine: two workgroups leave half of each SIMD's file)
    if (vacate && x == 0) return;
    if (threadIdx.x == 0) { lds[0] = 1; while (*stop == 0) __builtin_amdgcn_s_sleep(64); }
    __syncthreads();
}
__global__ void __launch_bounds__(320) leaf(int G, int* arrived, int* per_xcc, long long* spins_out, unsigned* where)
{
    __shared__ char big[48 * 1024];
    asm volatile("v_mov_b32 v249, 0" ::: "v249");   // (250 VGPRs: fits only on a CU with an empty SIMD 0 half... i.e. the CU the mask leaves out)
    const unsigned x = xcc_id();
    if (threadIdx.x == 0) { big[blockIdx.x] = 1; where[blockIdx.x] = 0x80000000u | (x << 16) | (hw_id() & 0xffffu); }
    if (threadIdx.x == 0) atomicAdd(&per_xcc[x], 1);
    if (x != 0) return;
    if (threadIdx.x == 0) {
        atomicAdd(arrived, 1);
        long long spins = 0;
        while (__hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < G && spins < 2000000) { __builtin_amdgcn_s_sleep(8); ++spins; }
        if (blockIdx.x < 8) *spins_out = spins;
    }
    __syncthreads();
}
int main(int argc, char** argv)
{
    const int vacate = argc > 1 ? atoi(argv[1]) : 1;
    const int wgs = argc > 2 ? atoi(argv[2]) : 496;
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipStream_t P; hipStreamCreateWithFlags(&P, hipStreamNonBlocking);
    int *stop, *cnt; long long* sp;
    hipHostMalloc(&stop, 4); *stop = 0;
    hipMalloc(&cnt, 64 * 4); hipMalloc(&sp, 8);
    unsigned *wr, *wl; hipMalloc(&wr, 4096 * 4); hipMalloc(&wl, 4096 * 4); hipMemset(wr, 0, 4096 * 4);
    uint32_t mask[8]; for (int i = 0; i < 8; ++i) mask[i] = 0xffffffffu; mask[0] = 0xffffff00u;
    hipStream_t E; printf("create: %s\n", hipGetErrorString(hipExtStreamCreateWithCUMask(&E, 8, mask)));
    hipFuncSetAttribute((const void*)resident, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipMemset(cnt, 0, 64 * 4); hipDeviceSynchronize();
    hipLaunchKernelGGL(resident, dim3(wgs), dim3(256), 65536, E, stop, cnt, vacate, wr);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int G : {4, 16, 32, 32, 32}) {
        for (int threads : {320, 576}) {
            hipMemsetAsync(cnt + 16, 0, 48 * 4, P);
            hipEventRecord(e0, P);
            hipLaunchKernelGGL(leaf, dim3(8 * G), dim3(threads > 320 ? 320 : threads), 0, P, G, cnt + 16, cnt + 32, sp, wl);
            hipEventRecord(e1, P);
            hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            std::vector<int> h(64); hipMemcpyAsync(h.data(), cnt, 256, hipMemcpyDeviceToHost, P); hipStreamSynchronize(P);
            long long s = 0; hipMemcpyAsync(&s, sp, 8, hipMemcpyDeviceToHost, P); hipStreamSynchronize(P);
            printf("G=%d: %.3f ms, arrived %d, spins %lld, per XCC:", G, ms, h[16], s);
            for (int x = 0; x < 8; ++x) printf(" %d", h[32 + x]);
            printf("   resident per XCC:"); for (int x = 0; x < 8; ++x) printf(" %d", h[x]); printf("\n");
            fflush(stdout);
        }
    }
    {
        std::vector<unsigned> hr(4096), hl(4096);
        hipMemcpyAsync(hr.data(), wr, 4096 * 4, hipMemcpyDeviceToHost, P); hipMemcpyAsync(hl.data(), wl, 4096 * 4, hipMemcpyDeviceToHost, P); hipStreamSynchronize(P);
        // HW_ID (gfx9): wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
        for (int x = 0; x < 8; ++x) {
            int rc[256] = {0}, lc[256] = {0};
            for (int b = 0; b < wgs; ++b) if ((hr[b] >> 31) && ((hr[b] >> 16) & 0xf) == (unsigned)x) rc[(hr[b] >> 8) & 0xff]++;
            for (int b = 0; b < 256; ++b) if ((hl[b] >> 31) && ((hl[b] >> 16) & 0xf) == (unsigned)x) lc[(hl[b] >> 8) & 0xff]++;
            printf("XCC %d resident per CU(se.sh.cu):", x);
            int ncu = 0;
            for (int c = 0; c < 256; ++c) if (rc[c]) { printf(" %x:%d", c, rc[c]); ++ncu; }
            printf("  [%d CUs]\n        last leaf per CU:", ncu);
            for (int c = 0; c < 256; ++c) if (lc[c]) printf(" %x:%d", c, lc[c]);
            printf("\n");
        }
    }
    *stop = 1;
    hipStreamSynchronize(E);
    printf("done\n");
    return 0;
}
