// Which SIMD does wave w of a workgroup land on?  (the cooperative leaf wants its communication wave alone on a SIMD)
// build: hipcc --offload-arch=gfx950 -O2 -o scripts/probes/bin/simdmap scripts/probes/simdmap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned* out, int early_exit_mask)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    if (lane == 0) out[blockIdx.x * 32 + wave] = hwid;
    if ((early_exit_mask >> wave) & 1) return;
    __syncthreads();
    // a second look after the early waves are gone (placement does not change, but make sure nothing hangs)
    if (lane == 0) out[blockIdx.x * 32 + 16 + wave] = hwid;
}
int main()
{
    unsigned* d;
    hipMalloc(&d, 64 * 32 * 4);
    for (int nw : {4, 5, 8, 9, 12}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(d, 0xff, 64 * 32 * 4);
            const int mask = (nw == 9 && rep == 1) ? 0x11 : 0;
            hipLaunchKernelGGL(probe, dim3(4), dim3(nw * 64), 0, 0, d, mask);
            hipDeviceSynchronize();
            unsigned h[4 * 32];
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            for (int b = 0; b < 4; ++b) {
                printf("waves=%2d exitmask=%#x block %d: SIMD of wave 0..: ", nw, mask, b);
                for (int w = 0; w < nw; ++w) printf("%u ", (h[b * 32 + w] >> 4) & 3);
                printf("  (CU %u SE %u)\n", (h[b * 32] >> 8) & 15, (h[b * 32] >> 13) & 7);
            }
        }
    }
    return 0;
}
