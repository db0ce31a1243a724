// Which CU masks does the runtime accept, and where do the workgroups of a masked queue go?  (round 2)
// Mask bit i selects CU i/8 of XCC i%8 (scripts/probes/cumask.hip).  Variants probe whether a queue can be kept off ONE
// whole XCC (so that XCC can be reserved for an XCD-local cooperative kernel) and how the dispatcher then distributes
// workgroups over the XCCs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <set>
#include <vector>
__global__ void where(unsigned* out)
{
    unsigned xcc = 0, hwid = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc & 0xf; out[2 * blockIdx.x + 1] = hwid; }
    for (volatile int i = 0; i < 20000; ++i) {}
}
int main()
{
    unsigned* d;
    hipMalloc(&d, 8 * 4096);
    const char* names[] = {"only XCC0 (32 CUs)", "all but XCC0 (224 CUs)", "XCC0 full + 1 CU of every other XCC",
                           "all but XCC0, plus 1 CU of XCC0", "all but XCC0 and XCC1", "XCC0 + XCC1 only"};
    for (int variant = 0; variant < 6; ++variant) {
        std::vector<uint32_t> mask(8, 0);
        for (int i = 0; i < 256; ++i) {
            const int xcc = i % 8, cu = i / 8;
            bool on = false;
            switch (variant) {
                case 0: on = xcc == 0; break;
                case 1: on = xcc != 0; break;
                case 2: on = xcc == 0 || cu == 0; break;
                case 3: on = xcc != 0 || cu == 0; break;
                case 4: on = xcc >= 2; break;
                case 5: on = xcc < 2; break;
            }
            if (on) mask[i / 32] |= 1u << (i % 32);
        }
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask.data());
        printf("variant %d [%s] create: %s\n", variant, names[variant], hipGetErrorString(e));
        if (e != hipSuccess) continue;
        hipMemsetAsync(d, 0xff, 8 * 4096, s);
        hipLaunchKernelGGL(where, dim3(1024), dim3(64), 0, s, d);
        e = hipStreamSynchronize(s);
        if (e != hipSuccess) { printf("  sync: %s\n", hipGetErrorString(e)); return 1; }
        std::vector<unsigned> h(2048);
        hipMemcpy(h.data(), d, 8 * 1024, hipMemcpyDeviceToHost);
        std::set<unsigned> cus;
        int per_xcc[16] = {0}, wg_xcc[16] = {0}, modmatch = 0;
        for (int b = 0; b < 1024; ++b) {
            unsigned xcc = h[2 * b], hw = h[2 * b + 1];
            unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
            cus.insert((xcc << 16) | (se << 8) | (sh << 4) | cu);
            wg_xcc[xcc & 15]++;
            if ((int)xcc == b % 8) ++modmatch;
        }
        for (auto c : cus) per_xcc[c >> 16]++;
        printf("  distinct CUs used: %zu ; CUs per XCC:", cus.size());
        for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
        printf(" ; WGs per XCC:");
        for (int x = 0; x < 8; ++x) printf(" %d", wg_xcc[x]);
        printf(" ; xcc==b%%8 for %d of 1024\n", modmatch);
        fflush(stdout);
        hipStreamDestroy(s);
    }
    return 0;
}
