// Does hipExtStreamCreateWithCUMask work here, and which physical CUs/XCDs do the mask bits select?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <set>
__global__ void where(unsigned* out) {
    unsigned xcc = 0, hwid = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc & 0xf; out[2 * blockIdx.x + 1] = hwid; }
    for (volatile int i = 0; i < 20000; ++i) {}
}
int main() {
    unsigned* d; hipMalloc(&d, 8 * 4096);
    for (int variant = 0; variant < 3; ++variant) {
        std::vector<uint32_t> mask(8, 0);
        for (int i = 0; i < 256; ++i) {
            bool on = variant == 0 ? (i % 8 == 0) : variant == 1 ? (i < 32) : (i % 8 != 0);
            if (on) mask[i / 32] |= 1u << (i % 32);
        }
        hipStream_t s; hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask.data());
        printf("variant %d create: %s\n", variant, hipGetErrorString(e));
        if (e != hipSuccess) continue;
        hipMemsetAsync(d, 0xff, 8 * 4096, s);
        hipLaunchKernelGGL(where, dim3(1024), dim3(64), 0, s, d);
        hipStreamSynchronize(s);
        std::vector<unsigned> h(2048); hipMemcpy(h.data(), d, 8 * 1024, hipMemcpyDeviceToHost);
        std::set<unsigned> cus; int per_xcc[16] = {0};
        for (int b = 0; b < 1024; ++b) { unsigned xcc = h[2*b], hw = h[2*b+1]; unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7; cus.insert((xcc << 16) | (se << 8) | (sh << 4) | cu); }
        for (auto c : cus) per_xcc[c >> 16]++;
        printf("  distinct CUs used: %zu ; per XCC:", cus.size()); for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]); printf("\n");
        hipStreamDestroy(s);
    }
    return 0;
}
