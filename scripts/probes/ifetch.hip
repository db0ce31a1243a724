// cold straight-line code vs the same body in a loop: is a once-executed instruction stream fetch-limited?
#include <hip/hip_runtime.h>
#include <stdio.h>
#define BODY(N) asm volatile(".rept " #N "\n v_add_u32 %0, 0x12345, %0\n .endr" : "+v"(x))
template <int COPIES, int NI>
__global__ void k(unsigned* out, long long* cyc, int loops)
{
    unsigned x = threadIdx.x;
    long long t0 = clock64();
    constexpr int LOOPS = 64 / COPIES;
#pragma unroll 1
    for (int it = 0; it < LOOPS; ++it) {
#pragma unroll
        for (int c = 0; c < COPIES; ++c) {
            if (NI == 50) BODY(50); else if (NI == 150) BODY(150); else if (NI == 300) BODY(300); else BODY(600);
            __syncthreads();
        }
    }
    long long t1 = clock64();
    out[threadIdx.x + blockIdx.x * blockDim.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int COPIES, int NI>
void run(unsigned* out, long long* cyc, int threads, int blocks, const char* nm)
{
    const int loops = 64 / COPIES;
    for (int rep = 0; rep < 2; ++rep) {
        k<COPIES, NI><<<blocks, threads>>>(out, cyc, loops);
        hipDeviceSynchronize();
    }
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-10s body %3d instrs (%4d B) x %2d copies x %2d loops, %3d threads x %3d blocks: %7.1f clk per body = %5.2f clk per instr\n", nm, NI, NI * 8, COPIES, loops, threads, blocks, (double)h / 64, (double)h / 64 / NI);
}
int main()
{
    unsigned* out; long long* cyc; hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 64);
    for (int threads : {64, 576}) for (int blocks : {1, 32}) {
        run<1, 50>(out, cyc, threads, blocks, "loop"); run<8, 50>(out, cyc, threads, blocks, "8 copies"); run<64, 50>(out, cyc, threads, blocks, "64 copies");
        run<1, 150>(out, cyc, threads, blocks, "loop"); run<8, 150>(out, cyc, threads, blocks, "8 copies"); run<64, 150>(out, cyc, threads, blocks, "64 copies");
        run<1, 300>(out, cyc, threads, blocks, "loop"); run<8, 300>(out, cyc, threads, blocks, "8 copies"); run<64, 300>(out, cyc, threads, blocks, "64 copies");
        run<1, 600>(out, cyc, threads, blocks, "loop"); run<8, 600>(out, cyc, threads, blocks, "8 copies"); run<64, 600>(out, cyc, threads, blocks, "64 copies");
    }
    return 0;
}
