// Measures the effective shader clock seen by small, latency-bound kernels: clock64() (shader cycles) vs wall_clock64()
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void probe(long long* out, int iters) {
    long long c0 = clock64(); long long w0 = wall_clock64();
    double x = threadIdx.x;
    for (int i = 0; i < iters; ++i) x = x * 1.0000001 + 0.5;
    long long c1 = clock64(); long long w1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = (long long)x; }
}
__global__ void heavy(double* p, int n) {
    double x = p[threadIdx.x];
    for (int i = 0; i < n; ++i) x = x * 1.0000001 + 0.5;
    p[threadIdx.x + blockIdx.x * blockDim.x] = x;
}
int main() {
    long long* d; hipMalloc(&d, 64); double* hp; hipMalloc(&hp, 8 * 256 * 4096);
    int wallrate = 0; hipDeviceGetAttribute(&wallrate, hipDeviceAttributeWallClockRate, 0);
    printf("wall clock rate kHz: %d\n", wallrate);
    long long h[3];
    auto run = [&](const char* tag, int iters) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, iters); hipDeviceSynchronize();
        hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("%-28s shader cycles %lld wall ticks %lld -> %.0f MHz\n", tag, h[0], h[1], (double)h[0] / ((double)h[1] / (wallrate * 1e3)) / 1e6);
    };
    run("cold small", 2000);
    run("small again", 2000);
    for (int i = 0; i < 20; ++i) run("small seq", 2000);
    hipLaunchKernelGGL(heavy, dim3(4096), dim3(256), 0, 0, hp, 2000000); 
    run("right after heavy (queued)", 2000);
    hipDeviceSynchronize();
    for (int i = 0; i < 5; ++i) run("after heavy", 2000);
    run("long single-wave", 2000000);
    return 0;
}
