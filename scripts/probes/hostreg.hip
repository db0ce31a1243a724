// probe: cost of hipHostRegister / unregister on a pageable 2 GiB buffer, and H2D / D2H rates pageable vs registered
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t bytes = 2ull << 30;
    char* h = (char*)malloc(bytes);
    memset(h, 1, bytes);
    void* d;
    hipMalloc(&d, bytes);
    hipStream_t s;
    hipStreamCreate(&s);
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        hipMemcpy(d, h, bytes, hipMemcpyHostToDevice);
        double t1 = now();
        hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost);
        double t2 = now();
        printf("pageable: H2D %.1f ms (%.1f GB/s)  D2H %.1f ms (%.1f GB/s)\n", t1 - t0, bytes / (t1 - t0) / 1e6, t2 - t1, bytes / (t2 - t1) / 1e6);
    }
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        hipError_t e = hipHostRegister(h, bytes, hipHostRegisterDefault);
        double t1 = now();
        hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s);
        double t1b = now();
        hipStreamSynchronize(s);
        double t2 = now();
        hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s);
        hipStreamSynchronize(s);
        double t3 = now();
        hipHostUnregister(h);
        double t4 = now();
        printf("register %.1f ms (rc %d)  H2D async issue %.2f ms, done %.1f ms (%.1f GB/s)  D2H %.1f ms (%.1f GB/s)  unregister %.1f ms\n", t1 - t0, (int)e, t1b - t1,
               t2 - t1, bytes / (t2 - t1) / 1e6, t3 - t2, bytes / (t3 - t2) / 1e6, t4 - t3);
    }
    // async from pageable: does the call block the host?
    double t0 = now();
    hipMemcpyAsync(d, h, bytes / 8, hipMemcpyHostToDevice, s);
    double t1 = now();
    hipStreamSynchronize(s);
    double t2 = now();
    printf("pageable hipMemcpyAsync of 256 MiB: call returns after %.2f ms, complete after %.2f ms\n", t1 - t0, t2 - t0);
    // both directions at once (registered)
    hipHostRegister(h, bytes, hipHostRegisterDefault);
    hipStream_t s2;
    hipStreamCreate(&s2);
    void* d2;
    hipMalloc(&d2, bytes / 2);
    t0 = now();
    hipMemcpyAsync(d, h, bytes / 2, hipMemcpyHostToDevice, s);
    hipMemcpyAsync(h + bytes / 2, d2, bytes / 2, hipMemcpyDeviceToHost, s2);
    hipStreamSynchronize(s);
    hipStreamSynchronize(s2);
    t1 = now();
    printf("1 GiB H2D + 1 GiB D2H concurrently: %.1f ms\n", t1 - t0);
    return 0;
}
