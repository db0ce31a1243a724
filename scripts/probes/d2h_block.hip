// Does a device-to-host copy into PAGEABLE memory on stream C wait for unrelated work on stream A?
// (host entry: the way back of finished block rows while the factorization still runs)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void spin(long long ticks) { long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8); }
__global__ void touch(double* p) { p[threadIdx.x] += 1.0; }
static double ms(std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); }
int main()
{
    const size_t rows = 2048, cols = 16384, ld = 16384;
    double* d; CK(hipMalloc(&d, ld * cols * 8));
    double* pageable = (double*)malloc(ld * cols * 8); memset(pageable, 0, ld * cols * 8);
    double* pinned; CK(hipHostMalloc(&pinned, rows * cols * 8));
    hipStream_t A, Cn, Cm;
    CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&Cn, hipStreamNonBlocking));
    uint32_t mask[8]; for (int i = 0; i < 8; ++i) mask[i] = i < 3 ? 0u : 0xffffffffu;
    CK(hipExtStreamCreateWithCUMask(&Cm, 8, mask));
    hipStream_t Am;   // the busy stream as a CU-masked one (such streams cannot be created non-blocking)
    { uint32_t mk[8]; for (int i = 0; i < 8; ++i) mk[i] = i < 1 ? 0u : 0xffffffffu; CK(hipExtStreamCreateWithCUMask(&Am, 8, mk)); }
    for (int variant = 0; variant < 12; ++variant) {
        hipStream_t C = (variant & 1) ? Cm : Cn;
        const bool busy_masked = variant >= 6;
        if (busy_masked) A = Am;
        const int kind = (variant % 6) >> 1;   // 0: 2-D pageable, 1: 1-D pageable, 2: 2-D pinned
        CK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        if (getenv("SATURATE")) hipLaunchKernelGGL(spin, dim3(256 * 8 * 50), dim3(256), 0, A, 200000LL);   // every CU slot taken, 50 rounds of 2 ms
        else hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, A, 10000000LL);   // 100 ms on stream A
        hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, C, d);
        if (kind == 0) CK(hipMemcpy2DAsync(pageable, ld * 8, d, ld * 8, rows * 8, cols, hipMemcpyDeviceToHost, C));
        if (kind == 1) CK(hipMemcpyAsync(pageable, d, rows * cols * 8, hipMemcpyDeviceToHost, C));
        if (kind == 2) CK(hipMemcpy2DAsync(pinned, rows * 8, d, ld * 8, rows * 8, cols, hipMemcpyDeviceToHost, C));
        const double t_call = ms(t0);
        CK(hipStreamSynchronize(C));
        const double t_done = ms(t0);
        CK(hipStreamSynchronize(A));
        printf("busy stream %s; %s, %s stream: call returned %.1f ms, copy complete %.1f ms, stream A done %.1f ms\n",
               busy_masked ? "CU-masked   " : "non-blocking", kind == 0 ? "2-D pageable" : kind == 1 ? "1-D pageable" : "2-D pinned  ", (variant & 1) ? "CU-masked  " : "non-blocking", t_call, t_done, ms(t0));
    }
    return 0;
}
