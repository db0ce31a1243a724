// probe: D2H rate of strided 2-D copies (block rows of a column-major matrix) into pageable host memory
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t n = 16384, sz = 8, bytes = n * n * sz;
    char* h = (char*)malloc(bytes);
    memset(h, 1, bytes);
    char* d;
    hipMalloc((void**)&d, bytes);
    hipMemset(d, 0, bytes);
    hipDeviceSynchronize();
    for (size_t rows : {512, 2048, 4096, 16384}) {
        // block row [0, rows) x all columns: width rows*8 bytes, height n, pitch n*8
        double t0 = now();
        hipMemcpy2D(h, n * sz, d, n * sz, rows * sz, n, hipMemcpyDeviceToHost);
        double t1 = now();
        printf("D2H 2-D width %6zu B x %zu columns (%.0f MiB): %.2f ms = %.1f GB/s\n", rows * sz, n, rows * sz * n / 1048576.0, t1 - t0,
               rows * sz * n / (t1 - t0) / 1e6);
        t0 = now();
        hipMemcpy2D(d, n * sz, h, n * sz, rows * sz, n, hipMemcpyHostToDevice);
        t1 = now();
        printf("H2D 2-D width %6zu B x %zu columns (%.0f MiB): %.2f ms = %.1f GB/s\n", rows * sz, n, rows * sz * n / 1048576.0, t1 - t0,
               rows * sz * n / (t1 - t0) / 1e6);
    }
    // contiguous column chunks
    for (size_t cols : {512, 2048}) {
        double t0 = now();
        hipMemcpy(h, d, cols * n * sz, hipMemcpyDeviceToHost);
        double t1 = now();
        printf("D2H contiguous %zu columns (%.0f MiB): %.2f ms = %.1f GB/s\n", cols, cols * n * sz / 1048576.0, t1 - t0, cols * n * sz / (t1 - t0) / 1e6);
    }
    return 0;
}
