// Latency/throughput probes for the building blocks of the latency-bound kernels (panel, trsm base).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N 4096
__global__ void fma_chain(double* out, long long* cyc, int nacc) {
    double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4=a0+4,a5=a0+5,a6=a0+6,a7=a0+7;
    const double m = out[1], c = out[2];
    long long t0 = clock64();
    if (nacc == 1) { for (int i = 0; i < N; ++i) { a0 = a0 * m + c; } }
    else if (nacc == 2) { for (int i = 0; i < N / 2; ++i) { a0 = a0 * m + c; a1 = a1 * m + c; } }
    else if (nacc == 4) { for (int i = 0; i < N / 4; ++i) { a0 = a0 * m + c; a1 = a1 * m + c; a2 = a2 * m + c; a3 = a3 * m + c; } }
    else { for (int i = 0; i < N / 8; ++i) { a0 = a0 * m + c; a1 = a1 * m + c; a2 = a2 * m + c; a3 = a3 * m + c; a4 = a4 * m + c; a5 = a5 * m + c; a6 = a6 * m + c; a7 = a7 * m + c; } }
    long long t1 = clock64();
    out[8 + threadIdx.x] = a0 + a1 + a2 + a3+a4+a5+a6+a7;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void f32_chain(float* out, long long* cyc) {
    float a0 = threadIdx.x; const float m = out[1], c = out[2];
    long long t0 = clock64();
    for (int i = 0; i < N; ++i) a0 = a0 * m + c;
    long long t1 = clock64();
    out[8 + threadIdx.x] = a0; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void cmp_chain(double* out, long long* cyc) {
    double v = out[threadIdx.x + 8]; unsigned p = threadIdx.x; const double* src = out + 100;
    long long t0 = clock64();
    for (int i = 0; i < N; ++i) { double ov = src[i & 7] ; unsigned op = i; if (ov > v || (ov == v && op < p)) { v = ov; p = op; } }
    long long t1 = clock64();
    out[8 + threadIdx.x] = v + p; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void bperm_chain(int* out, long long* cyc) {
    int v = threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < N; ++i) v = __shfl_xor(v, 32) + 1;
    long long t1 = clock64();
    out[threadIdx.x] = v; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void dpp_chain(int* out, long long* cyc) {
    int v = threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < N; ++i) v = __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true) + 1;
    long long t1 = clock64();
    out[threadIdx.x] = v; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void lds_chain(int* out, long long* cyc) {
    __shared__ int s[256];
    s[threadIdx.x] = (threadIdx.x + 1) & 63; __syncthreads();
    int v = threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < N; ++i) v = s[v];
    long long t1 = clock64();
    out[threadIdx.x] = v; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void barrier_chain(int* out, long long* cyc) {
    long long t0 = clock64();
    for (int i = 0; i < N; ++i) __syncthreads();
    long long t1 = clock64();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void gload_chain(unsigned long long* buf, long long* cyc, int mode) {
    unsigned long long idx = threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < 1024; ++i) {
        if (mode == 0) idx = buf[idx];
        else idx = __hip_atomic_load(buf + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    long long t1 = clock64();
    buf[4096 + threadIdx.x] = idx; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// ping-pong between two workgroups (different CUs): round-trip of an 8-byte tagged granule
__global__ void pingpong(unsigned long long* flags, long long* cyc, int rounds) {
    const int me = blockIdx.x;
    long long t0 = clock64();
    if (threadIdx.x == 0) {
        for (int r = 1; r <= rounds; ++r) {
            if (me == 0) {
                __hip_atomic_store(flags + 0, (unsigned long long)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(flags + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)r) {}
            } else {
                while (__hip_atomic_load(flags + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)r) {}
                __hip_atomic_store(flags + 64, (unsigned long long)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    long long t1 = clock64();
    if (threadIdx.x == 0 && me == 0) cyc[0] = t1 - t0;
}
int main() {
    double* d; hipMalloc(&d, 8 * 4096); long long* c; hipMalloc(&c, 64); long long h;
    double init[3] = {0, 1.0000001, 0.5}; hipMemcpy(d, init, 24, hipMemcpyHostToDevice);
    auto rep = [&](const char* name, double per) { hipDeviceSynchronize(); hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost); printf("%-44s %8.1f cycles/op\n", name, (double)h / per); };
    for (int na : {1, 2, 4, 8}) { hipLaunchKernelGGL(fma_chain, dim3(1), dim3(64), 0, 0, d, c, na); char b[64]; sprintf(b, "v_fma_f64, %d independent chains, 1 wave", na); rep(b, N); }
    hipLaunchKernelGGL(fma_chain, dim3(1), dim3(256), 0, 0, d, c, 1); rep("v_fma_f64 dependent, 4 waves (1/SIMD)", N);
    hipLaunchKernelGGL(f32_chain, dim3(1), dim3(64), 0, 0, (float*)d, c); rep("v_fma_f32 dependent chain", N);
    hipLaunchKernelGGL(cmp_chain, dim3(1), dim3(64), 0, 0, d, c); rep("f64 (key,pos) compare-select step", N);
    hipLaunchKernelGGL(bperm_chain, dim3(1), dim3(64), 0, 0, (int*)d, c); rep("__shfl_xor(32) dependent (ds_bpermute)", N);
    hipLaunchKernelGGL(dpp_chain, dim3(1), dim3(64), 0, 0, (int*)d, c); rep("DPP quad_perm mov + add dependent", N);
    hipLaunchKernelGGL(lds_chain, dim3(1), dim3(64), 0, 0, (int*)d, c); rep("LDS dependent read (pointer chase)", N);
    hipLaunchKernelGGL(barrier_chain, dim3(1), dim3(256), 0, 0, (int*)d, c); rep("__syncthreads, 4 waves", N);
    unsigned long long* g; hipMalloc(&g, 8 * 8192); unsigned long long hb[4096]; for (int i = 0; i < 4096; ++i) hb[i] = (i * 17 + 64) % 4096; hipMemcpy(g, hb, sizeof(hb), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(gload_chain, dim3(1), dim3(64), 0, 0, g, c, 0); rep("global load dependent (L1/L2 hit, plain)", 1024);
    hipLaunchKernelGGL(gload_chain, dim3(1), dim3(64), 0, 0, g, c, 1); rep("global load dependent (relaxed agent = sc1)", 1024);
    for (int pair = 0; pair < 3; ++pair) {
        hipMemset(g, 0, 8 * 8192);
        // blocks 0 and 1 land on XCD 0 and XCD 1 (b % 8); use gridDim 2 -> different XCDs. For same-XCD use blocks 0 and 8 (grid 9, others idle)
        hipLaunchKernelGGL(pingpong, dim3(2), dim3(64), 0, 0, g, c, 2000); rep("ping-pong round trip WG0<->WG1 (cross-XCD)", 2000);
    }
    return 0;
}
