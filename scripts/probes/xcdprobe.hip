// Round trip of a tagged 16-byte granule (sc1 store / sc1 load, as in the panel kernel) between workgroup 0 and workgroup p
// of one launch.  Workgroups are dispatched round-robin over the 8 XCDs, so p = 8, 16 share workgroup 0's XCD (and L2) and
// p = 1..7 sit on another XCD.   hipcc --offload-arch=gfx950 -O3 xcdprobe.hip -o xcdprobe && ./xcdprobe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u4v __attribute__((ext_vector_type(4)));
__global__ void pingpong(unsigned* buf, long long* cyc, int p, int rounds, int aux)
{
    const int me = blockIdx.x;
    if (me != 0 && me != p) return;
    if (threadIdx.x != 0) return;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 1 << 20, 0x00020000);
    const unsigned mine = me == 0 ? 0u : 4096u, theirs = me == 0 ? 4096u : 0u;
    long long t0 = clock64();
    for (int i = 1; i <= rounds; ++i) {
        if (me == 0) {
            u4v x = {(unsigned)i, (unsigned)i, 0u, (unsigned)i};
            if (aux == 16) __builtin_amdgcn_raw_buffer_store_b128(x, r, mine, 0, 16); else if (aux == 1) __builtin_amdgcn_raw_buffer_store_b128(x, r, mine, 0, 1); else __builtin_amdgcn_raw_buffer_store_b128(x, r, mine, 0, 17);
            for (int sp = 0; sp < 200000; ++sp) { asm volatile("" ::: "memory"); u4v y = aux == 16 ? __builtin_amdgcn_raw_buffer_load_b128(r, theirs, 0, 16) : aux == 1 ? __builtin_amdgcn_raw_buffer_load_b128(r, theirs, 0, 1) : __builtin_amdgcn_raw_buffer_load_b128(r, theirs, 0, 17); if (y[1] == (unsigned)i && y[3] == (unsigned)i) break; }
        } else {
            for (int sp = 0; sp < 200000; ++sp) { asm volatile("" ::: "memory"); u4v y = aux == 16 ? __builtin_amdgcn_raw_buffer_load_b128(r, theirs, 0, 16) : aux == 1 ? __builtin_amdgcn_raw_buffer_load_b128(r, theirs, 0, 1) : __builtin_amdgcn_raw_buffer_load_b128(r, theirs, 0, 17); if (y[1] == (unsigned)i && y[3] == (unsigned)i) break; }
            u4v x = {(unsigned)i, (unsigned)i, 0u, (unsigned)i};
            if (aux == 16) __builtin_amdgcn_raw_buffer_store_b128(x, r, mine, 0, 16); else if (aux == 1) __builtin_amdgcn_raw_buffer_store_b128(x, r, mine, 0, 1); else __builtin_amdgcn_raw_buffer_store_b128(x, r, mine, 0, 17);
        }
    }
    if (me == 0) cyc[0] = clock64() - t0;
}
int main()
{
    unsigned* buf; long long* cyc;
    hipMalloc(&buf, 1 << 20); hipMalloc(&cyc, 64);
    const int rounds = 500;   // spins are bounded: a non-coherent mode shows up as an absurd round trip, not as a hang
    for (int aux : {16, 17, 1}) {
        for (int p : {1, 2, 4, 7, 8, 16, 24, 9}) {
            hipMemset(buf, 0, 1 << 20);
            pingpong<<<32, 64>>>(buf, cyc, p, rounds, aux);
            hipError_t e = hipDeviceSynchronize();
            long long c = 0; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            printf("aux=%2d (%s) partner workgroup %2d (xcd %d): round trip %6.0f cycles%s\n", aux, aux == 16 ? "sc1" : aux == 17 ? "sc0+sc1" : "sc0", p, p % 8,
                   (double)c / rounds, e == hipSuccess ? "" : "  ERROR"); fflush(stdout);
        }
    }
    return 0;
}
