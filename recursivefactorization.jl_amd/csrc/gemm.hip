// gemm.hip -- trailing update  C <- C - A*B  on the MFMA matrix cores of gfx950 (CDNA4).
//
// Replaces schur_complement! (/root/reference/src/lu.jl:265-284: C[m,n] = (0 - sum_k A[m,k]*B[k,n]) + C[m,n]) and
// supplies the off-diagonal work of the recursive TRSM (src/lu.jl:235).  Dominant kernel of the path: N^3/2 of the
// 2N^3/3 LU flops plus ~90 % of the TRSM's N^3/6.
//
// Shape: all operands row-major (R layout).  One workgroup = 256 threads = 4 waves computes a 128x128 tile of C; each
// wave owns a 64x64 sub-tile as 4x4 MFMA fragments of 16x16 (v_mfma_f64_16x16x4_f64, or v_mfma_f32_16x16x4_f32 for
// Float32 -- both take ONE scalar per lane for A and B).  K is walked in slabs of 16 through a double-buffered LDS
// image (one barrier per slab): global loads for slab t+1 are issued before the 64 MFMAs of slab t and written to the
// other LDS buffer after them, so HBM/L2 latency hides behind ~4096 cycles of matrix work per wave.
//
// LDS images.  hipcc pairs the fragment reads into ds_read2_b64 and the staging writes are ds_write_b64: both are serviced
// in groups of 16 CONSECUTIVE lanes on a 32-bank map, bank = (byte address / 4) % 32 (MI355X_MICROARCH.md "LDS").  Round 1
// laid the images out for the 64-bank / 32-lane rule of a lone ds_read_b64 and paid for it: SQ_LDS_BANK_CONFLICT 8.1e9 cycles
// where the vendor BLAS kernel of the same tile shape shows 0 (scripts/pmc_gemm.sh), the B staging writes 8-way conflicted.
//   As[i][k], row stride SA = BK+1 = 17 doubles: the 16 lanes of a group read rows i..i+15 at one k -> dword 34*i, and
//             34*i mod 32 = 2*i: 16 distinct bank pairs; the staging writes (8 rows x 2 k-halves per group) land on
//             2*r + {0,16}: distinct as well
//   Bs[k][c + (c >> 4)], row stride SB = 144 doubles: a one-double gap after every 16 columns.  A group's fragment read
//             covers 16 consecutive columns of one block (contiguous, all 32 banks once); a staging write has lane l at
//             columns 8l..8l+7 -> dword 16*l + 2*(l/2) + 2e = {2a, 16+2a}: distinct bank pairs (was: 16*l -> 8-way)
// Two workgroups per CU (2 x 72 KiB LDS, <=256 VGPRs) keep each SIMD's matrix pipe fed across barriers.
//
// Roofline: bound = fp64 MFMA, 78.6 TFLOP/s chip peak (256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz);
// algorithmic work 2*M*N*K flops per launch.
#include "gemm_tile.hpp"

namespace rflu {

// C_FIRST: the accumulators start as the C tile (its read overlaps with the first operand slabs, the epilogue only
// stores): +28 % at K = 256, +8 % at K = 512 and, since the negation moved onto the MFMA (round 3), +3..7 % from K = 1024 up too.
template <typename T, bool C_FIRST>
__global__ void __launch_bounds__(256, 2) gemm_sub_kernel(GemmArgs<T> g)
{
    typedef typename Mfma<T>::acc_t acc_t;
    constexpr int VW = 16 / (int)sizeof(T);  // elements per 16-byte vector
    typedef T vec_t __attribute__((ext_vector_type(VW)));
    // C_FIRST computes c - sum a*b in the accumulators.  Round 2 negated every A element on its way into LDS (8 VALU ops per thread
    // and slab, between the wait for the global loads and the LDS writes of the slab transition).  Float64: the MFMA negates A itself
    // (NEGMOD; sustained 15872^2 x 512: 61.4 -> 64.4 TFLOP/s).  Float32 has no such modifier: the accumulators hold -(c - sum a*b)
    // instead, negated once on the way in and once on the way out (NEGACC; negation is exact, so the results are bit-identical).
    constexpr bool NEGMOD = C_FIRST && Mfma<T>::HAS_NEG;
    constexpr bool NEGACC = C_FIRST && !Mfma<T>::HAS_NEG;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* smem = reinterpret_cast<T*>(smem_raw);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    // ---- workgroup -> tile: bijective XCD-aware remap (block b runs on XCD b % 8; give each XCD a contiguous range
    // of tiles so neighbours share operand panels in that XCD's private L2), then GROUP_M-row grouped ordering.
    int tile_m, tile_n;
    const int n_first = g.na_tiles_n * g.tiles_m;                 // workgroups of the "first columns" region (0: none)
    const bool in_first = (int)blockIdx.x < n_first;
    {
        // two regions, each with the remap of its own: blocks [0, n_first) -> tile columns [0, na), the rest -> [na, tiles_n)
        const int nwg = in_first ? n_first : (int)gridDim.x - n_first;
        const int bid = in_first ? (int)blockIdx.x : (int)blockIdx.x - n_first;
        const int tn = in_first ? g.na_tiles_n : g.tiles_n - g.na_tiles_n;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
        const int per_group = G_GROUP_M * tn;
        const int group = wg / per_group;
        const int first_m = group * G_GROUP_M;
        const int gsz = min(g.tiles_m - first_m, G_GROUP_M);
        const int in_group = wg - group * per_group;
        tile_m = first_m + in_group % gsz;
        tile_n = in_group / gsz + (in_first ? 0 : g.na_tiles_n);
    }
    const int m0 = tile_m * G_BM, n0 = tile_n * G_BN;

    const bool full_mn = g.vec_ok && (m0 + G_BM <= g.M) && (n0 + G_BN <= g.N);
    // INTERIOR (workgroup-uniform): a full tile of an aligned problem whose K is a multiple of the slab depth and at least two
    // slabs -- every bulk update.  Its copy of the tile code has no bounds test and, more important, no branch around a load: the
    // compiler places a wait for "all but the n most recent loads" only where it can COUNT the loads issued since, and behind any
    // join it waits for everything.
    if (full_mn && (g.K % G_BK) == 0 && g.K >= 2 * G_BK) gemm_tile<T, C_FIRST, true>(g, smem, m0, n0);
    else gemm_tile<T, C_FIRST, false>(g, smem, m0, n0);
    if (in_first && g.sig_flag) {   // workgroup-uniform
        __syncthreads();
        if (tid == 0) {
            __threadfence();
            if (atomicInc(g.sig_cnt, (unsigned)n_first - 1) == (unsigned)n_first - 1)
                __hip_atomic_store(g.sig_flag, g.sig_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---- latency-bound updates of the panel recursion: K a small multiple of 64, few tiles -------------------------------------
// The recursion inside a block column issues C -= A*B with K = 64, 128, (256) and N = K on the critical path.  With the
// kernel above such a launch is a chain of K/16 dependent {global load -> LDS -> barrier -> MFMA} rounds (18 / 28 us
// standalone at M = 16384; 2-3x that next to the bulk update that owns most CUs).  Here a workgroup (4 waves) takes a
// 64x64 tile and issues EVERY load of a 64-wide K chunk at once:
//   * A never touches LDS: with the K index permuted as k = 16*kk + s (kk = lane>>4, s = MFMA step), the A operand of
//     lane (i, kk) for the 16 steps of a chunk is 16 CONTIGUOUS elements of row i -- four 16-byte loads;
//   * B (64 x 64 per chunk) goes through LDS once.  Layout Bs[k][c + (c >> 4)], row stride 68 doubles: a staging write has the
//     16 lanes of a group at rows r..r+3 x column blocks q = 0..3 -> dword 136*r + 34*q = 8*r + 2*q (mod 32): 16 distinct bank
//     pairs (the old odd stride 65 put the four column blocks of a row on the same banks: 4-way conflicts, 38 % of the
//     kernel's LDS cycles); a fragment read covers 16 consecutive columns of one block;
//   * the C tile is read into registers up front, so one memory round trip covers everything the tile needs.
// The sum over k is the same set of products; only the order inside the MFMA accumulation differs.
constexpr int S_BM = 64, S_BN = 64, S_KC = 64;
constexpr int S_SB = S_BN + 4;   // 64 columns + one gap double per 16-column block: column c sits at c + (c >> 4)

template <typename T, int NCHUNK>
__global__ void __launch_bounds__(256) gemm_skinny_kernel(GemmArgs<T> g)
{
    typedef typename Mfma<T>::acc_t acc_t;
    constexpr int VW = 16 / (int)sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(VW)));
    __shared__ T Bs[NCHUNK][S_KC * S_SB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile_m = blockIdx.x % g.tiles_m, tile_n = blockIdx.x / g.tiles_m;
    const int m0 = tile_m * S_BM, n0 = tile_n * S_BN;
    const int li = lane & 15, kk = lane >> 4;
    const int arow = m0 + wave * 16 + li;
    const bool arow_ok = arow < g.M;
    const T* Ap = g.A + (int64_t)(arow_ok ? arow : 0) * g.lda + kk * 16;

    // every load of the tile is issued before the first use: A operands (registers), B chunks, the C tile
    T ra[NCHUNK][16];
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
#pragma unroll
        for (int v = 0; v < 16 / VW; ++v) {
            const vec_t x = *reinterpret_cast<const vec_t*>(Ap + c * S_KC + v * VW);
#pragma unroll
            for (int e = 0; e < VW; ++e) ra[c][v * VW + e] = arow_ok ? x[e] : T(0);
        }
    }
    // B chunk: 64 rows x 64 columns, 16 elements per thread (row tid>>2, columns (tid&3)*16 ..)
    const int brow = tid >> 2, bcol = (tid & 3) * 16;
    T rb[NCHUNK][16];
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
        const T* Bp = g.B + (int64_t)(c * S_KC + brow) * g.ldb + n0 + bcol;
        if (n0 + S_BN <= g.N) {
#pragma unroll
            for (int v = 0; v < 16 / VW; ++v) {
                const vec_t x = *reinterpret_cast<const vec_t*>(Bp + v * VW);
#pragma unroll
                for (int e = 0; e < VW; ++e) rb[c][v * VW + e] = x[e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) rb[c][e] = (n0 + bcol + e < g.N) ? Bp[e] : T(0);
        }
    }
    // C tile: for a fixed (j, r) sixteen lanes cover 16 consecutive columns of one row
    T cin[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = m0 + wave * 16 + Mfma<T>::crow(lane, r);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + j * 16 + li;
            cin[j][r] = (row < g.M && col < g.N) ? g.C[(int64_t)row * g.ldc + col] : T(0);
        }
    }
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
#pragma unroll
        for (int e = 0; e < 16; ++e) Bs[c][brow * S_SB + bcol + (bcol >> 4) + e] = rb[c][e];   // bank-conflict free, see below
    }
    __syncthreads();
    acc_t acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = acc_t{0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            const T* brow_p = &Bs[c][(kk * 16 + st) * S_SB + li];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = Mfma<T>::run(ra[c][st], brow_p[j * 17], acc[j]);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = m0 + wave * 16 + Mfma<T>::crow(lane, r);
        if (row < g.M) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = n0 + j * 16 + li;
                if (col < g.N) g.C[(int64_t)row * g.ldc + col] = cin[j][r] - acc[j][r];
            }
        }
    }
}

template <typename T>
int launch_gemm(Handle* h, int64_t M, int64_t N, int64_t K, const T* A, int64_t lda, const T* B, int64_t ldb, T* C,
                int64_t ldc, GemmSignal sig)
{
    if (M <= 0 || N <= 0 || K <= 0) return RFLU_OK;
    GemmArgs<T> g;
    g.na_tiles_n = 0; g.sig_flag = nullptr; g.sig_val = 0; g.sig_cnt = nullptr;
#if defined(RFLU_GEMM_TRACE)
    g.stamps = g_gemm_stamps;
#endif
    g.M = (int)M; g.N = (int)N; g.K = (int)K;
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
    g.tiles_m = (int)((M + G_BM - 1) / G_BM);
    g.tiles_n = (int)((N + G_BN - 1) / G_BN);
    constexpr int VW = 16 / (int)sizeof(T);
    g.vec_ok = ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) % 16 == 0) && (lda % VW == 0) &&
               (ldb % VW == 0);
    g.flags = h->tune.gemm_flags;
    {   // small K, few tiles: the latency-optimised kernel (measured crossover, scripts/microbench_gemm_small.py)
        const int skinny_max_k = h->tune.skinny_max_k, skinny_wide = h->tune.skinny_wide;
        if (g.vec_ok && (K == S_KC || K == 2 * S_KC) && K <= skinny_max_k && (N <= 2 * K || (skinny_wide && K == S_KC))) {
            g.tiles_m = (int)((M + S_BM - 1) / S_BM);
            g.tiles_n = (int)((N + S_BN - 1) / S_BN);
            ProfScope ps(h, RFLU_K_GEMM_SMALL, 2.0 * (double)M * (double)N * (double)K,
                         sizeof(T) * ((double)M * K + (double)K * N + 2.0 * (double)M * N));
            const dim3 grid((unsigned)(g.tiles_m * g.tiles_n));
            if (K == S_KC) hipLaunchKernelGGL((gemm_skinny_kernel<T, 1>), grid, dim3(256), 0, h->stream, g);
            else           hipLaunchKernelGGL((gemm_skinny_kernel<T, 2>), grid, dim3(256), 0, h->stream, g);
            RFLU_HIP(hipGetLastError());
            return RFLU_OK;
        }
    }
    const size_t lds = 2 * (size_t)G_STAGE * sizeof(T);
    bool& attr_set = h->gemm_attr_set[sizeof(T) == 8 ? 0 : 1];   // per handle = per device (not a process-wide static)
    if (!attr_set) {
        RFLU_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_sub_kernel<T, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        RFLU_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_sub_kernel<T, false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    // K < 256: the leaf-wise schedule's per-leaf updates (K = 64) and the recursion's small merges -- HBM/latency-bound launches
    // that are accounted with the small-K kernel, so that RFLU_K_GEMM is the MFMA-bound bulk update alone
    ProfScope ps(h, K < 256 ? RFLU_K_GEMM_SMALL : RFLU_K_GEMM, 2.0 * (double)M * (double)N * (double)K,
                 sizeof(T) * ((double)M * K + (double)K * N + 2.0 * (double)M * N));  // A, B once; C in and out
    if (sig.flag && sig.first_cols > 0) {   // only the tiled kernel knows the two-region order (callers use it for K >= 256)
        g.na_tiles_n = (int)std::min<int64_t>((sig.first_cols + G_BN - 1) / G_BN, g.tiles_n);
        g.sig_flag = sig.flag; g.sig_val = sig.val; g.sig_cnt = sig.cnt;
    }
    const int64_t nwg = (int64_t)g.tiles_m * g.tiles_n;
    // C_FIRST (accumulators start as the C tile) for every K since round 3: with the products' sign on the MFMA it also wins at
    // large K (15360^2 x K, TFLOP/s: K=1024 62.7 -> 67.2, 2048 67.4 -> 69.2, 4096 68.0 -> 69.7; N=65536 3219 -> 3124 ms); round 2
    // switched to the late read-modify-write from K = 1024 on because the per-slab negation cost more there
    // (RFLU_GEMM_CFIRST_BELOW=1024 restores that)
    const int64_t cfirst_below = h->tune.gemm_cfirst_below;
    if (K < cfirst_below) hipLaunchKernelGGL((gemm_sub_kernel<T, true>), dim3((unsigned)nwg), dim3(256), lds, h->stream, g);
    else          hipLaunchKernelGGL((gemm_sub_kernel<T, false>), dim3((unsigned)nwg), dim3(256), lds, h->stream, g);
    RFLU_HIP(hipGetLastError());
    return RFLU_OK;
}

template int launch_gemm<double>(Handle*, int64_t, int64_t, int64_t, const double*, int64_t, const double*, int64_t,
                                 double*, int64_t, GemmSignal);
template int launch_gemm<float>(Handle*, int64_t, int64_t, int64_t, const float*, int64_t, const float*, int64_t, float*,
                                int64_t, GemmSignal);


// ---- clock keeper ----------------------------------------------------------------------------------------------------
// Register-only fp64 MFMA loop, no memory traffic: `wgs` workgroups of 256 threads spin for `iters` x 64 MFMAs per wave.
// MI355X power management lowers the clock while the chip is lightly loaded (a panel on 32 CUs) and takes ~15 ms of
// sustained load to bring it back (scripts/microbench_gemm_seq.py: a 3 ms idle gap costs the next GEMM 11 %); see
// DESIGN.md "clocks".  Operands are lane-dependent non-trivial values: with zeros the power draw -- and the effect -- vanish.
__global__ void __launch_bounds__(256) heat_kernel(int iters, double* sink)
{
    typedef double acc_t __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63;
    double a = 0.37 + 0.001 * lane, b = -0.61 + 0.002 * lane;
    acc_t c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0}, c3 = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, b, c3, 0, 0, 0);
        }
        a = -a;
    }
    const double r = c0[0] + c1[1] + c2[2] + c3[3];
    if (r == 12345.678 && sink) sink[0] = r;   // never true: keeps the loop alive
}

// ~`usec` microseconds of MFMA load on `cus` CUs of the stream h->stream (64 MFMAs x 64 clocks per iteration per wave,
// one wave per SIMD, at a nominal 2.4 GHz)
int launch_heat(Handle* h, int cus, double usec)
{
    const int iters = (int)(usec * 2400.0 / (64.0 * 64.0));
    if (iters <= 0 || cus <= 0) return RFLU_OK;
    hipLaunchKernelGGL(heat_kernel, dim3((unsigned)cus), dim3(256), 0, h->stream, iters, (double*)nullptr);
    RFLU_HIP(hipGetLastError());
    return RFLU_OK;
}

}  // namespace rflu
