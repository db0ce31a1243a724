// driver.cpp -- host side of librflu.so: handle, the Toledo recursion, and the C ABI of include/rflu.h.
//
// Host control flow restates /root/reference/src/lu.jl:
//   lu!(A, ipiv, pivot, thread; ...)  (:97-130)  -> rflu_getrf_* : NoPivot identity fill (:111-113), recursion, info
//   _recurse! fat-matrix tail         (:148-154) -> getrf_rm(): TRSM of the columns right of the square part
//   reckernel!                        (:189-263) -> rec(): factor left half, TRSM, Schur GEMM, factor right half
// MI355X-specific re-scheduling (results unchanged):
//   * leaves are 64 columns wide (one cooperative panel kernel, panel.hip) and the split is on 64-column boundaries
//     (the reference's nsplit, :158-162, rounds to 64 BYTES of column; SURVEY.md a2: "GPU picks its own split");
//   * the interchanges of a leaf are applied to ALL other columns right after the leaf (one full-width, perfectly
//     parallel laswp launch) instead of level by level (:233, :246) -- the same swaps in the same order on data that
//     nothing touches in between, hence identical results with log2(N/64) times fewer dependent launches;
//   * ipiv is written with global 1-based rows directly (the reference reaches the same values through P2 .+= n1,
//     :256-260) and info is the global index of the first zero pivot (the reference's offset fix-up :248-255).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <new>

#include "rflu_internal.hpp"

namespace rflu {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static int64_t round_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

int ensure_buffer(void** ptr, size_t* cap, size_t need)
{
    if (*cap >= need) return RFLU_OK;
    if (*ptr) RFLU_HIP(hipFree(*ptr));
    *ptr = nullptr;
    *cap = 0;
    RFLU_HIP(hipMalloc(ptr, need));
    *cap = need;
    return RFLU_OK;
}

int ensure_bookkeeping(Handle* h, int64_t rows)
{
    const int64_t chunks = (rows + NB - 1) / NB + 1;
    if (chunks <= h->pm_chunks) return RFLU_OK;
    if (h->pm_cnt) RFLU_HIP(hipFree(h->pm_cnt));
    if (h->pm_dst) RFLU_HIP(hipFree(h->pm_dst));
    if (h->pm_src) RFLU_HIP(hipFree(h->pm_src));
    if (h->linv) RFLU_HIP(hipFree(h->linv));
    h->linv = nullptr;
    h->pm_cnt = h->pm_dst = h->pm_src = nullptr;
    h->pm_chunks = 0;
    RFLU_HIP(hipMalloc((void**)&h->pm_cnt, (size_t)chunks * sizeof(int)));
    RFLU_HIP(hipMalloc((void**)&h->pm_dst, (size_t)chunks * 2 * NB * sizeof(int)));
    RFLU_HIP(hipMalloc((void**)&h->pm_src, (size_t)chunks * 2 * NB * sizeof(int)));
    RFLU_HIP(hipMalloc(&h->linv, (size_t)chunks * NB * NB * sizeof(double)));
    RFLU_HIP(hipMemset(h->pm_cnt, 0, (size_t)chunks * sizeof(int)));
    h->pm_chunks = chunks;
    return RFLU_OK;
}

// ---- B <- L^-1 B by recursive splitting on 64-row boundaries (off-diagonal work = MFMA GEMM) -----------------------
// linv: inverses of L's 64x64 diagonal blocks (one 64x64 dense block per 64 rows), or nullptr.  With them, triangles of
// up to 256 rows are solved by ONE fused strip kernel instead of 7 dependent launches.
constexpr int64_t TRSM_FUSED_MAX = 256;

template <typename T>
static int trsm_rec(Handle* h, int64_t n, int64_t nrhs, const T* L, int64_t ldl, T* B, int64_t ldb, const T* linv)
{
    if (n <= 0 || nrhs <= 0) return RFLU_OK;
    if (linv && n <= TRSM_FUSED_MAX) return launch_trsm_fused<T>(h, n, nrhs, L, ldl, linv, B, ldb);
    if (n <= NB) return launch_trsm_base<T>(h, n, nrhs, L, ldl, B, ldb);
    const int64_t leaves = (n + NB - 1) / NB;
    const int64_t n1 = ((leaves + 1) / 2) * NB;
    RFLU_TRY(trsm_rec<T>(h, n1, nrhs, L, ldl, B, ldb, linv));
    RFLU_TRY(launch_gemm<T>(h, n - n1, nrhs, n1, L + n1 * ldl, ldl, B, ldb, B + n1 * ldb, ldb));
    return trsm_rec<T>(h, n - n1, nrhs, L + n1 * ldl + n1, ldl, B + n1 * ldb, ldb,
                       linv ? linv + (n1 / NB) * NB * NB : nullptr);
}

// stand-alone TRSM (C ABI building block): invert the diagonal blocks first, then the fused path
template <typename T>
static int trsm_public(Handle* h, int64_t n, int64_t nrhs, const T* L, int64_t ldl, T* B, int64_t ldb)
{
    if (n <= 0 || nrhs <= 0) return RFLU_OK;
    const size_t need = (size_t)((n + NB - 1) / NB) * NB * NB * sizeof(T);
    RFLU_TRY(ensure_buffer(&h->linv_tmp, &h->linv_tmp_bytes, need));
    h->trsv_area = nullptr;  // the cooperative solve's exchange area shares this buffer: have it wiped before its next use
    T* li = static_cast<T*>(h->linv_tmp);
    RFLU_TRY(launch_diag_inv<T>(h, n, L, ldl, li));
    return trsm_rec<T>(h, n, nrhs, L, ldl, B, ldb, li);
}

// ---- B <- U^-1 B (upper, non-unit) by recursive splitting on 64-row boundaries: bottom block, GEMM, top block ------
template <typename T>
static int triu_solve_rec(Handle* h, int64_t n, int64_t nrhs, const T* U, int64_t ldu, T* B, int64_t ldb)
{
    if (n <= 0 || nrhs <= 0) return RFLU_OK;
    if (n <= NB) return launch_triu_base<T>(h, n, nrhs, U, ldu, B, ldb);
    const int64_t leaves = (n + NB - 1) / NB;
    const int64_t n1 = ((leaves + 1) / 2) * NB;  // rows of the top block; the (possibly partial) rest is the bottom
    RFLU_TRY(triu_solve_rec<T>(h, n - n1, nrhs, U + n1 * ldu + n1, ldu, B + n1 * ldb, ldb));
    RFLU_TRY(launch_gemm<T>(h, n1, nrhs, n - n1, U + n1, ldu, B + n1 * ldb, ldb, B, ldb));
    return triu_solve_rec<T>(h, n1, nrhs, U, ldu, B, ldb);
}

// ldiv!(F::LU, B): B <- U^-1 L^-1 P B on row-major device data (F as left by getrf_rm; B is n x nrhs, row-major).
template <typename T>
static int getrs_rm(Handle* h, int64_t n, int64_t nrhs, const T* R, int64_t ld, const int64_t* ipiv, T* B, int64_t ldb)
{
    if (n <= 0 || nrhs <= 0) return RFLU_OK;
    RFLU_TRY(ensure_bookkeeping(h, n));
    if (ipiv) {  // rows of B follow the factorization's interchanges (NULL = NotIPIV: nothing to apply)
        RFLU_TRY(launch_perm_build(h, ipiv, 0, n, n));
        RFLU_TRY(launch_laswp<T>(h, B, ldb, 0, nrhs, 0, (n + NB - 1) / NB));
    }
    // few right-hand sides (<= 32: measured crossover): one cooperative launch per triangle and pass of 8 (trsv.hip)
    // instead of ~n/32 dependent launches; many:
    // the recursive splitting, whose GEMMs then carry the work.  RFLU_TRSV_MAX_RHS moves the crossover (0 = never).
    static const int64_t trsv_max = [] { const char* e = getenv("RFLU_TRSV_MAX_RHS"); return e ? atoll(e) : 32ll; }();
    if (nrhs <= trsv_max && n <= (int64_t)NB * 256 * 4) {
        RFLU_HIP(hipMemsetAsync(h->info_dev, 0, 2 * sizeof(int64_t), h->stream));
        RFLU_TRY(launch_trsv_coop<T>(h, n, nrhs, R, ld, B, ldb));
        RFLU_HIP(hipMemcpyAsync(h->info_pinned, h->info_dev, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
        RFLU_HIP(hipStreamSynchronize(h->stream));
        if (h->info_pinned[1] != 0) {
            set_error("cooperative solve kernel timed out waiting for a peer workgroup");
            return RFLU_ERR_TIMEOUT;
        }
        return RFLU_OK;
    }
    RFLU_TRY(trsm_public<T>(h, n, nrhs, R, ld, B, ldb));
    return triu_solve_rec<T>(h, n, nrhs, R, ld, B, ldb);
}

// column-major device entry: F (n x n, lda) and B (n x nrhs, ldb) as LinearAlgebra.LU / LAPACK getrs hold them
template <typename T>
static int getrs_cm_dev(Handle* h, int64_t n, int64_t nrhs, const T* F, int64_t lda, const int64_t* ipiv, T* B,
                        int64_t ldb)
{
    if (n < 0 || nrhs < 0 || lda < std::max<int64_t>(n, 1) || ldb < std::max<int64_t>(n, 1)) {
        set_error("getrs: bad arguments n=%lld nrhs=%lld lda=%lld ldb=%lld", (long long)n, (long long)nrhs,
                  (long long)lda, (long long)ldb);
        return RFLU_ERR_ARG;
    }
    if (n == 0 || nrhs == 0) return RFLU_OK;
    const int64_t ldr = round_up(n, 16), ldx = round_up(nrhs, 16);
    RFLU_TRY(ensure_buffer(&h->work, &h->work_bytes, (size_t)n * (size_t)ldr * sizeof(T)));
    RFLU_TRY(ensure_buffer(&h->rhs_work, &h->rhs_work_bytes, (size_t)n * (size_t)ldx * sizeof(T)));
    T* R = static_cast<T*>(h->work);
    T* X = static_cast<T*>(h->rhs_work);
    RFLU_TRY(launch_transpose<T>(h, n, n, F, lda, R, ldr));
    RFLU_TRY(launch_transpose<T>(h, n, nrhs, B, ldb, X, ldx));
    RFLU_TRY(getrs_rm<T>(h, n, nrhs, R, ldr, ipiv, X, ldx));
    RFLU_TRY(launch_transpose<T>(h, nrhs, n, X, ldx, B, ldb));
    RFLU_HIP(hipStreamSynchronize(h->stream));
    return RFLU_OK;
}

template <typename T>
static int getrs_host(Handle* h, int64_t n, int64_t nrhs, const T* F, int64_t lda, const int64_t* ipiv, T* B, int64_t ldb)
{
    if (n < 0 || nrhs < 0 || lda < std::max<int64_t>(n, 1) || ldb < std::max<int64_t>(n, 1) ||
        (n > 0 && nrhs > 0 && (F == nullptr || B == nullptr))) {
        set_error("getrs: bad arguments");
        return RFLU_ERR_ARG;
    }
    if (n == 0 || nrhs == 0) return RFLU_OK;
    RFLU_TRY(ensure_buffer(&h->hostA_dev, &h->hostA_bytes, (size_t)n * (size_t)n * sizeof(T)));
    RFLU_TRY(ensure_buffer(&h->hostB_dev, &h->hostB_bytes, (size_t)n * (size_t)nrhs * sizeof(T)));
    if ((size_t)n > h->ipiv_cap) {
        if (h->ipiv_dev) RFLU_HIP(hipFree(h->ipiv_dev));
        h->ipiv_dev = nullptr;
        h->ipiv_cap = 0;
        RFLU_HIP(hipMalloc((void**)&h->ipiv_dev, (size_t)n * sizeof(int64_t)));
        h->ipiv_cap = (size_t)n;
    }
    T* dF = static_cast<T*>(h->hostA_dev);
    T* dB = static_cast<T*>(h->hostB_dev);
    RFLU_HIP(hipMemcpy2DAsync(dF, (size_t)n * sizeof(T), F, (size_t)lda * sizeof(T), (size_t)n * sizeof(T), (size_t)n,
                              hipMemcpyHostToDevice, h->stream));
    RFLU_HIP(hipMemcpy2DAsync(dB, (size_t)n * sizeof(T), B, (size_t)ldb * sizeof(T), (size_t)n * sizeof(T), (size_t)nrhs,
                              hipMemcpyHostToDevice, h->stream));
    if (ipiv) RFLU_HIP(hipMemcpyAsync(h->ipiv_dev, ipiv, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
    RFLU_TRY(getrs_cm_dev<T>(h, n, nrhs, dF, n, ipiv ? h->ipiv_dev : nullptr, dB, n));
    RFLU_HIP(hipMemcpy2DAsync(B, (size_t)ldb * sizeof(T), dB, (size_t)n * sizeof(T), (size_t)n * sizeof(T), (size_t)nrhs,
                              hipMemcpyDeviceToHost, h->stream));
    RFLU_HIP(hipStreamSynchronize(h->stream));
    return RFLU_OK;
}

// error flags raised by the cooperative kernels (info_dev[1], copied to info_pinned[1] by the caller)
static int panel_flags_status(Handle* h)
{
    const int64_t f = h->info_pinned[1];
    if (f & 2) {
        set_error("a workgroup of the XCD-local panel kernel ran on an unexpected XCD; results discarded "
                  "(set RFLU_PANEL_LOCAL=0 to use the placement-independent kernel)");
        return RFLU_ERR_PLACEMENT;
    }
    if (f != 0) {
        set_error("cooperative panel kernel timed out waiting for a peer workgroup");
        return RFLU_ERR_TIMEOUT;
    }
    return RFLU_OK;
}

template <typename T>
struct Fact {
    Handle* h;
    T* R;
    int64_t ld, m, n;  // full matrix: m rows, n columns
    int64_t* ipiv;
    int pivot;
    int64_t sw_lo = 0, sw_hi = -1;  // column range that receives a leaf's interchanges right away ([0, n) by default)
    int64_t roff = 0;               // row of the diagonal minus its column (non-zero for a block column of a slab)
    hipEvent_t gate = nullptr;      // if set: wait for it after the next leaf's panel kernel, before its interchanges

    T* linv_at(int64_t row) const { return static_cast<T*>(h->linv) + (row / NB) * NB * NB; }

    // leaf: rows [c0+roff, m), columns [c0, c0+w): cooperative panel + the interchanges on every other column
    int leaf(int64_t c0, int64_t w)
    {
        const int64_t r0 = c0 + roff;
        RFLU_TRY(launch_panel<T>(h, R, ld, m, r0, c0, w, ipiv, pivot));
        const int64_t hi = sw_hi < 0 ? n : sw_hi;
        // one launch: the leaf's interchanges on the other columns + the inverse of its diagonal block (fused TRSMs)
        if (pivot) RFLU_TRY(launch_laswp2<T>(h, R, ld, sw_lo, c0 - sw_lo, c0 + w, hi - (c0 + w), r0 / NB, r0 / NB + 1, w,
                                             R + r0 * ld + c0, linv_at(r0)));
        else RFLU_TRY(launch_diag_inv<T>(h, w, R + r0 * ld + c0, ld, linv_at(r0)));
        return RFLU_OK;
    }

    // two full leaves in one cooperative launch (panel.hip: panel_pivot_pair_kernel) + one interchange launch: both
    // leaves' interchanges on the columns outside the pair, leaf B's on leaf A's columns, both diagonal inverses
    int leaf_pair(int64_t c0)
    {
        const int64_t r0 = c0 + roff;
        RFLU_TRY(launch_panel_pair<T>(h, R, ld, m, r0, c0, ipiv));
        const int64_t hi = sw_hi < 0 ? n : sw_hi;
        return launch_laswp3<T>(h, R, ld, sw_lo, c0 - sw_lo, c0 + 2 * NB, hi - (c0 + 2 * NB), c0, NB, r0 / NB, r0 / NB + 2,
                                NB, 2, R + r0 * ld + c0, linv_at(r0));
    }

    bool pair_ok(int64_t c0) const
    {
        // opt-in (RFLU_PAIR=1): correct and parity-tested, but at its current inter-leaf cost (~100 us: slot gather 25,
        // LDS solve 24, LDS-bandwidth-bound Schur update 40) it is 6 % slower than the three launches it replaces
        const char* e = getenv("RFLU_PAIR");
        const bool on = e != nullptr && e[0] == '1';
        const int64_t rows = m - (c0 + roff);
        return pivot && on && rows >= 2 * NB && (rows + PANEL_THREADS - 1) / PANEL_THREADS <= MAX_PANEL_WGS;
    }

    // reckernel! (src/lu.jl:189-263) on columns [c0, c1), rows [c0+roff, m)
    int rec(int64_t c0, int64_t c1)
    {
        const int64_t w = c1 - c0;
        if (w <= 0) return RFLU_OK;
        if (w <= NB) return leaf(c0, w);
        if (w == 2 * NB && pair_ok(c0)) return leaf_pair(c0);
        const int64_t leaves = (w + NB - 1) / NB;
        const int64_t n1 = ((leaves + 1) / 2) * NB;
        const int64_t cm = c0 + n1;
        RFLU_TRY(rec(c0, cm));
        T* A11 = R + (c0 + roff) * ld + c0;
        T* A12 = R + (c0 + roff) * ld + cm;
        T* A21 = R + (cm + roff) * ld + c0;
        T* A22 = R + (cm + roff) * ld + cm;
        RFLU_TRY(trsm_rec<T>(h, n1, c1 - cm, A11, ld, A12, ld, linv_at(c0 + roff)));              // src/lu.jl:235
        RFLU_TRY(launch_gemm<T>(h, m - (cm + roff), c1 - cm, n1, A21, ld, A12, ld, A22, ld));     // src/lu.jl:240
        return rec(cm, c1);
    }
};

static int get_event(Handle* h, size_t idx, hipEvent_t* ev)
{
    while (h->events.size() <= idx) {
        hipEvent_t e;
        RFLU_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        h->events.push_back(e);
    }
    *ev = h->events[idx];
    return RFLU_OK;
}

// An update stream leaves `reserve` CUs (a multiple of 32, 32..224) to the critical-path stream so that the cooperative
// panel kernel (one 512-thread workgroup per CU) finds all its workgroups a home at once.  Streams are created once per
// reservation and kept for the life of the handle.
int get_ustream(Handle* h, int reserve, hipStream_t* out)
{
    const int r = reserve / 32;
    if (reserve % 32 != 0 || r < 1 || r > 7) { set_error("CU reservation %d not in 32..224 step 32", reserve); return RFLU_ERR_ARG; }
    if (!h->ustreams[r]) {
        // CU mask bits are enumerated round-robin over the 8 XCDs (scripts/probes/cumask.hip): bits 0..31 are 4 CUs of
        // every XCD, and so on.  A mask that empties an XCD is ignored by the runtime, so whole 32-bit words are cleared.
        // The mask covers the CUs the device actually reports (num_cus / 32 words); callers only ask for a reservation
        // on a full 256-CU device (factor_lookahead).
        uint32_t mask[8];
        const int words = std::min(8, (h->num_cus + 31) / 32);
        for (int i = 0; i < 8; ++i) mask[i] = (i < r || i >= words) ? 0u : 0xffffffffu;
        if (words <= r || hipExtStreamCreateWithCUMask(&h->ustreams[r], (uint32_t)words, mask) != hipSuccess) {
            (void)hipGetLastError();
            RFLU_HIP(hipStreamCreateWithFlags(&h->ustreams[r], hipStreamNonBlocking));
        }
    }
    *out = h->ustreams[r];
    return RFLU_OK;
}

// ---- cost model of the lookahead schedule (microseconds; calibrated on MI355X) ----
static double model_panel_us(int64_t rows, int64_t W)
{
    const double G = double((rows + PANEL_THREADS - 1) / PANEL_THREADS);
    const double step = 2.6 + 0.025 * G;                      // one pivot step of the cooperative kernel
    return double(W) * step + double(W) / NB * 70.0           // + per-leaf interchanges / solves / launches
           + double(rows) * double(W) * double(W) / 30e6;     // + the recursion's own GEMMs (small K, ~30 TFLOP/s)
}
static double model_gemm_flops_per_us(int64_t K, int cus, size_t elem)
{
    const double tf = (K >= 2048 ? 64.0 : K >= 1024 ? 60.0 : K >= 512 ? 55.0 : 50.0) * (elem == 4 ? 1.6 : 1.0);
    return tf * 1e6 * double(cus) / 256.0;
}

// Right-looking over block columns of width W with one block column of lookahead (two streams).
//   P (h->stream, all CUs)      : panel_b -> restB_{b-1} -> evP[b] -> [wait evU1[b-1]] next_b (update of block column b+1)
//                                 -> panel_{b+1} ...
//   U (CU-masked update stream) : [wait evP[b]] left swaps_b -> rest_b.part1 (block column b+2) -> evU1[b] -> restA_b
// rest_b (the update of everything right of block column b+1) is split by columns: restA_b is sized by the cost model to
// take as long as P's next_b + panel_{b+1}, and runs next to them on the CUs the mask leaves it; what does not fit in
// that time (restB_b: the early, update-bound block columns and every tall panel) follows panel_{b+1} on P with the
// whole GPU.  The mask reserves ceil(panel workgroups / 32) * 32 CUs, chosen per block column.
// Every block column receives the same operations in the same order as in the recursion; only independent pieces
// overlap in time, so the factors are those of the one-stream path.
template <typename T>
static int factor_lookahead(Fact<T>& f, int64_t W)
{
    Handle* h = f.h;
    int min_reserve = 32;
    double split_scale = 1.0;
    const bool split_all = getenv("RFLU_SPLIT_ALL") != nullptr;
    double split_share = 0.5;   // tuning knob: how much of what is left after the modelled time stays on the update stream
    if (const char* e = getenv("RFLU_SPLIT_SHARE")) split_share = atof(e);
    int max_reserve = 64;       // taller panels (> 64 workgroups) take too many CUs from the update: one stream instead
    if (const char* e = getenv("RFLU_MAX_RESERVE")) max_reserve = atoi(e);
    if (const char* e = getenv("RFLU_RESERVE_CUS")) {  // tuning knob: least number of CUs kept away from the update stream
        const int v = atoi(e);
        if (v >= 32 && v <= 224 && v % 32 == 0) min_reserve = v;
    }
    if (const char* e = getenv("RFLU_SPLIT_SCALE")) {  // tuning knob: scales the modelled critical-path time (0 = no split)
        split_scale = atof(e);
    }
    hipStream_t P = h->stream;
    const int64_t m = f.m, n = f.n, ld = f.ld, mn = std::min(m, n);
    T* R = f.R;
    const int64_t nblk = (mn + W - 1) / W;
    hipEvent_t ev;

    auto update = [&](hipStream_t st, int64_t j0, int64_t jb, int64_t c0, int64_t c1) -> int {
        // apply block column [j0, j0+jb) to columns [c0, c1): interchanges, block-row solve, Schur update
        if (c1 <= c0) return RFLU_OK;
        hipStream_t saved = h->stream;
        h->stream = st;
        int rc = RFLU_OK;
        const int64_t je = j0 + jb;
        if (f.pivot) rc = launch_laswp<T>(h, R, ld, c0, c1 - c0, j0 / NB, (je + NB - 1) / NB);
        if (rc == RFLU_OK) rc = trsm_rec<T>(h, jb, c1 - c0, R + j0 * ld + j0, ld, R + j0 * ld + c0, ld, f.linv_at(j0));
        if (rc == RFLU_OK && m > je)
            rc = launch_gemm<T>(h, m - je, c1 - c0, jb, R + je * ld + j0, ld, R + j0 * ld + c0, ld, R + je * ld + c0, ld);
        h->stream = saved;
        return rc;
    };

    // events: 4b+1 = evP[b], 4b+2 = evU1[b], 4b+3 = evUend[b]
    hipStream_t Uprev = nullptr;      // update stream of the previous overlapped block column
    int64_t uend_prev = -1;           // its block index (evUend recorded), -1: none
    bool prev_overlapped = false;
    struct { bool valid = false; int64_t j0 = 0, jb = 0, c0 = 0; int64_t need_uend = -1; } pend;  // restB of block b-1

    auto flush_pending = [&]() -> int {
        if (!pend.valid) return RFLU_OK;
        if (pend.need_uend >= 0) {  // its columns were last written by restA of the block column before
            hipEvent_t e2;
            RFLU_TRY(get_event(h, 4 * pend.need_uend + 3, &e2));
            RFLU_HIP(hipStreamWaitEvent(P, e2, 0));
        }
        pend.valid = false;
        return update(P, pend.j0, pend.jb, pend.c0, n);
    };

    for (int64_t b = 0; b < nblk; ++b) {
        const int64_t j0 = b * W, jb = std::min(W, mn - j0), je = j0 + jb;
        // ---- panel b on P: Toledo recursion on the block column, interchanges confined to its own columns ----
        f.sw_lo = j0;
        f.sw_hi = je;
        RFLU_TRY(f.rec(j0, je));
        RFLU_TRY(flush_pending());                                   // restB_{b-1}: whole GPU, after the panel
        // the panel that will run next to this block column's update is panel b+1
        const int64_t rows_next = m - je;
        const int64_t g_next = (rows_next + PANEL_THREADS - 1) / PANEL_THREADS;
        const int reserve = std::max<int>(min_reserve, int((std::max<int64_t>(g_next, 1) + 31) / 32 * 32));
        if (reserve > std::min(max_reserve, 224)) {
            // the next panel needs (almost) the whole GPU: run this block column on one stream
            if (uend_prev >= 0) {
                RFLU_TRY(get_event(h, 4 * uend_prev + 3, &ev));
                RFLU_HIP(hipStreamWaitEvent(P, ev, 0));
                uend_prev = -1;
            }
            if (f.pivot && j0 > 0) RFLU_TRY(launch_laswp<T>(h, R, ld, 0, j0, j0 / NB, (je + NB - 1) / NB));
            RFLU_TRY(update(P, j0, jb, je, n));
            prev_overlapped = false;
            Uprev = nullptr;
            continue;
        }
        hipStream_t U;
        RFLU_TRY(get_ustream(h, reserve, &U));
        RFLU_TRY(get_event(h, 4 * b + 1, &ev));
        RFLU_HIP(hipEventRecord(ev, P));
        RFLU_HIP(hipStreamWaitEvent(U, ev, 0));
        if (Uprev && Uprev != U && uend_prev >= 0) {                 // a different mask: order the two update streams
            RFLU_TRY(get_event(h, 4 * uend_prev + 3, &ev));
            RFLU_HIP(hipStreamWaitEvent(U, ev, 0));
        }
        // ---- U: interchanges on the finished columns to the left ----
        if (f.pivot && j0 > 0) {
            hipStream_t saved = h->stream;
            h->stream = U;
            const int rc = launch_laswp<T>(h, R, ld, 0, j0, j0 / NB, (je + NB - 1) / NB);
            h->stream = saved;
            RFLU_TRY(rc);
        }
        if (je >= n) {
            RFLU_TRY(get_event(h, 4 * b + 3, &ev));
            RFLU_HIP(hipEventRecord(ev, U));
            uend_prev = b;
            Uprev = U;
            break;
        }
        const int64_t n1e = std::min(je + W, n);                                            // end of block column b+1
        const int64_t n2e = std::min(n1e + W, n);                                           // end of block column b+2
        // ---- P: next block column (needs rest_{b-1}.part1, which updated exactly these columns).  Handing all but its
        // first leaf to U (and gating P's second leaf on it) was measured slower: U's in-order queue is still busy with
        // rest_{b-1} in the early, update-bound block columns.
        if (b > 0 && prev_overlapped) {
            RFLU_TRY(get_event(h, 4 * (b - 1) + 2, &ev));
            RFLU_HIP(hipStreamWaitEvent(P, ev, 0));
        }
        RFLU_TRY(update(P, j0, jb, je, n1e));
        // ---- U: block column b+2 first (the next `next`), then as much of the rest as fits next to P's work ----
        RFLU_TRY(update(U, j0, jb, n1e, n2e));
        RFLU_TRY(get_event(h, 4 * b + 2, &ev));
        RFLU_HIP(hipEventRecord(ev, U));
        int64_t cA = n;                                                                      // restA = [n2e, cA)
        // With the minimal reservation the masked stream keeps 7/8 of the GPU and a split cannot win more than ~1 %
        // (measured: nothing); it pays for the tall panels, whose reservation takes a quarter to half of the CUs.
        if (split_scale > 0 && m > je && (reserve > 32 || split_all)) {
            const double rateU = model_gemm_flops_per_us(jb, 256 - reserve, sizeof(T));
            const double rateP = model_gemm_flops_per_us(jb, 256, sizeof(T));
            const double col_flops = 2.0 * double(m - je) * double(jb);                      // per trailing column
            const double tP = split_scale * ((je < mn ? model_panel_us(m - je, std::min(W, mn - je)) : 0.0)
                                             + col_flops * double(n1e - je) / rateP);
            const double colsA = tP * rateU / col_flops;                                     // columns U finishes in tP
            const int64_t left = n - n1e;
            // what is left after tP is shared by both streams; below one GEMM tile column it is not worth a launch
            int64_t a = int64_t(colsA) / 128 * 128;
            a = std::max<int64_t>(a, n2e - n1e);
            if (left - a >= 512)
                cA = n1e + a + int64_t(split_share * double(left - a) * double(256 - reserve) / 512.0) / 128 * 128;
        }
        RFLU_TRY(update(U, j0, jb, n2e, cA));
        RFLU_TRY(get_event(h, 4 * b + 3, &ev));
        RFLU_HIP(hipEventRecord(ev, U));
        if (cA < n) {
            pend.valid = true;
            pend.j0 = j0;
            pend.jb = jb;
            pend.c0 = cA;
            pend.need_uend = uend_prev;   // restA_{b-1} may have written columns right of cA
        }
        uend_prev = b;
        Uprev = U;
        prev_overlapped = true;
    }
    RFLU_TRY(flush_pending());
    f.sw_lo = 0;
    f.sw_hi = -1;
    f.gate = nullptr;
    // join: P continues only after U has drained
    if (uend_prev >= 0) {
        RFLU_TRY(get_event(h, 4 * uend_prev + 3, &ev));
        RFLU_HIP(hipStreamWaitEvent(P, ev, 0));
    }
    return RFLU_OK;
}

// Factor the row-major m x n matrix R in place (see rflu.h for `blocksize`).
template <typename T>
static int getrf_rm(Handle* h, int64_t m, int64_t n, T* R, int64_t ld, int64_t* ipiv, int pivot, int64_t blocksize,
                    int64_t* info)
{
    if (m < 0 || n < 0 || ld < std::max<int64_t>(n, 1) || (m > 0 && n > 0 && R == nullptr)) {
        set_error("getrf: bad arguments m=%lld n=%lld ld=%lld", (long long)m, (long long)n, (long long)ld);
        return RFLU_ERR_ARG;
    }
    if (pivot && ipiv == nullptr && std::min(m, n) > 0) {
        set_error("getrf: pivot != 0 needs an ipiv buffer");
        return RFLU_ERR_ARG;
    }
    *info = 0;
    const int64_t mn = std::min(m, n);
    h->last_path = RFLU_PATH_NONE;
    if (mn == 0) return RFLU_OK;
    RFLU_TRY(ensure_bookkeeping(h, m));
    RFLU_HIP(hipMemsetAsync(h->info_dev, 0, 2 * sizeof(int64_t), h->stream));
    if (!pivot && ipiv) RFLU_TRY(launch_iota_ipiv(h, ipiv, 0, mn));  // src/lu.jl:111-113

    Fact<T> f{h, R, ld, m, n, ipiv, pivot};
    bool fat_tail_done = false;
    if (blocksize == 0)  // measured on MI355X (bench.py --blocksize sweep): the knee moves right with the matrix size
        blocksize = mn < 1024 ? -1 : (mn <= 8192 ? 128 : (mn <= 12288 ? 256 : (mn <= 16384 ? 512 : (mn <= 24576 ? 1024 : 2048))));
    if (blocksize < 0 || blocksize >= mn) {
        h->last_path = RFLU_PATH_HIP_RECURSIVE;
        RFLU_TRY(f.rec(0, mn));
    } else if (!h->prof && h->num_cus == 256) {   // the CU reservation of the two-stream schedule is laid out for 8 x 32 CUs
        h->last_path = RFLU_PATH_HIP_LOOKAHEAD;
        RFLU_TRY(factor_lookahead<T>(f, round_up(blocksize, NB)));
        fat_tail_done = true;  // the block-column updates already reached the columns right of the square part
    } else {
        h->last_path = RFLU_PATH_HIP_BLOCKED;
        const int64_t bs = round_up(blocksize, NB);
        for (int64_t j = 0; j < mn; j += bs) {
            const int64_t jb = std::min(bs, mn - j);
            RFLU_TRY(f.rec(j, j + jb));
            const int64_t je = j + jb;
            if (je < mn) {  // trailing update of the remaining square part
                RFLU_TRY(trsm_rec<T>(h, jb, mn - je, R + j * ld + j, ld, R + j * ld + je, ld, f.linv_at(j)));
                RFLU_TRY(launch_gemm<T>(h, m - je, mn - je, jb, R + je * ld + j, ld, R + j * ld + je, ld,
                                        R + je * ld + je, ld));
            }
        }
    }
    if (m < n && !fat_tail_done)  // fat matrix: AR <- L^-1 AR (src/lu.jl:148-154; interchanges already applied there)
        RFLU_TRY(trsm_rec<T>(h, m, n - m, R, ld, R + m, ld, f.linv_at(0)));

    RFLU_HIP(hipMemcpyAsync(h->info_pinned, h->info_dev, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
    RFLU_HIP(hipStreamSynchronize(h->stream));
    RFLU_TRY(panel_flags_status(h));
    *info = h->info_pinned[0];
    return RFLU_OK;
}

// column-major device entry: R-layout workspace, transpose in, factor, transpose out
template <typename T>
static int getrf_cm_dev(Handle* h, int64_t m, int64_t n, T* A, int64_t lda, int64_t* ipiv, int pivot,
                        int64_t blocksize, int64_t* info)
{
    if (m < 0 || n < 0 || lda < std::max<int64_t>(m, 1) || info == nullptr) {
        set_error("getrf: bad arguments m=%lld n=%lld lda=%lld", (long long)m, (long long)n, (long long)lda);
        return RFLU_ERR_ARG;
    }
    *info = 0;
    if (m == 0 || n == 0) return RFLU_OK;
    const int64_t ldr = round_up(n, 16);
    RFLU_TRY(ensure_buffer(&h->work, &h->work_bytes, (size_t)m * (size_t)ldr * sizeof(T)));
    T* R = static_cast<T*>(h->work);
    RFLU_TRY(launch_transpose<T>(h, m, n, A, lda, R, ldr));
    RFLU_TRY(getrf_rm<T>(h, m, n, R, ldr, ipiv, pivot, blocksize, info));
    RFLU_TRY(launch_transpose<T>(h, n, m, R, ldr, A, lda));
    RFLU_HIP(hipStreamSynchronize(h->stream));
    return RFLU_OK;
}

// host entry: stage through device buffers owned by the handle
template <typename T>
static int getrf_host(Handle* h, int64_t m, int64_t n, T* A, int64_t lda, int64_t* ipiv, int pivot, int64_t blocksize,
                      int64_t* info)
{
    if (m < 0 || n < 0 || lda < std::max<int64_t>(m, 1) || info == nullptr || (m > 0 && n > 0 && A == nullptr)) {
        set_error("getrf: bad arguments m=%lld n=%lld lda=%lld", (long long)m, (long long)n, (long long)lda);
        return RFLU_ERR_ARG;
    }
    *info = 0;
    const int64_t mn = std::min(m, n);
    if (mn == 0) return RFLU_OK;
    RFLU_TRY(ensure_buffer(&h->hostA_dev, &h->hostA_bytes, (size_t)m * (size_t)n * sizeof(T)));
    if ((size_t)mn > h->ipiv_cap) {
        if (h->ipiv_dev) RFLU_HIP(hipFree(h->ipiv_dev));
        h->ipiv_dev = nullptr;
        h->ipiv_cap = 0;
        RFLU_HIP(hipMalloc((void**)&h->ipiv_dev, (size_t)mn * sizeof(int64_t)));
        h->ipiv_cap = (size_t)mn;
    }
    T* dA = static_cast<T*>(h->hostA_dev);
    RFLU_HIP(hipMemcpy2DAsync(dA, (size_t)m * sizeof(T), A, (size_t)lda * sizeof(T), (size_t)m * sizeof(T), (size_t)n,
                              hipMemcpyHostToDevice, h->stream));
    const bool want_ipiv = (ipiv != nullptr);
    RFLU_TRY(getrf_cm_dev<T>(h, m, n, dA, m, (pivot || want_ipiv) ? h->ipiv_dev : nullptr, pivot, blocksize, info));
    RFLU_HIP(hipMemcpy2DAsync(A, (size_t)lda * sizeof(T), dA, (size_t)m * sizeof(T), (size_t)m * sizeof(T), (size_t)n,
                              hipMemcpyDeviceToHost, h->stream));
    if (want_ipiv)
        RFLU_HIP(hipMemcpyAsync(ipiv, h->ipiv_dev, (size_t)mn * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
    RFLU_HIP(hipStreamSynchronize(h->stream));
    return RFLU_OK;
}

}  // namespace rflu

using namespace rflu;

static Handle* H(rflu_handle_t h) { return reinterpret_cast<Handle*>(h); }
namespace rflu { int get_ustream(Handle* h, int reserve, hipStream_t* out); }

// Every API entry runs on the handle's device and leaves the caller's current device as it found it (a framework with
// tensors on several GPUs must not find its current device changed by a library call).
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int dev)
    {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) {
            err = hipSetDevice(dev);
            switched = (err == hipSuccess);
        }
    }
    ~DeviceGuard()
    {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

#define CHECK_HANDLE(h)                                  \
    if ((h) == nullptr) {                                \
        set_error("null handle");                        \
        return RFLU_ERR_ARG;                             \
    }                                                    \
    DeviceGuard device_guard__(H(h)->device);            \
    RFLU_HIP(device_guard__.err)

extern "C" {

int rflu_version(void) { return 100; }

const char* rflu_last_error(void) { return g_err; }

int rflu_create(rflu_handle_t* handle, int device)
{
    if (handle == nullptr) { set_error("null handle pointer"); return RFLU_ERR_ARG; }
    *handle = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("no HIP device visible (this library has no CPU fallback)");
        return RFLU_ERR_NODEVICE;
    }
    if (device < 0 || device >= ndev) { set_error("device %d out of range (0..%d)", device, ndev - 1); return RFLU_ERR_ARG; }
    DeviceGuard device_guard__(device);
    RFLU_HIP(device_guard__.err);
    hipDeviceProp_t prop;
    RFLU_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; librflu is built for gfx950 (MI355X) only", device, prop.gcnArchName);
        return RFLU_ERR_NODEVICE;
    }
    Handle* h = new (std::nothrow) Handle();
    if (!h) { set_error("out of host memory"); return RFLU_ERR_ARG; }
    h->device = device;
    h->num_cus = prop.multiProcessorCount;
    // a BLOCKING stream: it orders itself against the legacy default stream, so buffers produced by a framework on
    // stream 0 (PyTorch's default) need no extra synchronisation before/after a call
    {
        // highest priority: in the lookahead driver this stream carries the critical path (panels + next block column)
        // and competes for CUs with the bulk trailing update on the second stream
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (getenv("RFLU_NO_PRIORITY")) hi = 0;
        RFLU_HIP(hipStreamCreateWithPriority(&h->own_stream, hipStreamDefault, hi));
    }
    h->stream = h->own_stream;
    RFLU_HIP(hipMalloc((void**)&h->info_dev, 2 * sizeof(int64_t)));
    RFLU_HIP(hipHostMalloc((void**)&h->info_pinned, 2 * sizeof(int64_t)));
    h->pscratch_bytes = panel_scratch_bytes();
    RFLU_HIP(hipMalloc((void**)&h->pscratch, h->pscratch_bytes));
    RFLU_HIP(hipMemset(h->pscratch, 0, h->pscratch_bytes));
    RFLU_HIP(hipEventCreate(&h->ev0));
    RFLU_HIP(hipEventCreate(&h->ev1));
    // the cooperative kernels spin on peer workgroups: all of a launch's workgroups must be resident at once.  Ask the
    // runtime how many fit (one 512/576-thread workgroup per CU with these register counts) instead of assuming it.
    h->panel_max_wgs = panel_resident_limit(h->num_cus);
    if (h->panel_max_wgs <= 0) { set_error("occupancy query for the cooperative panel kernels failed"); delete h; return RFLU_ERR_HIP; }
    if (const char* e = getenv("RFLU_COOP_LAUNCH")) h->coop_launch = atoi(e) != 0;
    if (const char* e = getenv("RFLU_PANEL_LOCAL")) h->panel_local = atoi(e);
    if (h->panel_local == 1) h->panel_local_maxg = 32;   // one XCD has 32 CUs
    if (const char* e = getenv("RFLU_PANEL_LOCAL_MAXG")) h->panel_local_maxg = atoi(e);
    *handle = reinterpret_cast<rflu_handle_t>(h);
    return RFLU_OK;
}

int rflu_destroy(rflu_handle_t handle)
{
    if (!handle) return RFLU_OK;
    Handle* h = H(handle);
    DeviceGuard device_guard__(h->device);
    (void)hipStreamSynchronize(h->stream);
    if (h->work) (void)hipFree(h->work);
    if (h->ipiv_dev) (void)hipFree(h->ipiv_dev);
    if (h->hostA_dev) (void)hipFree(h->hostA_dev);
    if (h->rhs_work) (void)hipFree(h->rhs_work);
    if (h->hostB_dev) (void)hipFree(h->hostB_dev);
    if (h->pm_cnt) (void)hipFree(h->pm_cnt);
    if (h->pm_dst) (void)hipFree(h->pm_dst);
    if (h->pm_src) (void)hipFree(h->pm_src);
    if (h->linv) (void)hipFree(h->linv);
    if (h->linv_tmp) (void)hipFree(h->linv_tmp);
    if (h->pscratch) (void)hipFree(h->pscratch);
    if (h->info_dev) (void)hipFree(h->info_dev);
    if (h->info_pinned) (void)hipHostFree(h->info_pinned);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    for (hipStream_t us : h->ustreams)
        if (us) (void)hipStreamDestroy(us);
    for (hipEvent_t e : h->events) (void)hipEventDestroy(e);
    for (auto& r : h->async_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (hipEvent_t e : h->async_pool) (void)hipEventDestroy(e);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
    return RFLU_OK;
}

int rflu_set_stream(rflu_handle_t handle, void* hip_stream)
{
    CHECK_HANDLE(handle);
    H(handle)->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : H(handle)->own_stream;
    return RFLU_OK;
}

int rflu_synchronize(rflu_handle_t handle)
{
    CHECK_HANDLE(handle);
    RFLU_HIP(hipStreamSynchronize(H(handle)->stream));
    return RFLU_OK;
}

int rflu_last_path(rflu_handle_t handle) { return handle ? H(handle)->last_path : RFLU_PATH_NONE; }

int rflu_update_stream(rflu_handle_t handle, void** hip_stream_out)
{
    CHECK_HANDLE(handle);
    if (hip_stream_out == nullptr) { set_error("null output pointer"); return RFLU_ERR_ARG; }
    hipStream_t us;
    RFLU_TRY(get_ustream(H(handle), 32, &us));
    *hip_stream_out = reinterpret_cast<void*>(us);
    return RFLU_OK;
}

#define DEFINE_TYPED(SFX, T)                                                                                          \
    int rflu_getrf_##SFX(rflu_handle_t handle, int64_t m, int64_t n, T* A, int64_t lda, int64_t* ipiv, int pivot,     \
                         int64_t blocksize, int64_t* info)                                                            \
    {                                                                                                                 \
        CHECK_HANDLE(handle);                                                                                         \
        return getrf_host<T>(H(handle), m, n, A, lda, ipiv, pivot, blocksize, info);                                  \
    }                                                                                                                 \
    int rflu_getrf_##SFX##_dev(rflu_handle_t handle, int64_t m, int64_t n, T* A, int64_t lda, int64_t* ipiv,          \
                               int pivot, int64_t blocksize, int64_t* info)                                           \
    {                                                                                                                 \
        CHECK_HANDLE(handle);                                                                                         \
        return getrf_cm_dev<T>(H(handle), m, n, A, lda, ipiv, pivot, blocksize, info);                                \
    }                                                                                                                 \
    int rflu_getrf_rm_##SFX##_dev(rflu_handle_t handle, int64_t m, int64_t n, T* R, int64_t ld, int64_t* ipiv,        \
                                  int pivot, int64_t blocksize, int64_t* info)                                        \
    {                                                                                                                 \
        CHECK_HANDLE(handle);                                                                                         \
        if (info == nullptr) { set_error("null info"); return RFLU_ERR_ARG; }                                         \
        return getrf_rm<T>(H(handle), m, n, R, ld, ipiv, pivot, blocksize, info);                                     \
    }                                                                                                                 \
    int rflu_panel_rm_##SFX##_dev(rflu_handle_t handle, int64_t m, int64_t r0, int64_t c0, int64_t w, T* R,           \
                                  int64_t ld, int64_t* ipiv, int pivot, int64_t* info)                                \
    {                                                                                                                 \
        CHECK_HANDLE(handle);                                                                                         \
        Handle* h = H(handle);                                                                                        \
        if (info == nullptr || w < 0 || r0 < 0 || c0 < 0 || m < r0 + w) { set_error("panel: bad arguments"); return RFLU_ERR_ARG; } \
        RFLU_TRY(ensure_bookkeeping(h, m));                                                                           \
        RFLU_HIP(hipMemsetAsync(h->info_dev, 0, 2 * sizeof(int64_t), h->stream));                                     \
        if (!pivot && ipiv) RFLU_TRY(launch_iota_ipiv(h, ipiv, r0, w));                                               \
        /* wide panels are factored by the same Toledo recursion as the single-GPU path, restricted to the columns   \
           [c0, c0+w) of the slab, diagonal at (r0, c0), interchanges confined to those columns */                   \
        {                                                                                                             \
            Fact<T> f{h, R, ld, m, c0 + w, ipiv, pivot};                                                              \
            f.sw_lo = c0; f.sw_hi = c0 + w; f.roff = r0 - c0;                                                         \
            RFLU_TRY(f.rec(c0, c0 + w));                                                                              \
        }                                                                                                             \
        RFLU_HIP(hipMemcpyAsync(h->info_pinned, h->info_dev, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream)); \
        RFLU_HIP(hipStreamSynchronize(h->stream));                                                                    \
        RFLU_TRY(panel_flags_status(h));                                                                              \
        *info = h->info_pinned[0];                                                                                    \
        return RFLU_OK;                                                                                               \
    }                                                                                                                 \
    int rflu_laswp_rm_##SFX##_dev(rflu_handle_t handle, T* R, int64_t ld, int64_t m, int64_t c0, int64_t ncols,       \
                                  const int64_t* ipiv, int64_t k0, int64_t k1)                                        \
    {                                                                                                                 \
        CHECK_HANDLE(handle);                                                                                         \
        if (k0 % NB != 0 || k1 < k0) { set_error("laswp: k0 must be a multiple of 64"); return RFLU_ERR_ARG; }        \
        RFLU_TRY(ensure_bookkeeping(H(handle), std::max(m, k1)));                                                     \
        RFLU_TRY(launch_perm_build(H(handle), ipiv, k0, k1, m));                                                      \
        return launch_laswp<T>(H(handle), R, ld, c0, ncols, k0 / NB, (k1 + NB - 1) / NB);                             \
    }                                                                                                                 \
    int rflu_trsm_rm_##SFX##_dev(rflu_handle_t handle, int64_t n, int64_t nrhs, const T* L, int64_t ldl, T* B,        \
                                 int64_t ldb)                                                                         \
    {                                                                                                                 \
        CHECK_HANDLE(handle);                                                                                         \
        return trsm_public<T>(H(handle), n, nrhs, L, ldl, B, ldb);                                                       \
    }                                                                                                                 \
    int rflu_gemm_rm_##SFX##_dev(rflu_handle_t handle, int64_t M, int64_t N, int64_t K, const T* A, int64_t lda,      \
                                 const T* B, int64_t ldb, T* C, int64_t ldc)                                          \
    {                                                                                                                 \
        CHECK_HANDLE(handle);                                                                                         \
        return launch_gemm<T>(H(handle), M, N, K, A, lda, B, ldb, C, ldc);                                            \
    }                                                                                                                 \
    int rflu_cm_to_rm_##SFX##_dev(rflu_handle_t handle, int64_t m, int64_t n, const T* A, int64_t lda, T* R,          \
                                  int64_t ldr)                                                                        \
    {                                                                                                                 \
        CHECK_HANDLE(handle);                                                                                         \
        return launch_transpose<T>(H(handle), m, n, A, lda, R, ldr);                                                  \
    }                                                                                                                 \
    int rflu_rm_to_cm_##SFX##_dev(rflu_handle_t handle, int64_t m, int64_t n, const T* R, int64_t ldr, T* A,          \
                                  int64_t lda)                                                                        \
    {                                                                                                                 \
        CHECK_HANDLE(handle);                                                                                         \
        return launch_transpose<T>(H(handle), n, m, R, ldr, A, lda);                                                  \
    }                                                                                                                 \
    int rflu_getrs_##SFX(rflu_handle_t handle, int64_t n, int64_t nrhs, const T* F, int64_t lda, const int64_t* ipiv,  \
                         T* B, int64_t ldb)                                                                           \
    {                                                                                                                 \
        CHECK_HANDLE(handle);                                                                                         \
        return getrs_host<T>(H(handle), n, nrhs, F, lda, ipiv, B, ldb);                                               \
    }                                                                                                                 \
    int rflu_getrs_##SFX##_dev(rflu_handle_t handle, int64_t n, int64_t nrhs, const T* F, int64_t lda,                \
                               const int64_t* ipiv, T* B, int64_t ldb)                                                \
    {                                                                                                                 \
        CHECK_HANDLE(handle);                                                                                         \
        return getrs_cm_dev<T>(H(handle), n, nrhs, F, lda, ipiv, B, ldb);                                             \
    }                                                                                                                 \
    int rflu_getrs_rm_##SFX##_dev(rflu_handle_t handle, int64_t n, int64_t nrhs, const T* R, int64_t ld,              \
                                  const int64_t* ipiv, T* B, int64_t ldb)                                             \
    {                                                                                                                 \
        CHECK_HANDLE(handle);                                                                                         \
        RFLU_TRY(getrs_rm<T>(H(handle), n, nrhs, R, ld, ipiv, B, ldb));                                               \
        RFLU_HIP(hipStreamSynchronize(H(handle)->stream));                                                            \
        return RFLU_OK;                                                                                               \
    }                                                                                                                 \
    int rflu_fill_uniform_##SFX##_dev(rflu_handle_t handle, T* A, int64_t m, int64_t n, int64_t ld, int row_major,    \
                                      uint64_t seed, int64_t M_global, int64_t i0, int64_t j0, double diag_add)       \
    {                                                                                                                 \
        CHECK_HANDLE(handle);                                                                                         \
        return launch_fill_uniform<T>(H(handle), A, m, n, ld, row_major, seed, M_global, i0, j0, diag_add);           \
    }

DEFINE_TYPED(f64, double)
DEFINE_TYPED(f32, float)

/* experiment hook (not in rflu.h): a stream restricted to an arbitrary CU mask (8 x 32 bits); the caller owns it */
int rflu_debug_masked_stream(rflu_handle_t handle, const unsigned* mask8, void** stream_out)
{
    CHECK_HANDLE(handle);
    hipStream_t st = nullptr;
    RFLU_HIP(hipExtStreamCreateWithCUMask(&st, 8, mask8));
    *stream_out = reinterpret_cast<void*>(st);
    return RFLU_OK;
}

/* experiment hook (not in rflu.h): copy the RFLU_PANEL_TRACE clock stamps of the last panel launch to the host */
int rflu_debug_panel_trace(rflu_handle_t handle, long long* out512)
{
    CHECK_HANDLE(handle);
    Handle* h = H(handle);
    RFLU_HIP(hipStreamSynchronize(h->stream));
    RFLU_HIP(hipMemcpy(out512, (char*)h->pscratch + panel_trace_offset_bytes(), (8 * NB + 16) * sizeof(long long), hipMemcpyDeviceToHost));
    return RFLU_OK;
}

/* experiment hook (not in rflu.h): the all-workgroup wall-clock stamps of a -DRFLU_PANEL_TRACE_ALL build (0 words otherwise) */
int rflu_debug_panel_trace_all(rflu_handle_t handle, long long* out, long long max_words)
{
    CHECK_HANDLE(handle);
    Handle* h = H(handle);
    RFLU_HIP(hipStreamSynchronize(h->stream));
    const size_t nw = std::min<size_t>(panel_trace_all_words(), (size_t)std::max<long long>(max_words, 0));
    if (nw) RFLU_HIP(hipMemcpy(out, (char*)h->pscratch + panel_trace_all_offset_bytes(), nw * sizeof(long long), hipMemcpyDeviceToHost));
    return (int)nw;
}

int rflu_profile_enable(rflu_handle_t handle, int enable)
{
    CHECK_HANDLE(handle);
    Handle* h = H(handle);
    h->prof = enable == 1;
    h->prof_async = enable == 2;
    for (int k = 0; k < RFLU_K_COUNT; ++k) h->slots[k] = ProfSlot();
    for (auto& r : h->async_recs) { h->async_pool.push_back(r.a); h->async_pool.push_back(r.b); }
    h->async_recs.clear();
    return RFLU_OK;
}

// in-schedule mode: fold the pending event pairs into the per-class timers (waits for the recorded work)
static int profile_resolve(Handle* h)
{
    for (auto& r : h->async_recs) {
        RFLU_HIP(hipEventSynchronize(r.b));
        float ms = 0.f;
        RFLU_HIP(hipEventElapsedTime(&ms, r.a, r.b));
        h->slots[r.k].ms += ms;
        h->slots[r.k].launches += 1;
        h->slots[r.k].work += r.work;
        h->slots[r.k].bytes += r.bytes;
        h->async_pool.push_back(r.a);
        h->async_pool.push_back(r.b);
    }
    h->async_recs.clear();
    return RFLU_OK;
}

int rflu_profile_get(rflu_handle_t handle, int kclass, double* ms, int64_t* launches, double* work)
{
    CHECK_HANDLE(handle);
    if (kclass < 0 || kclass >= RFLU_K_COUNT) { set_error("bad kernel class %d", kclass); return RFLU_ERR_ARG; }
    RFLU_TRY(profile_resolve(H(handle)));
    const ProfSlot& s = H(handle)->slots[kclass];
    if (ms) *ms = s.ms;
    if (launches) *launches = s.launches;
    if (work) *work = s.work;
    return RFLU_OK;
}

int rflu_profile_get_bytes(rflu_handle_t handle, int kclass, double* bytes)
{
    CHECK_HANDLE(handle);
    if (kclass < 0 || kclass >= RFLU_K_COUNT || bytes == nullptr) { set_error("bad kernel class %d", kclass); return RFLU_ERR_ARG; }
    RFLU_TRY(profile_resolve(H(handle)));
    *bytes = H(handle)->slots[kclass].bytes;
    return RFLU_OK;
}

}  // extern "C"
