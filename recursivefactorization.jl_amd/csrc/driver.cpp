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
#include "engine.hpp"
#include <atomic>
#include <chrono>
#include <thread>

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

// ---- the environment, read in ONE place (Tune::load_env / load_handle_env: at rflu_create and on rflu_reload_tuning) -----------
static const char* env_str(const char* name) { return getenv(name); }
static void env_get(const char* name, int& v) { if (const char* e = env_str(name)) v = atoi(e); }
static void env_get(const char* name, int64_t& v) { if (const char* e = env_str(name)) v = atoll(e); }
static void env_get(const char* name, double& v) { if (const char* e = env_str(name)) v = atof(e); }
static void env_flag(const char* name, int& v) { if (env_str(name)) v = 1; }   // set = on, whatever the value

void Tune::load_env()
{
    *this = Tune();
    // ---- the shipped library's surface: schedule selection, the knobs the tests and the measurement scripts
    // switch, tracing.  Everything below the #ifdef is tuning that was measured once and settled: read by the
    // experiments build only (RFLU_EXPERIMENTS=1 at BUILD time -> librflu_exp.so), compiled-in defaults otherwise.
    env_get("RFLU_PANEL_LOCAL_ROWS", panel_local_rows);
    env_get("RFLU_GEMM_MASKED", gemm_masked);
    env_get("RFLU_TRSV_MAX_RHS", trsv_max_rhs);
    env_get("RFLU_TRSM_CHAIN_MAX_RHS", trsm_chain_max_rhs);
    env_get("RFLU_QUEUE_CHECK", queue_check);
    env_flag("RFLU_QUEUE_TRACE", queue_trace);
    env_flag("RFLU_SPLIT_ALL", split_all);
    env_get("RFLU_SPLIT_SCALE", split_scale);
    env_get("RFLU_LEAFWISE", leafwise);
    env_get("RFLU_LEAFWISE_ROWS", leafwise_rows);
    env_flag("RFLU_GATE_TRACE", gate_trace);
    if (const char* e = env_str("RFLU_SCHEDULE")) schedule_events = strcmp(e, "events") == 0;
    // rocprofv3 --pmc exports this into the profiled process and runs one kernel at a time: a device-side gate would only ever
    // see its timeout, so a counter-collection run takes the event schedule by itself (RFLU_SCHEDULE=gates overrides)
    else if (env_str("ROCPROF_COUNTER_COLLECTION")) schedule_events = 1;
    env_get("RFLU_LEAF_FUSE", leaf_fuse);
    env_get("RFLU_HOST_EARLY_OUT", host_early_out);
    env_flag("RFLU_HOST_TRACE", host_trace);
    env_get("RFLU_HOST_THREADS", host_threads);
    env_get("RFLU_MGPU_BIG_RESERVE", mgpu_big_reserve);
    env_get("RFLU_DEBUG_GHOST_LEAF", debug_ghost_leaf);
    env_get("RFLU_ENGINE", engine);
    env_get("RFLU_ENGINE_ROWS", engine_rows);
    env_get("RFLU_ENGINE_HOST", engine_host);
    env_get("RFLU_ENGINE_REPLAY", engine_replay);
    env_get("RFLU_ENGINE_RETIRE", engine_retire);
    env_get("RFLU_ENGINE_AHEAD", engine_ahead);
#ifdef RFLU_EXPERIMENTS
    env_get("RFLU_PANEL_PW", panel_pw);
    env_get("RFLU_PANEL_MAXG", panel_maxg);
    env_get("RFLU_PANEL_RPW", panel_rpw);
    if (panel_rpw != 64 && panel_rpw != 128 && panel_rpw != 256 && panel_rpw != 384 && panel_rpw != 512) panel_rpw = 0;   // the kernels that exist
    env_get("RFLU_PANEL_SPARE", panel_spare);
    env_get("RFLU_PANEL_SPARE_MIN", panel_spare_min);
    env_get("RFLU_PANEL_BALLAST", panel_ballast);
    env_get("RFLU_PANEL_LOCAL_MIN", panel_local_min);
    env_get("RFLU_PANEL_LOCAL_PW8_ROWS", panel_local_pw8_rows);
    env_get("RFLU_POLL_DELAY", poll_delay);
    env_get("RFLU_POLL_ADAPT", poll_adapt);
    env_get("RFLU_LASWP_LPR", laswp_lpr);
    env_get("RFLU_GEMM_FLAGS", gemm_flags);
    env_get("RFLU_SKINNY_MAXK", skinny_max_k);
    env_get("RFLU_SKINNY_WIDE", skinny_wide);
    env_get("RFLU_GEMM_CFIRST_BELOW", gemm_cfirst_below);
    env_get("RFLU_LD_PAD", ld_pad);
    ld_pad = (ld_pad / 16) * 16;
    env_get("RFLU_TRSM_CHAIN_SPLIT", trsm_chain_split);
    env_get("RFLU_TRSM_CHAIN_CACHED", trsm_chain_cached);
    env_get("RFLU_SPLIT_SHARE", split_share);
    env_get("RFLU_MAX_RESERVE", max_reserve);
    env_get("RFLU_WIDE_NARROW", wide_narrow);
    env_get("RFLU_NARROW_COLS", narrow_cols);
    env_get("RFLU_RESERVE_CUS", min_reserve);
    env_get("RFLU_CONFINE_ROWS", confine_rows);
    env_get("RFLU_MERGE_ROWS", merge_rows);
    env_get("RFLU_SWAP_SU", swap_su);
    env_get("RFLU_SWAP_LATE", swap_late);
    env_get("RFLU_SWAP_ROWS", swap_rows);
    env_get("RFLU_GATE_FOLD", gate_fold);
    env_flag("RFLU_TIME_ENQUEUE", time_enqueue);
    env_get("RFLU_TAIL_OVERLAP", tail_overlap);
    env_get("RFLU_MGPU_TALL_ROWS", mgpu_tall_rows);
    env_get("RFLU_MGPU_SYNC", mgpu_sync);
    env_get("RFLU_ENGINE_POLICY", engine_policy);
    env_get("RFLU_ENGINE_WGS", engine_wgs);
    env_get("RFLU_ENGINE_WC", engine_wc);
    env_get("RFLU_ENGINE_WRITE_THROUGH", engine_write_through);
    env_get("RFLU_ENGINE_LEAF_XCDS", engine_leaf_xcds);
    env_get("RFLU_ENGINE_LEAF_WGS", engine_leaf_wgs);
    env_get("RFLU_ENGINE_HOST_LAG", engine_host_lag);
    env_get("RFLU_ENGINE_SOLVE_RL", engine_solve_rl);
#endif
}

// the handle's own switches (kernel routing) + its Tune
static void load_handle_env(Handle* h)
{
    h->tune.load_env();
    h->coop_launch = false;
    h->panel_local = 2;
    h->panel_single = 1;
    h->panel_blocked = 0;
    h->panel_local_maxg = 64;
    int v = 0;
    env_get("RFLU_COOP_LAUNCH", v);
    h->coop_launch = v != 0;
    env_get("RFLU_PANEL_LOCAL", h->panel_local);
    env_get("RFLU_PANEL_SINGLE", h->panel_single);
    env_get("RFLU_PANEL_BLOCKED", h->panel_blocked);
    if (h->panel_local == 1) h->panel_local_maxg = 32;   // one XCD has 32 CUs
#ifdef RFLU_EXPERIMENTS
    env_get("RFLU_PANEL_LOCAL_MAXG", h->panel_local_maxg);
#endif
}

// Leading dimension of the row-major workspace for n columns: a multiple of 16 elements (rows start on 128-byte lines).
// RFLU_LD_PAD=<elements> adds a padding when that is a multiple of 512 elements (a power-of-two row pitch): measured on MI355X
// (round 3, N=16384: 82.7 ms / laswp 3.11 TB/s without, 82.3-82.9 ms / 3.0-3.17 TB/s with 16..272 elements) it changes nothing --
// the HBM address hash already spreads equal columns of consecutive rows over the channels -- so the default is none.
static int64_t workspace_ld(const Handle* h, int64_t n)
{
    const int64_t pad = h->tune.ld_pad;
    int64_t ld = round_up(std::max<int64_t>(n, 1), 16);
    if (pad > 0 && ld % 512 == 0) ld += pad;
    return ld;
}

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
    const int64_t trsv_max = h->tune.trsv_max_rhs;
    // a block of right-hand sides (33 .. trsm_chain_max_rhs): the same cooperative chain in passes of 64 columns on the MFMA units
    // (trsv.hip: trsm_chain_kernel; n = 16384, 64 right-hand sides: 28.8 ms with the recursive splitting below)
    const bool wide = nrhs > trsv_max && nrhs <= h->tune.trsm_chain_max_rhs;
    if ((nrhs <= trsv_max || wide) && n <= (int64_t)NB * 256 * 4) {
        RFLU_HIP(hipMemsetAsync(h->info_dev, 0, 2 * sizeof(int64_t), h->stream));
        RFLU_TRY(launch_trsv_coop<T>(h, n, nrhs, R, ld, B, ldb, wide));
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
    const int64_t ldr = workspace_ld(h, n), ldx = round_up(nrhs, 16);
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
    if (f != 0 && h->eng_state && env_str("RFLU_ENGINE_DUMP")) {   // debugging: where the engine and the chain stood when somebody gave up
        EngState es;
        unsigned long long gate[3] = {0, 0, 0};
        (void)hipMemcpy(&es, h->eng_state, offsetof(EngState, cb) + 16 * sizeof(EngCB), hipMemcpyDeviceToHost);
        for (int i = 0; i < 3; ++i) (void)hipMemcpy(&gate[i], h->gate_ptr[i], 8, hipMemcpyDeviceToHost);
        fprintf(stderr, "[rflu] engine dump: flags 0x%llx gates %llu %llu %llu arrived %llu remaining %llu abort %llu epoch %llu\n", (unsigned long long)f, gate[0], gate[1], gate[2],
                es.arrived, es.remaining, es.abort, es.epoch);
        for (int c = 0; c < 16; ++c)
            fprintf(stderr, "   cb %2d: claim %llx done %llx lclaim %llx ldone %llu prog %llu leftdone %llx lprog %llu bigdone %llu\n", c, es.cb[c].claim, es.cb[c].done, es.cb[c].lclaim,
                    es.cb[c].ldone, es.cb[c].prog, es.cb[c].leftdone, es.cb[c].lprog, es.cb[c].bigdone);
    }
    if (f != 0) {
        set_error("cooperative panel kernel timed out waiting for a peer workgroup (flags 0x%llx: 1 = a leaf / gate, 16 = the engine idle, 32 = a wait for the engine, 64 = the XCD-local leaf)", (unsigned long long)f);
        return RFLU_ERR_TIMEOUT;
    }
    return RFLU_OK;
}

// C-ABI GEMM.  RFLU_GEMM_MASKED=<reserve> (measurement only): run it on the CU-masked update stream of the lookahead
// schedule and wait for it, so that scripts/microbench_gemm_k.py can time the kernel on 256 - reserve CUs.
int get_ustream(Handle* h, int reserve, hipStream_t* out);
template <typename T>
static int gemm_public(Handle* h, int64_t M, int64_t N, int64_t K, const T* A, int64_t lda, const T* B, int64_t ldb, T* C,
                       int64_t ldc)
{
    if (h->tune.gemm_masked < 0) return launch_gemm<T>(h, M, N, K, A, lda, B, ldb, C, ldc);
    hipStream_t U, saved = h->stream;
    RFLU_TRY(get_ustream(h, h->tune.gemm_masked, &U));
    RFLU_HIP(hipStreamSynchronize(saved));
    h->stream = U;
    const int rc = launch_gemm<T>(h, M, N, K, A, lda, B, ldb, C, ldc);
    h->stream = saved;
    RFLU_TRY(rc);
    RFLU_HIP(hipStreamSynchronize(U));
    return RFLU_OK;
}

// measured on MI355X (bench.py --blocksize sweep): the knee moves right with the matrix size
static int64_t default_blocksize(int64_t mn)
{
    // (512 from 11265 columns on: where the update engine, which wants 512-wide block columns, starts to win -- N=11264 38.4 vs 38.6 ms at 256
    // through the streams, N=12288 41.5 vs 43.5, N=10240 33.8 vs 32.9: round 6)
    return mn < 1024 ? -1 : (mn <= 11264 ? 256 : (mn <= 16384 ? 512 : (mn <= 24576 ? 1024 : 2048)));
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
    hipEvent_t tail = nullptr;      // columns right of the first block column become valid with this event (getrf_cm_dev)

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
        // (NoPivot: launch_panel has already inverted the diagonal block into linv_at(r0), next to inv(U11) for its own rows)
        return RFLU_OK;
    }

    // reckernel! (src/lu.jl:189-263) on columns [c0, c1), rows [c0+roff, m)
    int rec(int64_t c0, int64_t c1)
    {
        const int64_t w = c1 - c0;
        if (w <= 0) return RFLU_OK;
        if (w <= NB) return leaf(c0, w);
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

// workgroups of the cooperative leaf on `rows` rows (the CUs a schedule has to keep free for it)
static int64_t panel_wgs(const Handle* h, int64_t rows, int pivot, size_t esize)
{
    return panel_plan_wgs(h, rows, esize, pivot);
}

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
static int get_masked_stream(Handle* h, hipStream_t* slot, int r);
int get_ustream(Handle* h, int reserve, hipStream_t* out)
{
    const int r = reserve / 32;
    if (reserve % 32 != 0 || r < 1 || r > 7) { set_error("CU reservation %d not in 32..224 step 32", reserve); return RFLU_ERR_ARG; }
    RFLU_TRY(get_masked_stream(h, &h->ustreams[r], r));
    *out = h->ustreams[r];
    return RFLU_OK;
}
static int get_masked_stream(Handle* h, hipStream_t* slot, int r)
{
    if (!*slot) {
        // CU mask bits are enumerated round-robin over the 8 XCDs (scripts/probes/cumask.hip): bits 0..31 are 4 CUs of
        // every XCD, and so on.  A mask that empties an XCD is ignored by the runtime, so whole 32-bit words are cleared.
        // The mask covers the CUs the device actually reports (num_cus / 32 words); callers only ask for a reservation
        // on a full 256-CU device (factor_lookahead).
        uint32_t mask[8];
        const int words = std::min(8, (h->num_cus + 31) / 32);
        for (int i = 0; i < 8; ++i) mask[i] = (i < r || i >= words) ? 0u : 0xffffffffu;
        if (words <= r || hipExtStreamCreateWithCUMask(slot, (uint32_t)words, mask) != hipSuccess) {
            (void)hipGetLastError();
            h->mask_failed = true;
            RFLU_HIP(hipStreamCreateWithFlags(slot, hipStreamNonBlocking));
        }
    }
    return RFLU_OK;
}

// The complement of get_ustream's mask: a stream confined to the `reserve` CUs the update stream never touches.
// While the factorization is update-bound the critical path has time to spare, and the workgroups of ITS GEMMs that land on
// shared CUs delay the update (scripts/microbench_gemm_vs_rec.py: -2 % on the masked 15872 x 14848 x 512 GEMM) -- so in that
// phase the critical path is kept on its own CUs (N=16384: 88.1 -> 86.8 ms).
static int get_pstream(Handle* h, int reserve, hipStream_t* out)
{
    const int r = reserve / 32;
    if (reserve % 32 != 0 || r < 1 || r > 7) { set_error("CU reservation %d not in 32..224 step 32", reserve); return RFLU_ERR_ARG; }
    if (!h->pstreams[r]) {
        uint32_t mask[8];
        const int words = std::min(8, (h->num_cus + 31) / 32);
        for (int i = 0; i < 8; ++i) mask[i] = (i < r && i < words) ? 0xffffffffu : 0u;
        if (words <= r || hipExtStreamCreateWithCUMask(&h->pstreams[r], (uint32_t)words, mask) != hipSuccess) {
            (void)hipGetLastError();
            RFLU_HIP(hipStreamCreateWithFlags(&h->pstreams[r], hipStreamNonBlocking));
        }
    }
    *out = h->pstreams[r];
    return RFLU_OK;
}

// ---- the schedules' streams on different hardware pipes -------------------------------------------------------------------------
// Every stream with a CU mask is an HSA queue of its own, and queues are spread round-robin over the 4 pipes of the compute
// micro-engine in the order the PROCESS created them.  Two busy queues on one pipe cost every kernel of both ~25 us
// (queue_probe_rate: two backlogged streams drain their one-thread kernels at 1.7 us per kernel on different pipes, 3.1 us when they are
// one and the same stream, 28 us when they share a pipe; N=4096 12 -> 20 ms, N=16384 80 -> 108 ms when the update or the side stream
// lands on the critical path's pipe).  With more than four busy streams somebody has to share; the library uses at most four.  Which
// queue index a new stream gets depends on how many queues the host program created before -- so it is measured, not assumed: each of
// the library's masked streams is probed against the caller's stream and the ones already accepted, and replaced by a new one with
// the same mask (the next queue index) until it shares a pipe with none of them; the rejected streams stay parked, idle.
// Re-checked when the caller's stream changes (rflu_set_stream) or a new masked stream appears.  RFLU_QUEUE_CHECK=0 skips it.
static int validate_queues(Handle* h)
{
    if (!h->tune.queue_check || h->queue_giveup) return RFLU_OK;
    int created = 0;
    for (int r = 1; r < 8; ++r) created += (h->ustreams[r] != nullptr) + (h->pstreams[r] != nullptr);
    const bool new_masked = h->queues_ok_count != created;   // a masked stream has appeared since the last check: everything is checked again
    if (!new_masked)
        for (hipStream_t ok : h->queues_ok_streams)
            if (ok == h->stream) return RFLU_OK;   // (a host program that alternates between a few streams is checked once per stream)
    if (!h->qprobe_slots) RFLU_HIP(hipMalloc((void**)&h->qprobe_slots, 8 * sizeof(long long)));
    const hipStream_t P = h->stream;
    constexpr int NPROBE = 128;   // the second half is timed (queue_probe_rate)
    constexpr size_t MAX_PARKED = 16;   // replaced streams stay parked (idle) for the life of the handle: bounded
    double base = 0;
    RFLU_TRY(queue_probe_rate(P, P, NPROBE, h->qprobe_slots, &base));
    RFLU_TRY(queue_probe_rate(P, P, NPROBE, h->qprobe_slots, &base));   // the first pass warms the launch path
    const double limit = std::min(std::max(2.0 * base, base + 5.0), 12.0);   // (a slow first reading of the base must not raise the bar to what a shared pipe reads)   // base = the caller's stream against itself (3.1 us); a shared pipe reads 28
    // a masked stream has to get along with the current caller stream and with the masked streams accepted before it.  (Not with the
    // caller streams it was accepted next to earlier: those are idle while this one is in use, and with two caller streams + three
    // masked streams there are more queues than pipes -- asking for that left the host entry's way-back stream on a shared pipe:
    // 120 -> 145 ms host to host.  A host that alternates between caller streams gets a new check when a stream had to be
    // replaced for the other one; the cap on parked streams and the give-up rule below bound what that can cost.)
    std::vector<hipStream_t> accepted{P};
    const bool verbose = h->tune.queue_trace != 0;
    bool unresolved = false;
    auto worst_next_to = [&](hipStream_t s, double* worst) -> int {
        *worst = 0;
        for (hipStream_t a : accepted) {
            double us = 0;
            RFLU_TRY(queue_probe_rate(a, s, NPROBE, h->qprobe_slots, &us));
            *worst = std::max(*worst, us);
        }
        return RFLU_OK;
    };
    auto settle = [&](hipStream_t* slot, int r, bool complement) -> int {
        for (int attempt = 0; attempt < 8; ++attempt) {
            double worst = 0;
            RFLU_TRY(worst_next_to(*slot, &worst));
            // wall-clock readings: a marginally slow one has to repeat before it counts; a clear one (a shared pipe reads ~28 us, nine
            // times the base) is taken at once -- the repeat of a clear reading was seen to come back low and leave the collision in place
            if (worst > limit && worst < 2.0 * limit) RFLU_TRY(worst_next_to(*slot, &worst));
            if (verbose)
                fprintf(stderr, "[rflu] queue check %s[%d] attempt %d: %.1f us per kernel next to the accepted streams (alone %.1f)\n",
                        complement ? "pstream" : "ustream", r, attempt, worst, base);
            if (worst <= limit) {
                // a good reading is confirmed once: a colliding pair was seen to read low now and then (the whole process then runs with two
                // queues on one pipe: N=8192 35 instead of 24 ms, N=16384 100 instead of 76 -- one process in a few dozen)
                double again = 0;
                RFLU_TRY(worst_next_to(*slot, &again));
                if (verbose && again > limit)
                    fprintf(stderr, "[rflu] queue check %s[%d] attempt %d: second reading %.1f us\n", complement ? "pstream" : "ustream", r, attempt, again);
                worst = std::max(worst, again);
            }
            if (worst <= limit) break;
            if (attempt == 7 || h->parked_streams.size() >= MAX_PARKED) {   // keep this one: never fail a factorization over placement
                unresolved = true;
                break;
            }
            h->parked_streams.push_back(*slot);
            h->queues_ok_streams.clear();   // what was accepted next to other caller streams is no longer what is in use
            *slot = nullptr;
            hipStream_t fresh;
            if (complement) RFLU_TRY(get_pstream(h, 32 * r, &fresh));
            else RFLU_TRY(get_ustream(h, 32 * r, &fresh));
            if (h->mask_failed) return RFLU_OK;
        }
        accepted.push_back(*slot);
        return RFLU_OK;
    };
    for (int r = 1; r < 8; ++r)
        if (h->ustreams[r]) RFLU_TRY(settle(&h->ustreams[r], r, false));
    for (int r = 1; r < 8; ++r)
        if (h->pstreams[r]) RFLU_TRY(settle(&h->pstreams[r], r, true));
    // more busy caller streams than there are pipes to spare (or a GPU shared with another process, whose load reads like a
    // conflict): after three checks that could not be settled the placement is taken as it is
    if (unresolved && ++h->queue_unresolved >= 3) {
        h->queue_giveup = true;
        if (verbose) fprintf(stderr, "[rflu] queue check: placement not settled after %d checks, %zu streams parked: no further checks on this handle\n",
                             h->queue_unresolved, h->parked_streams.size());
    }
    bool known = false;
    for (hipStream_t ok : h->queues_ok_streams) known = known || ok == P;
    if (!known) {
        if (h->queues_ok_streams.size() >= 8) h->queues_ok_streams.erase(h->queues_ok_streams.begin());
        h->queues_ok_streams.push_back(P);
    }
    h->queues_ok_count = 0;
    for (int r = 1; r < 8; ++r) h->queues_ok_count += (h->ustreams[r] != nullptr) + (h->pstreams[r] != nullptr);
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
// b_end < number of block columns: stop after block column b_end-1 (its update issued, block column b_end brought up to date on
// P) and hand over to factor_leafwise; *U_last = the update stream of that block column.
template <typename T>
static int factor_lookahead(Fact<T>& f, int64_t W, int64_t b_end, hipStream_t* U_last, int64_t W_wide = 0, int64_t wide_end = 0)
{
    // Block columns: [0, wide_end) in pieces of W_wide (a multiple of W; the update-bound part of a large matrix, whose bulk GEMM
    // wants the deeper K), the rest in pieces of W.  A block column is numbered by its first column / W ("id"): events, gate
    // values and b_end use that number, so the narrow part -- and factor_leafwise behind it -- see the numbering they would
    // see without a wide part.
    Handle* h = f.h;
    // tuning knobs (Tune): split_share = how much of what is left after the modelled time stays on the update stream;
    // max_reserve: taller panels (> 64 workgroups) take too many CUs from the update: one stream instead; min_reserve: least number
    // of CUs kept away from the update stream; split_scale scales the modelled critical-path time (0 = no split)
    const int min_reserve = (h->tune.min_reserve >= 32 && h->tune.min_reserve <= 224 && h->tune.min_reserve % 32 == 0) ? h->tune.min_reserve : 32;
    const double split_scale = h->tune.split_scale, split_share = h->tune.split_share;
    const bool split_all = h->tune.split_all != 0;
    const int max_reserve = h->tune.max_reserve;
    hipStream_t P = h->stream;
    const int64_t m = f.m, n = f.n, ld = f.ld, mn = std::min(m, n);
    T* R = f.R;
    if (W_wide <= W || wide_end <= 0) { W_wide = W; wide_end = 0; }
    wide_end = std::min(wide_end / W_wide * W_wide, mn);
    std::vector<int64_t> bstart;
    for (int64_t c = 0; c < mn; c += (c < wide_end ? W_wide : W)) bstart.push_back(c);
    bstart.push_back(mn);
    const int64_t nblk = (int64_t)bstart.size() - 1;            // block columns
    const int64_t nid = (mn + W - 1) / W;                       // ids
    auto id_of = [&](int64_t b) { return b >= nblk ? nid : bstart[b] / W; };
    auto width_of = [&](int64_t b) { return b < nblk ? bstart[b + 1] - bstart[b] : W; };
    hipEvent_t ev;
    // The critical path moves between the caller's stream and a stream confined to the reserved CUs (get_pstream); h->stream
    // follows it, and is put back on every way out of this function.
    struct Restore { Handle* h; hipStream_t s; ~Restore() { h->stream = s; } } restore{h, h->stream};
    const hipStream_t userS = h->stream;
    // Confining the critical path to the reserved CUs while the update is the bottleneck: round 2 measured +1.3 ms in its favour,
    // round 3 -2.4 ms against it with all four streams on pipes of their own (validate_queues; without that a fourth stream may share
    // a pipe with one of the other three, which costs 25-60 %): off unless RFLU_CONFINE_ROWS asks for it.
    const int64_t confine_rows = h->tune.confine_rows;
    auto move_P = [&](hipStream_t to, int64_t b) -> int {
        if (to == P) return RFLU_OK;
        hipEvent_t e0;
        RFLU_TRY(get_event(h, 4 * id_of(b) + 0, &e0));
        RFLU_HIP(hipEventRecord(e0, P));
        RFLU_HIP(hipStreamWaitEvent(to, e0, 0));
        P = to;
        h->stream = to;
        return RFLU_OK;
    };

    auto update = [&](hipStream_t st, int64_t j0, int64_t jb, int64_t c0, int64_t c1, LaswpGate gate = LaswpGate{},
                      GemmSignal sig = GemmSignal{}) -> int {
        // apply block column [j0, j0+jb) to columns [c0, c1): interchanges, block-row solve, Schur update
        // gate: hold the first launch until another stream's counter is reached; sig: publish when the first columns are done
        if (c1 <= c0) return RFLU_OK;
        hipStream_t saved = h->stream;
        h->stream = st;
        int rc = RFLU_OK;
        const int64_t je = j0 + jb;
        if (f.pivot) rc = launch_laswp2<T>(h, R, ld, c0, c1 - c0, 0, 0, j0 / NB, (je + NB - 1) / NB, 0, nullptr, nullptr, gate);
        else if (gate.wait_flag) rc = launch_gate_wait(h, gate.wait_flag, gate.wait_val);
        if (rc == RFLU_OK) rc = trsm_rec<T>(h, jb, c1 - c0, R + j0 * ld + j0, ld, R + j0 * ld + c0, ld, f.linv_at(j0));
        if (rc == RFLU_OK && m > je)
            rc = launch_gemm<T>(h, m - je, c1 - c0, jb, R + je * ld + j0, ld, R + j0 * ld + c0, ld, R + je * ld + c0, ld, sig);
        h->stream = saved;
        return rc;
    };
    // While the update stream is the bottleneck, block column b+2 is not updated by a launch sequence of its own (interchanges,
    // solves and a 496-tile GEMM that fills 1.1 rounds of the 448 workgroup slots: ~450 us per block column at N=16384) but as the
    // FIRST tile columns of the one bulk update; the GEMM publishes a gate when those tiles are done and the critical path waits
    // on that gate instead of an event.
    int64_t merge_rows = h->tune.merge_rows >= 0 ? h->tune.merge_rows : (sizeof(T) == 8 ? 8192 : (int64_t)1 << 40);
    if (h->tune.schedule_events) merge_rows = (int64_t)1 << 40;   // RFLU_SCHEDULE=events: no device-side gates (see getrf_rm)
    {   // the gate needs P and U to run concurrently: only with a real CU-masked update stream (create it now to find out)
        hipStream_t probe;
        RFLU_TRY(get_ustream(h, 32, &probe));
        if (h->mask_failed) merge_rows = (int64_t)1 << 40;
    }
    const unsigned long long ubase = h->gate_epoch;
    h->gate_epoch += (unsigned long long)nid + 2;
    auto uval = [&](int64_t b) { return ubase + (unsigned long long)id_of(b) + 1; };
    bool prev_merged = false;

    // events: 4b+1 = evP[b], 4b+2 = evU1[b], 4b+3 = evUend[b]
    hipStream_t Uprev = nullptr;      // update stream of the previous overlapped block column
    int64_t uend_prev = -1;           // its id (evUend recorded), -1: none
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

    for (int64_t b = 0; b < nblk && id_of(b) < b_end; ++b) {
        const int64_t j0 = bstart[b], jb = width_of(b), je = j0 + jb;
        const bool last_here = !(b + 1 < nblk && id_of(b + 1) < b_end);   // the next block column is somebody else's (or there is none)
        {
            const int64_t g_b = panel_wgs(h, m - j0, f.pivot, sizeof(T));
            const int res_b = std::max<int>(min_reserve, int((std::max<int64_t>(g_b, 1) + 31) / 32 * 32));
            hipStream_t to = userS;
            if (b > 0 && prev_overlapped && m - j0 >= confine_rows && res_b == 32)   // taller panels: restB needs the whole GPU
                RFLU_TRY(get_pstream(h, res_b, &to));
            RFLU_TRY(move_P(to, b));
        }
        // ---- panel b on P: Toledo recursion on the block column, interchanges confined to its own columns ----
        f.sw_lo = j0;
        f.sw_hi = je;
        RFLU_TRY(f.rec(j0, je));
        if (b == 0 && f.tail) {   // everything after the first panel may touch the columns whose layout change ran next to it
            RFLU_HIP(hipStreamWaitEvent(P, f.tail, 0));
            f.tail = nullptr;
        }
        RFLU_TRY(flush_pending());                                   // restB_{b-1}: whole GPU, after the panel
        // the panel that will run next to this block column's update is panel b+1
        const int64_t rows_next = m - je;
        const int64_t g_next = panel_wgs(h, rows_next, f.pivot, sizeof(T));
        int reserve = std::max<int>(min_reserve, int((std::max<int64_t>(g_next, 1) + 31) / 32 * 32));
        // RFLU_SWAP_LATE (default on): the last block column in front of a leaf-wise part that starts swapped sends its update to the
        // 192-CU stream, so that the 224-CU stream is free to be the side stream of the first leaf-wise block column (factor_leafwise)
        // (RFLU_SWAP_SU overrides factor_leafwise's stream assignment: only its "swapped behind a lookahead part" mode wants this move)
        const bool swapped_at_handover = h->tune.swap_su >= 0 ? h->tune.swap_su == 2 : h->tune.swap_late != 0;
        if (swapped_at_handover && reserve == 32 && last_here && b_end < nid && rows_next <= h->tune.swap_rows) reserve = 64;
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
            prev_merged = false;
            Uprev = nullptr;
            continue;
        }
        hipStream_t U;
        RFLU_TRY(get_ustream(h, reserve, &U));
        RFLU_TRY(get_event(h, 4 * id_of(b) + 1, &ev));
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
            RFLU_TRY(get_event(h, 4 * id_of(b) + 3, &ev));
            RFLU_HIP(hipEventRecord(ev, U));
            uend_prev = id_of(b);
            Uprev = U;
            break;
        }
        const int64_t n1e = std::min(je + width_of(b + 1), n);                              // end of block column b+1
        const int64_t n2e = std::min(n1e + width_of(b + 2), n);                             // end of block column b+2
        // ---- P: next block column (needs rest_{b-1}.part1, which updated exactly these columns).  Handing all but its
        // first leaf to U (and gating P's second leaf on it) was measured slower: U's in-order queue is still busy with
        // rest_{b-1} in the early, update-bound block columns.
        LaswpGate pgate;
        if (b > 0 && prev_overlapped) {
            if (prev_merged) {
                pgate.wait_flag = h->gates + 3;
                pgate.wait_val = uval(b - 1);
                pgate.info = h->info_dev;
            } else {
                RFLU_TRY(get_event(h, 4 * id_of(b - 1) + 2, &ev));
                RFLU_HIP(hipStreamWaitEvent(P, ev, 0));
            }
        }
        RFLU_TRY(update(P, j0, jb, je, n1e, pgate));
        // ---- U: block column b+2 first (the next `next`), then as much of the rest as fits next to P's work ----
        const bool merged = m - je >= merge_rows && reserve == 32 && jb >= 256 && n2e > n1e && m > je &&
                            !(b_end < nid && last_here) && b + 1 < nblk;
        if (!merged) {
            RFLU_TRY(update(U, j0, jb, n1e, n2e));
            RFLU_TRY(get_event(h, 4 * id_of(b) + 2, &ev));
            RFLU_HIP(hipEventRecord(ev, U));
        }
        int64_t cA = n;                                                                      // restA = [n2e, cA)
        // With the minimal reservation the masked stream keeps 7/8 of the GPU and a split cannot win more than ~1 %
        // (measured: nothing); it pays for the tall panels, whose reservation takes a quarter to half of the CUs.
        if (split_scale > 0 && m > je && (reserve > 32 || split_all)) {
            const double rateU = model_gemm_flops_per_us(jb, 256 - reserve, sizeof(T));
            const double rateP = model_gemm_flops_per_us(jb, 256, sizeof(T));
            const double col_flops = 2.0 * double(m - je) * double(jb);                      // per trailing column
            const double tP = split_scale * ((je < mn ? model_panel_us(m - je, std::min(width_of(b + 1), mn - je)) : 0.0)
                                             + col_flops * double(n1e - je) / rateP);
            const double colsA = tP * rateU / col_flops;                                     // columns U finishes in tP
            const int64_t left = n - n1e;
            // what is left after tP is shared by both streams; below one GEMM tile column it is not worth a launch
            int64_t a = int64_t(colsA) / 128 * 128;
            a = std::max<int64_t>(a, n2e - n1e);
            if (left - a >= 512)
                cA = n1e + a + int64_t(split_share * double(left - a) * double(256 - reserve) / 512.0) / 128 * 128;
        }
        if (merged) {
            GemmSignal sig;
            sig.first_cols = n2e - n1e;
            sig.flag = h->gates + 3;
            sig.val = uval(b);
            sig.cnt = reinterpret_cast<unsigned*>(h->gates + 5);
            RFLU_TRY(update(U, j0, jb, n1e, cA, LaswpGate{}, sig));
        } else {
            RFLU_TRY(update(U, j0, jb, n2e, cA));
        }
        prev_merged = merged;
        RFLU_TRY(get_event(h, 4 * id_of(b) + 3, &ev));
        RFLU_HIP(hipEventRecord(ev, U));
        if (cA < n) {
            pend.valid = true;
            pend.j0 = j0;
            pend.jb = jb;
            pend.c0 = cA;
            pend.need_uend = uend_prev;   // restA_{b-1} may have written columns right of cA
        }
        uend_prev = id_of(b);
        Uprev = U;
        prev_overlapped = true;
        if (h->progress) RFLU_TRY(h->progress(j0));   // block column b-1's last piece (restB) went out after panel b: rows above j0 are settled
    }
    RFLU_TRY(flush_pending());
    if (h->progress) RFLU_TRY(h->progress(std::min(std::min(nid, b_end) * W, mn)));
    RFLU_TRY(move_P(userS, nblk));
    f.sw_lo = 0;
    f.sw_hi = -1;
    if (U_last) *U_last = Uprev;
    if (b_end < nid) return RFLU_OK;   // factor_leafwise goes on from here and joins at its end
    // join: P continues only after U has drained
    if (uend_prev >= 0) {
        RFLU_TRY(get_event(h, 4 * uend_prev + 3, &ev));
        RFLU_HIP(hipStreamWaitEvent(P, ev, 0));
    }
    return RFLU_OK;
}

// The persistent update engine (engine.hip) can serve a factorization when the column blocks start on tile boundaries, the
// workspace allows 16-byte accesses and nothing else wants to follow the schedule from the host (the host entry's progress hook).
template <typename T>
static int engine_usable(const Handle* h, const Fact<T>& f, int64_t W)
{
    constexpr int64_t VW = 16 / (int64_t)sizeof(T);
    return W % 128 == 0 && f.roff == 0 && reinterpret_cast<uintptr_t>(f.R) % 16 == 0 && f.ld % VW == 0 &&
           f.m < (int64_t)1 << 30 && f.n < (int64_t)1 << 30 && (f.n + W - 1) / W <= ENG_MAX_CB && !h->progress && !h->mask_failed &&
           (!h->tune.schedule_events || h->tune.engine_replay) && h->num_cus == 256 &&
           (f.m >= f.n || f.m % W == 0);   // (a fat matrix whose last panel ends inside a column block: the columns right of it in that block)
}

// the engine's state block (device) and the pinned host image of its initial value: both or neither (a half-made pair would
// have the next call write its image through a null pointer); freed by rflu_destroy
static int ensure_engine_state(Handle* h)
{
    if (h->eng_state && h->eng_host) return RFLU_OK;
    if (!h->eng_host) RFLU_HIP(hipHostMalloc(&h->eng_host, sizeof(EngState), hipHostMallocDefault));
    if (!h->eng_state) {
        if (hipMalloc(&h->eng_state, sizeof(EngState)) != hipSuccess) {
            (void)hipGetLastError();
            h->eng_state = nullptr;
            set_error("hipMalloc of the update engine's state failed");
            return RFLU_ERR_HIP;
        }
    }
    return RFLU_OK;
}

// Leaf-wise schedule: the critical path is nothing but the chain of cooperative leaves.
//
// The recursion's merges (solve + Schur update of the right half) and the block-column lookahead put ~640 us of small
// dependent launches between the leaves of every 512-column block (scripts/trace_timeline.sh) -- as much as a third of the
// late, panel-bound phase.  Here every leaf g (64 columns) is applied right-looking, and only the 64 columns the NEXT leaf
// needs stay on the critical-path stream:
//   P  : leaf g -> [wait: leaf g-1 applied to LA = [c0+64, c0+128) by the side stream] {interchanges of leaf g on LA, inverse
//        of its diagonal block} (gate P >= g; both gates ride on this launch) -> solve + update of LA (K = 64) -> leaf g+1 ...
//   S  : [wait gate P >= g] leaf g applied to the rest of its own block column (gate S-in >= g) and to the next block column
//        (first leaf of a block: after evU1[b-1]) (gate S-all >= g)
//   U  : once per block column b, after its last leaf: [wait gate S-all] the deferred interchanges on the columns to the
//        left, then block column b (K = W) applied to everything right of block column b+1 -- block column b+2 first
//        (evU1[b]) -- exactly the update stream of factor_lookahead.
// Queues: the critical path stays on the caller's stream and there is ONE side stream (the next block column's part of the first
// leaf of a block waits for evU1 between two gates of its own).  An earlier version with two side streams and the critical path
// on a third, CU-confined stream ran at 115-118 ms for N=16384 instead of 88 -- two of its streams shared a hardware pipe, as round 3
// found out (validate_queues now places every stream; with four streams there is no pipe to spare).  hipEvent edges per leaf cost 40-50 us of
// bubble per record/wait on the hot stream, hipStreamWaitValue64/WriteValue64 were slower still: hence the device-side gates.
// Measured (Float64, ms): N=4096 13.6 -> 12.0, N=8192 29.7 -> 26.8, N=12288 51.6 -> 49.6; N=16384 whole matrix 88.3, from the
// first panel of <= 8192 rows on (after factor_lookahead) 84.8 vs 86.1.
// S and U share the CU mask that keeps the panel's CUs free.  Every column receives the same eliminations in the same order
// as in reckernel! (src/lu.jl:189-263); inside a block column the Schur complement is accumulated 64 pivots at a time instead
// of in the recursion's growing chunks, so factors agree with the one-stream path to rounding, pivots exactly.
// Engine mode (eng_end > 0, b_begin == 0): the side stream's and the update stream's work of block columns [0, eng_end) -- and the
// block-column updates every column right of them needs from those -- is pulled by the persistent update engine (engine.hip) from
// per-column-block counters instead of being enqueued on S and U; P is unchanged except that its lookahead launch waits for the
// engine's progress word of the lookahead strip's column block instead of a side-stream gate.  From block column eng_end on the
// streams take over again (short panels: the XCD-local leaves need CUs the resident engine does not give back).
template <typename T>
static int factor_leafwise(Fact<T>& f, int64_t W, int64_t b_begin, hipStream_t U_before, int64_t eng_end = 0)
{
    Handle* h = f.h;
    const int64_t m = f.m, n = f.n, ld = f.ld, mn = std::min(m, n);
    T* R = f.R;
    const hipStream_t userS = h->stream;
    struct Restore { Handle* h; hipStream_t s; ~Restore() { h->stream = s; } } restore{h, userS};
    hipStream_t P = userS;
    const int64_t nblk = (mn + W - 1) / W, nleaf = (mn + NB - 1) / NB;
    // events as in factor_lookahead: block b -> 4b+2 (evU1), 4b+3 (evUend); stream moves of this function from EX on
    const size_t EX = 4 * (size_t)nblk + 8;
    auto evU1 = [](int64_t b) { return 4 * (size_t)b + 2; };
    auto evUend = [](int64_t b) { return 4 * (size_t)b + 3; };
    // RFLU_CONFINE_ROWS: panels at least this tall run on the stream confined to the reserved CUs (default: never -- the
    // critical path stays on the caller's stream: three active queues in all, see the header comment)
    // Which of the two masked streams is which.  When the whole matrix is factored leaf-wise (no lookahead part before it) the
    // SIDE stream gets the 224-CU mask and the update stream the 192-CU one: 32 CUs the bulk GEMM never touches are then always
    // free for the side stream's per-leaf kernels, and the update has the slack to pay for it (N=4096 12.18 -> 12.03 ms, N=8192
    // 26.91 -> 26.49).  After a lookahead part the 224-CU stream is still busy with that part's last bulk update when the first
    // leaf needs the side stream (2.4 ms stall), and moving that update to the 192-CU stream costs what the swap wins (N=16384
    // 85.1 vs 85.3 ms, N=12288 49.0 vs 48.8): there the update keeps 224 CUs and the side stream takes the 192-CU stream.
    // (Float32 at N=16384 is leaf-wise from block column 0 as well, but there the update still needs its 224 CUs: 63.9 vs 62.1 ms.)
    // Round 4: swapped from the first panel of at most swap_rows (8192) rows on, wherever that is (swap_mode 2): from block column
    // bs on the side stream is the 224-CU stream, and from U(bs-1) on every update goes to the 192-CU one -- the last update in front
    // of the swap too (it is factor_lookahead's when bs is the first leaf-wise block column), so that the 224-CU stream is idle when
    // the side stream moves there.  N=16384 79.1 -> 77.75 ms (swapped one block column later, without moving that update: 78.0),
    // N=12288 45.9 -> 45.7.  What it is for (scripts/rocpd_timeline.py, scripts/gate_trace.py): a bulk GEMM that STARTS fills every
    // workgroup slot its mask allows at once, and its tiles then finish in rounds of ~130 us -- a side stream confined to the same
    // CUs gets its three small kernels per leaf placed one round boundary at a time (96 + 128 + 211 us instead of 6 + 12 + 30) and
    // the critical path stalls on gate 1 at the third / fourth leaf of every block column (200..500 us each, 3 ms in all).
    const int swap_mode = h->tune.swap_su >= 0 ? h->tune.swap_su : ((b_begin == 0 && m <= 8192) ? 1 : h->tune.swap_late ? 2 : 0);
    const int64_t bs = std::max<int64_t>(b_begin, (std::max<int64_t>(m - h->tune.swap_rows, 0) + W - 1) / W);   // first swapped block column
    auto swap_s = [&](int64_t b) { return swap_mode == 1 || (swap_mode == 2 && b >= bs); };       // side stream of block column b on the 224-CU stream
    auto swap_u = [&](int64_t b) { return swap_mode == 1 || (swap_mode == 2 && b + 1 >= bs); };   // U(b) on the 192-CU stream
    const bool fold = h->tune.gate_fold != 0 && !h->tune.gate_trace;
    const int64_t confine_rows = h->tune.confine_rows;
    auto reserve_for = [&](int64_t rows) {
        const int64_t g = panel_wgs(h, rows, f.pivot, sizeof(T));
        return std::max<int>(32, int((g + 31) / 32 * 32));
    };
    auto wait_on = [&](hipStream_t st, size_t idx) -> int {
        hipEvent_t e;
        RFLU_TRY(get_event(h, idx, &e));
        RFLU_HIP(hipStreamWaitEvent(st, e, 0));
        return RFLU_OK;
    };
    auto record_on = [&](hipStream_t st, size_t idx) -> int {
        hipEvent_t e;
        RFLU_TRY(get_event(h, idx, &e));
        RFLU_HIP(hipEventRecord(e, st));
        return RFLU_OK;
    };
    // leaf (rows r0.., columns c0..c0+w) applied to columns [a, b): interchanges (optional), block-row solve, Schur update
    auto apply_leaf = [&](hipStream_t st, int64_t c0, int64_t w, int64_t a, int64_t b, bool swaps) -> int {
        if (b <= a) return RFLU_OK;
        hipStream_t saved = h->stream;
        h->stream = st;
        int rc = RFLU_OK;
        if (swaps && f.pivot) rc = launch_laswp<T>(h, R, ld, a, b - a, c0 / NB, c0 / NB + 1);
        if (rc == RFLU_OK) rc = launch_trsm_inv64<T>(h, w, b - a, f.linv_at(c0), R + c0 * ld + a, ld);
        if (rc == RFLU_OK && m > c0 + w)
            rc = launch_gemm<T>(h, m - c0 - w, b - a, w, R + (c0 + w) * ld + c0, ld, R + c0 * ld + a, ld, R + (c0 + w) * ld + a, ld);
        h->stream = saved;
        return rc;
    };
    auto update = [&](hipStream_t st, int64_t j0, int64_t jb, int64_t c0, int64_t c1) -> int {
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
    hipStream_t Uprev = U_before, Sprev = nullptr;   // U_before: the update stream of block b_begin-1 (factor_lookahead)
    const unsigned long long gbase = h->gate_epoch;
    h->gate_epoch += (unsigned long long)nleaf + 2;
    auto val = [&](int64_t g) { return gbase + (unsigned long long)g + 1; };
    const int64_t gfirst = b_begin * W / NB;   // the leaves before it were factored (and applied everywhere) by factor_lookahead
    if (h->tune.gate_trace && !h->gate_stamps) {
        RFLU_HIP(hipMalloc((void**)&h->gate_stamps, 3 * 4096 * sizeof(long long)));
        RFLU_HIP(hipMemset(h->gate_stamps, 0, 3 * 4096 * sizeof(long long)));
    }
    auto stamp = [&](int which, int64_t g) -> long long* { return (h->gate_stamps && g < 4096) ? h->gate_stamps + which * 4096 + g : nullptr; };
    // ---- engine mode: state, launch ----
    const int64_t LPB = W / NB;
    EngGeo geo{};
    EngState* est = nullptr;
    hipStream_t E = nullptr;
    int64_t eng_retire_leaf = -1;   // engine mode: the leaf in front of which the engine's workgroups on the chain's XCD are gone (-1: they stay)
    struct RestoreLocal { Handle* h; int64_t rows; bool on; ~RestoreLocal() { if (on) h->tune.panel_local_rows = rows; } } restore_local{h, h->tune.panel_local_rows, false};
    if (eng_end > 0) {
        if (b_begin != 0) { set_error("factor_leafwise: the engine starts at block column 0"); return RFLU_ERR_ARG; }
        eng_end = std::min(eng_end, nblk);
        RFLU_TRY(get_ustream(h, 32, &E));
        RFLU_TRY(ensure_engine_state(h));
        est = static_cast<EngState*>(h->eng_state);
        EngState* img = static_cast<EngState*>(h->eng_host);
        geo.m = (int)m; geo.n = (int)n; geo.mn = (int)mn; geo.W = (int)W; geo.nbp = (int)eng_end;
        geo.Wc = (h->tune.engine_wc >= 128 && h->tune.engine_wc % 128 == 0 && W % h->tune.engine_wc == 0) ? h->tune.engine_wc : (int)W;
        geo.ncb = (int)((n + geo.Wc - 1) / geo.Wc);
        geo.pivot = f.pivot;
        geo.ahead = std::max(1, std::min(h->tune.engine_ahead, 4));
        if (geo.ncb > ENG_MAX_CB) { geo.Wc = (int)W; geo.ncb = (int)((n + W - 1) / W); }
        const size_t bytes = offsetof(EngState, cb) + (size_t)geo.ncb * sizeof(EngCB);
        const size_t skip = offsetof(EngState, remaining);   // (the arrival word in front belongs to the feeding stream: getrf_host_engine)
        memset(img, 0, bytes);
        for (int cb = 0; cb < geo.ncb; ++cb) {
            EngCB& c = img->cb[cb];
            const int end = 2 * eng_nops(geo, cb);
            int sq = 0;
            while (sq < end && eng_units_of(eng_op(geo, cb, sq >> 1), sq & 1, geo.m) == 0) ++sq;
            c.prog = 2ull * (unsigned long long)(sq >> 1);
            c.claim = sq < end ? (unsigned long long)sq << 32 : (unsigned long long)ENG_SEQ_DONE << 32;
            img->remaining += sq < end;
            const int nleft = eng_nleft(geo, cb);
            int lk = 0;
            while (lk < nleft && eng_left_units<T>(geo, cb, lk) == 0) ++lk;
            c.lclaim = lk < nleft ? (unsigned long long)lk << 32 : (unsigned long long)ENG_SEQ_DONE << 32;
            c.lprog = (unsigned long long)lk;   // (left ops without units count as done)
            if (nleft > 0 && lk > 0) img->cb[eng_first_cb(geo, eng_pb(geo, cb))].leftdone += 1ull << 32;   // ... towards the block column's own count too
            img->remaining += lk < nleft;
        }
        if (h->tune.engine_replay) {   // measurement: every leaf counts as done before the engine starts (the chain below is skipped)
            if (eng_end < nblk) { set_error("RFLU_ENGINE_REPLAY needs the engine to the end"); return RFLU_ERR_ARG; }
            RFLU_TRY(launch_gate_signal(h, h->gate_ptr[0], val(nleaf - 1)));
            if (f.tail) { RFLU_HIP(hipStreamWaitEvent(P, f.tail, 0)); f.tail = nullptr; }
        }
        // the initial state travels on the caller's stream, in front of everything the engine is going to wait for
        RFLU_HIP(hipMemcpyAsync(reinterpret_cast<char*>(est) + skip, reinterpret_cast<char*>(img) + skip, bytes - skip, hipMemcpyHostToDevice, P));
        RFLU_TRY(record_on(P, EX + (size_t)nblk + 1));
        RFLU_TRY(wait_on(E, EX + (size_t)nblk + 1));
        EngArgs<T> a;
        a.R = R; a.ld = ld; a.g = geo; a.policy = h->tune.engine_policy;
        a.linv = static_cast<const T*>(h->linv); a.pm_cnt = h->pm_cnt; a.pm_dst = h->pm_dst; a.pm_src = h->pm_src;
        a.st = est; a.leaf_gate = h->gate_ptr[0]; a.gate_base = gbase; a.info = h->info_dev; a.gemm_flags = h->tune.gemm_flags;
        a.arrived = h->eng_host_mode ? &est->arrived : nullptr;
        a.rows_final = h->eng_host_mode ? h->eng_rows_final_dev : nullptr;
        a.write_through = h->tune.engine_write_through != 0;
        a.leaf_xcds = h->tune.engine_leaf_xcds;
        a.leaf_wgs = h->tune.engine_leaf_wgs;
        a.solve_rl = h->tune.engine_solve_rl != 0;
        // host entry: whole-block-column operations that lag the chain by this many block columns go first (engine.hip), so that
        // block rows become final -- and leave -- while the factorization runs
        a.host_lag = h->eng_host_mode ? h->tune.engine_host_lag : 0;
        // Engine to the end: from the first panel of at most `local_rows` rows on the chain wants its XCD-local leaves back (worth 1.2 ms at
        // N=16384), and the engine has little left to do: its workgroups on the chain's XCD retire two leaves earlier, the first such
        // leaf waits until they are gone (RFLU_ENGINE_RETIRE=0: they stay, every leaf any-placement)
        eng_retire_leaf = -1;
        a.retire_xcc = -1; a.retire_leaf = 0;
        {
            // (RFLU_ENGINE_RETIRE = the panel height from which on: at most what the XCD-local leaf is used for anyway)
            const int64_t local_max = restore_local.rows >= 0 ? restore_local.rows : (sizeof(T) == 4 ? 8192 : 4096);
            // (default -1: 4096 rows, 2048 from 16384 rows on, where the end is bound by the engine's throughput and its workgroups are worth
            // more than the faster leaves for longer: N=16384 75.2-76.5 ms at 4096 / 74.4-75.2 at 2048, N=12288 44.8 / 47.1, N=8192 25.8 / 27.7)
            const int64_t retire_rows = h->tune.engine_retire >= 0 ? h->tune.engine_retire : (m >= 16384 ? 2048 : 4096);
            const int64_t local_rows = std::min<int64_t>(local_max, retire_rows);
            if (retire_rows > 0 && eng_end >= nblk && !h->eng_host_mode && !h->tune.engine_replay && f.pivot && h->panel_local == 2 && !h->coop_launch && local_rows >= 1024 && m > local_rows + 4 * NB) {
                eng_retire_leaf = (m - local_rows + NB - 1) / NB;   // first leaf whose panel has at most local_rows rows
                a.retire_xcc = h->panel_xcc;
                a.retire_leaf = (int)std::max<int64_t>(1, eng_retire_leaf - 2);
            }
        }
        a.trace = nullptr;
        long long*& eng_trace_buf = h->eng_trace_buf;   // measurement only (RFLU_ENGINE_TRACE=1): stamps of the leaf windows, printed at the next call
        if (env_str("RFLU_ENGINE_TRACE")) {
            if (!eng_trace_buf) RFLU_HIP(hipMalloc((void**)&eng_trace_buf, (4096 * 4 + 16) * sizeof(long long)));
            else {
                std::vector<long long> hs(4096 * 4 + 16);
                RFLU_HIP(hipMemcpy(hs.data(), eng_trace_buf, hs.size() * sizeof(long long), hipMemcpyDeviceToHost));
                double s01 = 0, s12 = 0, s23 = 0, sq = 0; int cnt = 0;
                for (int g = 1; g + 1 < (int)nleaf && g < 4095; ++g) {
                    if (!hs[g * 4] || !hs[g * 4 + 3] || !hs[(g - 1) * 4 + 3]) continue;
                    s01 += (hs[g * 4 + 1] - hs[g * 4]) / 100.0; s12 += (hs[g * 4 + 2] - hs[g * 4 + 1]) / 100.0; s23 += (hs[g * 4 + 3] - hs[g * 4 + 2]) / 100.0;
                    sq += (hs[g * 4] - hs[(g - 1) * 4 + 3]) / 100.0; ++cnt;
                }
                if (cnt) fprintf(stderr, "[rflu] engine trace (previous call, %d leaf windows on their first column block): first claim -> stage 0 done %.1f us, -> first tile claimed %.1f, -> window complete %.1f; previous window complete -> first claim %.1f us\n", cnt, s01 / cnt, s12 / cnt, s23 / cnt, sq / cnt);
                {
                    const long long* ac = hs.data() + 4096 * 4;
                    const double tot = (double)(ac[0] + ac[1] + ac[2] + ac[3] + ac[4]);
                    if (tot > 0)
                        fprintf(stderr, "[rflu] engine workgroup time (previous call, %.1f workgroup-ms): block-column tiles %.1f %%, leaf-window tiles %.1f %%, strips + solves %.1f %%, deferred interchanges %.1f %%, between units %.1f %% (of which asleep with nothing eligible %.1f %%, count + publication behind a unit %.1f %%, scan / claim / acquire %.1f %%)\n",
                                tot / 1e5, 100.0 * ac[0] / tot, 100.0 * ac[1] / tot, 100.0 * ac[2] / tot, 100.0 * ac[3] / tot, 100.0 * ac[4] / tot, 100.0 * ac[5] / tot, 100.0 * ac[6] / tot, 100.0 * (ac[4] - ac[5] - ac[6]) / tot);
                    if (tot > 0 && ac[15] > 0)   // the scan by itself, per call (= per unit), in microseconds
                        fprintf(stderr, "[rflu] engine scan (previous call, %lld calls, %.2f sweeps per call), us per call: epoch / gate sample %.2f, exit words %.2f, claim words + choice %.2f, ticket %.2f, "
                                        "scan of the deferred interchanges %.2f, acquire + hand-over %.2f; asleep %.2f\n",
                                ac[15], (double)ac[14] / ac[15], ac[8] / 100.0 / ac[15], ac[9] / 100.0 / ac[15], ac[10] / 100.0 / ac[15], ac[11] / 100.0 / ac[15], ac[12] / 100.0 / ac[15],
                                ac[13] / 100.0 / ac[15], ac[5] / 100.0 / ac[15]);
                }
                // leaf by leaf (RFLU_ENGINE_TRACE=first:count): when LEAF(g) was first claimed on the column block of its first columns (ms since LEAF(0)'s first
                // claim), its three phases, and how long that column block had been idle before (the engine waiting for the chain) -- for the LAST
                // leaf but one of a block column that column block is the NEXT block column's: the window the chain's last leaf waits for
                int tg0 = 40, tgn = 4;
                if (const char* e = env_str("RFLU_ENGINE_TRACE")) { if (strchr(e, ':')) sscanf(e, "%d:%d", &tg0, &tgn); }
                for (int g = std::max(tg0, 1); g < tg0 + tgn && g + 1 < (int)nleaf && g < 4095; ++g)
                {
                    fprintf(stderr, "   leaf %d: first claim at %.3f ms | stage 0 %.1f us | to first tile %.1f | tiles %.1f | idle before %.1f", g, (hs[g * 4] - hs[0]) / 1e5,
                            (hs[g * 4 + 1] - hs[g * 4]) / 100.0, (hs[g * 4 + 2] - hs[g * 4 + 1]) / 100.0, (hs[g * 4 + 3] - hs[g * 4 + 2]) / 100.0, (hs[g * 4] - hs[(g - 1) * 4 + 3]) / 100.0);
                    const long long* n4 = hs.data() + (size_t)(2048 + g) * 4;   // the same leaf on the next block column's first column block
                    if (g < 2048 && n4[0] && n4[3])
                        fprintf(stderr, " || next block column: first claim at %.3f ms (%.1f us after the previous leaf's window there was complete) | stage 0 %.1f | to first tile %.1f | tiles %.1f",
                                (n4[0] - hs[0]) / 1e5, (n4[0] - n4[-1]) / 100.0, (n4[1] - n4[0]) / 100.0, (n4[2] - n4[1]) / 100.0, (n4[3] - n4[2]) / 100.0);
                    fprintf(stderr, "\n");
                    // behind the last leaf of a block column b: what stands between it and BIG(b) being complete on the column block the chain needs next
                    const int LPBt = (int)(W / NB);
                    if ((g + 1) % LPBt == 0 && g / LPBt < 512) {
                        const int b = g / LPBt;
                        const long long* bg = hs.data() + (size_t)(1024 + b) * 4;
                        const long long* lf = hs.data() + (size_t)(1536 + b) * 4;
                        const long long t0 = n4[0];   // LEAF(last leaf of b) first claimed on the next block column: the leaf has just been factored
                        if (t0 && bg[0] && bg[3])
                            fprintf(stderr, "   block column %d ends (its last leaf's window is claimed at +0): own deferred interchanges +%.0f .. +%.0f us | BIG(%d) on the first column block of block column %d: "
                                            "stage 0 +%.0f .. +%.0f | tiles +%.0f .. +%.0f us\n",
                                    b, (lf[0] - t0) / 100.0, (lf[1] - t0) / 100.0, b, b + 2, (bg[0] - t0) / 100.0, (bg[1] - t0) / 100.0, (bg[2] - t0) / 100.0, (bg[3] - t0) / 100.0);
                    }
                }
            }
            RFLU_HIP(hipMemsetAsync(eng_trace_buf, 0, (4096 * 4 + 16) * sizeof(long long), P));
            a.trace = eng_trace_buf;
        }
        const int wgs = h->tune.engine_wgs > 0 ? h->tune.engine_wgs : 2 * (h->num_cus - 32);
        if (img->remaining > 0) {
            // measurement (rflu_profile_enable(2)): the engine kernel as ONE launch of the class the bulk GEMM reports under -- its flops are
            // the Schur updates it performs (every operation's 2 M N K), its duration the whole residency, waiting included
            // ... and its algorithmic bytes: per operation the panel pieces once (A: M x K, B: K x N), the Schur block in and out (2 M N), the
            // solved block row in and out (2 K N), and the interchanges (two rows read + written per pivot and column: 4 per entry), the
            // deferred ones on the finished columns to the left included
            double eng_flops = 0, eng_bytes = 0;
            for (int cb = 0; cb < geo.ncb; ++cb) {
                for (int k = 0; k < eng_nops(geo, cb); ++k) {
                    const EngOp o = eng_op(geo, cb, k);
                    if (o.nc <= 0) continue;
                    const double M = (double)std::max(geo.m - (o.j0 + o.jb), 0), N = (double)o.nc, K = (double)o.jb;
                    eng_flops += 2.0 * M * N * K;
                    eng_bytes += sizeof(T) * (M * K + K * N + 2.0 * M * N + 2.0 * K * N + 0.5 * K * K + (f.pivot ? 4.0 * K * N : 0.0));
                }
                if (f.pivot && eng_pb(geo, cb) < geo.nbp) {   // left op 0 (on average half of the block column's pivots per strip) + the later block columns as a whole
                    const double nc = (double)(std::min<int64_t>(n, (int64_t)(cb + 1) * geo.Wc) - (int64_t)cb * geo.Wc);
                    const double later = (double)std::max<int64_t>(std::min<int64_t>((int64_t)geo.nbp * W, mn) - (int64_t)(eng_pb(geo, cb) + 1) * W, 0);
                    eng_bytes += sizeof(T) * 4.0 * nc * (0.5 * (double)std::min<int64_t>(W, mn - (int64_t)eng_pb(geo, cb) * W) + later);
                }
            }
            hipStream_t saved = h->stream;
            h->stream = E;
            int rc;
            {
                ProfScope ps(h, RFLU_K_GEMM, eng_flops, eng_bytes);
                rc = launch_engine<T>(h, E, a, wgs);
            }
            h->stream = saved;
            RFLU_TRY(rc);
        }
        RFLU_TRY(record_on(E, evUend(eng_end - 1)));   // the engine leaves when every column block has received everything it owes
        h->eng_active = true;
        // While the engine is resident the only CUs with room are the 4 per XCD its mask leaves out: the any-placement leaves (at most
        // 32 workgroups, one per CU) fit there, the XCD-local ones (all participants on ONE XCD) would wait for CUs it never gives back
        restore_local.on = true;
        h->tune.panel_local_rows = 0;
        Uprev = E;
    }
    for (int64_t b = b_begin; b < nblk; ++b) {
        const bool in_eng = b < eng_end;
        if (in_eng && h->tune.engine_replay) continue;   // (measurement: the engine alone)
        if (eng_end > 0 && b == eng_end) {   // the streams take over: the leaves are short enough for the XCD-local exchange again
            h->tune.panel_local_rows = restore_local.rows;
            restore_local.on = false;
        }
        const int64_t j0 = b * W, jb = std::min(W, mn - j0), je = j0 + jb;
        const int64_t bend = std::min(j0 + W, n), wend = std::min(j0 + 2 * W, n);
        const int res = reserve_for(m - j0);
        // The side stream is the update stream of the 64-CU reservation: it keeps away from the panel's 32 CUs like U does,
        // and a taller matrix has already used it for its first block columns -- one queue less to place (validate_queues).
        if (res != 32) { set_error("factor_leafwise: panel of %lld rows needs more than 32 CUs", (long long)(m - j0)); return RFLU_ERR_ARG; }
        hipStream_t S = nullptr;
        if (!in_eng) RFLU_TRY(get_ustream(h, swap_s(b) ? 32 : 64, &S));
        if (!in_eng)
        {   // the critical path runs on the reserved CUs while the update stream is the bottleneck (see get_pstream)
            hipStream_t to = userS;
            if (m - j0 >= confine_rows && res == 32) RFLU_TRY(get_pstream(h, res, &to));
            if (to != P) {
                RFLU_TRY(record_on(P, EX + (size_t)b));
                RFLU_TRY(wait_on(to, EX + (size_t)b));
                P = to;
                h->stream = to;
            }
        }
        const int64_t g0 = j0 / NB, nl = (jb + NB - 1) / NB;
        if (in_eng && h->eng_host_mode)   // host entry: the block column (and the lookahead strip of its last leaf) has to be in place
            RFLU_TRY(launch_eng_wait(h, &est->arrived, (unsigned long long)std::min<int64_t>(n, je + NB)));
        for (int64_t i = 0; i < nl; ++i) {
            const int64_t g = g0 + i, c0 = j0 + i * NB, w = std::min<int64_t>(NB, je - c0);
            if (in_eng && g == eng_retire_leaf) {   // the chain's XCD is its own again: XCD-local leaves from here on
                RFLU_TRY(launch_eng_wait_retired(h, h->panel_xcc));
                h->tune.panel_local_rows = restore_local.rows;   // (every panel from here on is at most engine_retire rows tall)
            }
            RFLU_TRY(launch_panel<T>(h, R, ld, m, c0, c0, w, f.ipiv, f.pivot));
            const int64_t la0 = c0 + w, la1 = std::min(la0 + NB, n);
            const unsigned long long* wflag = nullptr;   // leaf g-1 reached LA through the side stream: in its own block's part, or the next block's
            unsigned long long wval = g > 0 ? val(g - 1) : 0;
            if (la1 > la0 && g > gfirst) {
                if (eng_end > 0 && (g - 1) / LPB < eng_end) {   // ... through the engine: LEAF(g - 1) is complete on the lookahead strip's column block
                    const int cb_la = (int)(la0 / geo.Wc);
                    wflag = &est->cb[cb_la].prog;
                    wval = 2ull * (unsigned long long)eng_leaf_op_index(geo, cb_la, (int)(g - 1)) + 1;   // (its first tile column: engine.hpp, prog)
                } else {
                    wflag = h->gate_ptr[la0 < std::min(((c0 - NB) / W + 1) * W, n) ? 1 : 2];
                }
            }
            if (eng_end > 0 && f.tail && la1 > W) {   // the first launch of the critical path that touches the columns whose layout change
                RFLU_HIP(hipStreamWaitEvent(P, f.tail, 0));   // ran on the engine's stream next to the first leaves
                f.tail = nullptr;
            }
            // one launch for {interchanges on LA, diagonal inverse, 64-row solve of LA}: full leaves with a full, 16-byte aligned LA
            const bool fuse = f.pivot && fold && h->tune.leaf_fuse && w == NB && la1 - la0 == NB && m - c0 > NB &&
                              reinterpret_cast<uintptr_t>(R) % 16 == 0 && ld % (16 / (int64_t)sizeof(T)) == 0;
            if (fuse) {
                LaswpGate gt;
                gt.wait_flag = wflag;
                gt.wait_val = wflag ? wval : 0;
                gt.signal_flag = h->gate_ptr[0];
                gt.signal_val = val(g);
                gt.signal_cnt = reinterpret_cast<unsigned*>(h->gates + 4);
                gt.info = h->info_dev;
                RFLU_TRY(launch_leaf_la<T>(h, R, ld, la0, c0 / NB, c0, R + c0 * ld + c0, f.linv_at(c0), gt));
                RFLU_TRY(launch_gemm<T>(h, m - c0 - w, la1 - la0, w, R + (c0 + w) * ld + c0, ld, R + c0 * ld + la0, ld, R + (c0 + w) * ld + la0, ld));
            } else if (f.pivot && fold) {   // both gates ride on the interchange launch: two launches less per leaf on this stream
                LaswpGate gt;
                gt.wait_flag = wflag;
                gt.wait_val = wflag ? wval : 0;
                gt.signal_flag = h->gate_ptr[0];
                gt.signal_val = val(g);
                gt.signal_cnt = reinterpret_cast<unsigned*>(h->gates + 4);
                gt.info = h->info_dev;
                RFLU_TRY(launch_laswp2<T>(h, R, ld, la0, la1 - la0, 0, 0, c0 / NB, c0 / NB + 1, w, R + c0 * ld + c0, f.linv_at(c0), gt));
            } else {
                if (wflag) RFLU_TRY(launch_gate_wait(h, wflag, wval));
                if (f.pivot) RFLU_TRY(launch_laswp2<T>(h, R, ld, la0, la1 - la0, 0, 0, c0 / NB, c0 / NB + 1, w, R + c0 * ld + c0, f.linv_at(c0)));

                RFLU_TRY(launch_gate_signal(h, h->gate_ptr[0], val(g), stamp(0, g)));
            }
            if (!fuse) RFLU_TRY(apply_leaf(P, c0, w, la0, la1, false));
            if (in_eng) continue;   // (the engine applies the leaf to the rest of this block column and to the next one)
            // ---- side stream: leaf g on the rest of this block column and on the next one ----
            h->stream = S;
            int rc = launch_gate_wait(h, h->gate_ptr[0], val(g));
            if (rc == RFLU_OK && i == 0 && S != Sprev && g > gfirst && !(eng_end > 0 && (g - 1) / LPB < eng_end))
                rc = launch_gate_wait(h, h->gate_ptr[2], val(g - 1));
            if (i == 0 && b > 0) {
                // the next block column holds U(b-1)'s update only after evU1[b-1]; the critical path needs this block
                // column's part first, so the leaf is applied in two pieces with a gate of its own in between.
                // (Round 4 tried leaving the next block column's part of the first 1..6 leaves to a later leaf, so that the in-order
                // side stream does not sit on the event with the own parts of the next leaves queued behind it: no gain, N=16384
                // 78.2-78.9 vs 78.7-79.1 ms -- the event is not what the side stream waits for, see swap_mode above.)
                if (eng_end > 0 && b == eng_end) {
                    // hand-over from the update engine: the rest of THIS block column is the side stream's from here on, and the engine may
                    // still be applying the previous block column's last leaves to it (the chain's wait covered the first tile column of the
                    // lookahead strip's column block only): every column block of block column b must have completed its sequence
                    for (int c = eng_first_cb(geo, (int)b); rc == RFLU_OK && c < eng_first_cb(geo, (int)b) + eng_cbs_of_block(geo, (int)b); ++c)
                        if (eng_nops(geo, c) > 0) {
                            h->stream = S;
                            rc = launch_eng_wait(h, &est->cb[c].prog, 2ull * (unsigned long long)eng_nops(geo, c));
                        }
                }
                if (rc == RFLU_OK) rc = apply_leaf(S, c0, w, la1, bend, true);
                h->stream = S;
                if (rc == RFLU_OK) rc = launch_gate_signal(h, h->gate_ptr[1], val(g), stamp(1, g));
                hipEvent_t e;
                if (eng_end > 0 && b == eng_end) {
                    // behind the update engine: the next block column is up to date when all its operations are complete
                    for (int c = eng_first_cb(geo, (int)(b + 1)); rc == RFLU_OK && c < eng_first_cb(geo, (int)(b + 1)) + eng_cbs_of_block(geo, (int)(b + 1)); ++c)
                        if (eng_nops(geo, c) > 0) {
                            h->stream = S;
                            rc = launch_eng_wait(h, &est->cb[c].prog, 2ull * (unsigned long long)eng_nops(geo, c));
                        }
                } else {
                if (rc == RFLU_OK) rc = get_event(h, evU1(b - 1), &e);
                if (rc == RFLU_OK && hipStreamWaitEvent(S, e, 0) != hipSuccess) { set_error("hipStreamWaitEvent failed"); rc = RFLU_ERR_HIP; }
                }
                if (rc == RFLU_OK) rc = apply_leaf(S, c0, w, std::max(la1, bend), wend, true);
                h->stream = S;
                if (rc == RFLU_OK) rc = launch_gate_signal(h, h->gate_ptr[2], val(g), stamp(2, g));
            } else {
                if (rc == RFLU_OK) rc = apply_leaf(S, c0, w, la1, wend, true);
                h->stream = S;
                if (rc == RFLU_OK) rc = launch_gate_signal(h, h->gate_ptr[1], val(g), stamp(1, g));
                if (rc == RFLU_OK) rc = launch_gate_signal(h, h->gate_ptr[2], val(g), stamp(2, g));
            }
            h->stream = P;
            RFLU_TRY(rc);
        }
        if (in_eng) continue;
        Sprev = S;
        // ---- U(b): everything right of block column b+1, and the interchanges nobody needed until now ----
        hipStream_t U;
        RFLU_TRY(get_ustream(h, swap_u(b) ? 64 : reserve_for(m - je), &U));
        const int64_t glast = g0 + nl - 1;
        {
            h->stream = U;
            const int rc = launch_gate_wait(h, h->gate_ptr[2], val(glast));
            h->stream = P;
            RFLU_TRY(rc);
        }
        if (Uprev && Uprev != U) RFLU_TRY(wait_on(U, evUend(b - 1)));
        if (f.pivot) {
            hipStream_t saved = h->stream;
            h->stream = U;
            int rc = RFLU_OK;
            for (int64_t i = 0; i + 1 < nl && rc == RFLU_OK; ++i)   // leaf i's columns: the later leaves' interchanges
                rc = launch_laswp<T>(h, R, ld, j0 + i * NB, NB, g0 + i + 1, g0 + nl);
            if (rc == RFLU_OK && j0 > 0) rc = launch_laswp<T>(h, R, ld, 0, j0, g0, g0 + nl);
            h->stream = saved;
            RFLU_TRY(rc);
        }
        const int64_t p1e = std::min(wend + W, n);
        RFLU_TRY(update(U, j0, jb, wend, p1e));
        RFLU_TRY(record_on(U, evU1(b)));
        RFLU_TRY(update(U, j0, jb, p1e, n));
        RFLU_TRY(record_on(U, evUend(b)));
        Uprev = U;
        if (h->progress) RFLU_TRY(h->progress(je));
    }
    if (P != userS) {
        RFLU_TRY(record_on(P, EX + (size_t)nblk));
        RFLU_TRY(wait_on(userS, EX + (size_t)nblk));
        P = userS;
        h->stream = userS;
    }
    RFLU_TRY(wait_on(userS, evUend(nblk - 1)));
    h->eng_active = false;
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
    h->coop_leaf_seq = 0;
    RFLU_HIP(hipMemsetAsync(h->info_dev, 0, 2 * sizeof(int64_t), h->stream));
    // the wrapping "last workgroup" counters of the folded gates: a factorization that timed out or was aborted may have left
    // them mid-count, and a stale count would publish the next factorization's gate early or never
    RFLU_HIP(hipMemsetAsync(h->gates + 4, 0, 2 * sizeof(unsigned long long), h->stream));
    if (!pivot && ipiv) RFLU_TRY(launch_iota_ipiv(h, ipiv, 0, mn));  // src/lu.jl:111-113

    Fact<T> f{h, R, ld, m, n, ipiv, pivot};
    bool fat_tail_done = false;
    const bool default_bs = blocksize == 0;
    if (blocksize == 0) blocksize = default_blocksize(mn);
    // column-major entry with the tail of the layout change still in flight (getrf_cm_dev): only factor_lookahead knows where the
    // first access to those columns is; every other path waits for it here
    hipEvent_t tail = h->tail_event;
    h->tail_event = nullptr;
    const bool two_stream = !(blocksize < 0 || blocksize >= mn) && !(h->prof || h->prof_one_stream) && h->num_cus == 256;
    if (tail && !two_stream) {
        RFLU_HIP(hipStreamWaitEvent(h->stream, tail, 0));
        tail = nullptr;
    }
    if (blocksize < 0 || blocksize >= mn) {
        h->last_path = RFLU_PATH_HIP_RECURSIVE;
        RFLU_TRY(f.rec(0, mn));
    } else if (!(h->prof || h->prof_one_stream) && h->num_cus == 256) {   // the CU reservation of the two-stream schedule is laid out for 8 x 32 CUs
        h->last_path = RFLU_PATH_HIP_LOOKAHEAD;
        // tall block columns (update-bound): factor_lookahead; from the first panel of at most lw_rows rows on: factor_leafwise.
        // RFLU_LEAFWISE=0 keeps the lookahead schedule to the end.
        int leafwise = h->tune.leafwise;
        // rocprofv3 --pmc runs ONE kernel at a time across all queues: a gate kernel waiting for another stream's kernel would
        // never see it start (the 2 s gate timeout turns that into RFLU_ERR_TIMEOUT).  A counter-collection run therefore asks
        // for RFLU_SCHEDULE=events: the lookahead schedule, whose cross-stream edges are hipEvents, to the end (same kernels on
        // the same shapes for the bulk of the work; scripts/collect_profiles.sh sets it for the --pmc passes only).
        if (h->tune.schedule_events && !h->tune.engine_replay) leafwise = 0;
        {   // the leaf-wise / deep schedules hand work between streams through device-side gates: only with real CU-masked streams
            hipStream_t probe;
            RFLU_TRY(get_ustream(h, 32, &probe));
            RFLU_TRY(get_ustream(h, 64, &probe));
            if (h->mask_failed) leafwise = 0;
            if (!h->mask_failed) RFLU_TRY(validate_queues(h));
        }
        int64_t Wb = round_up(blocksize, NB);
        // Large matrices with the default block width: WIDE block columns (1024 / 2048: the bulk GEMM at K >= 1024 runs at 0.88-0.91 of
        // the MFMA peak instead of 0.83) while the update is the bottleneck, i.e. up to the last `narrow_cols` columns; those are
        // factored the way a matrix of that size is -- 512-wide block columns, leaf-wise from 8192 rows on -- because there the
        // chain of panels sets the pace and a 2048-column recursion on the critical path is what costs.
        int64_t W_wide = 0, wide_end = 0;
        if (default_bs && h->tune.wide_narrow && Wb > 512 && mn >= 20480) {
            const int64_t narrow_cols = std::max<int64_t>(h->tune.narrow_cols, 2048);
            W_wide = Wb;
            Wb = 512;
            wide_end = std::max<int64_t>(mn - narrow_cols, 0) / W_wide * W_wide;
        }
        const auto t_enq0 = std::chrono::steady_clock::now();
        const int64_t nblk = (mn + Wb - 1) / Wb;
        int64_t b_switch = nblk;   // first block column of the leaf-wise part
        // (block columns wider than 512 make the side stream's per-leaf window -- up to 2 W columns at K = 64 -- too much work to
        //  finish within one leaf: N=32768 at W=2048 is 0.8 % (Float64) / 5 % (Float32) slower leaf-wise, so those stay as they were)
        if (leafwise && Wb >= 2 * NB && Wb <= 512) {
            // Float64: panels at most this tall are the bottleneck of their block column (N=16384: 84.8 ms at 7168-8192, 85.4 at
            // 9216, 88.3 for the whole matrix); Float32's faster GEMM leaves the panel the bottleneck everywhere (61.7 vs 64.5 ms)
            int64_t lw_rows = sizeof(T) == 8 ? 8192 : 16384;   // 16384 rows = 32 workgroups: the most the 32 reserved CUs take
            if (h->tune.leafwise_rows >= 0) lw_rows = h->tune.leafwise_rows;
            lw_rows = std::min<int64_t>(lw_rows, 32 * (int64_t)PANEL_THREADS);
            b_switch = m <= lw_rows ? 0 : std::min(nblk, (m - lw_rows + Wb - 1) / Wb);
        }
        // the block column in front of the leaf-wise part has to be a narrow one (factor_leafwise finds its events by number)
        if (W_wide > 0 && b_switch < nblk) wide_end = std::min(wide_end, std::max<int64_t>(b_switch - 1, 0) * Wb / W_wide * W_wide);
        hipStream_t U_last = nullptr;
        {
            // the update engine serves the block columns whose panels are taller than engine_rows (the leaf-wise schedule from block
            // column 0, its side / update streams replaced by the engine); below that the streams and the XCD-local leaves take over
            int64_t eng_end = 0;
            // the engine where asked for (RFLU_ENGINE=1, the host entry) or, by default, where it measures faster than the streams: with
            // pivoting at the default block width of 512, i.e. more than 11264 columns (N=16384 Float64: 71.5 vs 75 ms, Float32 -- round 6 --
            // 55.3 vs 58.8; N=12288: 41.5 vs 43.5 through the streams at 256, Float32 36.9 vs 38.4; NoPivot at N=16384: 68.9 vs 64.1, the
            // streams stay; below, at 256-wide block columns, the streams win: N=10240 32.9 vs 33.8, N=8192 24.0 vs 24.9).  A Float32 pivot search may answer another summation order with another (equally valid) pivot sequence from a
            // near-tie on -- between the stream schedules too: tests hold Float32 to the residual and a floor of equal
            // leading pivots, not to the bits of another schedule.
            const bool eng_wanted = h->eng_host_mode || h->tune.engine == 1 || h->tune.engine_replay ||
                                    (h->tune.engine < 0 && pivot && default_bs && Wb == 512 && mn > 11264 && m >= n);
            if (eng_wanted && leafwise && Wb >= 2 * NB && Wb <= 512 && W_wide == 0 && m <= 32 * (int64_t)PANEL_THREADS && engine_usable<T>(h, f, Wb)) {
                const int64_t er = h->eng_host_mode ? 0 : std::max<int64_t>(h->tune.engine_rows, 0);   // (host entry: every block column through the engine)
                eng_end = m <= er ? 0 : std::min(nblk, (m - er + Wb - 1) / Wb);
            }
            if (h->eng_host_mode && eng_end < nblk) {   // the caller feeds the matrix in behind our back: only the engine waits for it
                set_error("getrf: host entry through the engine asked for a schedule the engine cannot serve");
                return RFLU_ERR_ARG;
            }
            if (tail && b_switch == 0 && eng_end == 0) {
                RFLU_HIP(hipStreamWaitEvent(h->stream, tail, 0));
                tail = nullptr;
            }
            f.tail = tail;
            if (eng_end > 0) {
                h->last_path = RFLU_PATH_HIP_ENGINE;
                RFLU_TRY(factor_leafwise<T>(f, Wb, 0, nullptr, eng_end));
                b_switch = nblk;
            } else if (b_switch > 0) RFLU_TRY(factor_lookahead<T>(f, Wb, b_switch, &U_last, W_wide, wide_end));
            if (b_switch < nblk) RFLU_TRY(factor_leafwise<T>(f, Wb, b_switch, U_last));
        }
        if (h->tune.time_enqueue)
            fprintf(stderr, "[rflu] host enqueue time %.2f ms\n",
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_enq0).count());
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

    if (h->before_sync) RFLU_TRY(h->before_sync());
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
    const int64_t ldr = workspace_ld(h, n);
    RFLU_TRY(ensure_buffer(&h->work, &h->work_bytes, (size_t)m * (size_t)ldr * sizeof(T)));
    T* R = static_cast<T*>(h->work);
    // Layout change in two pieces: the first block column on the caller's stream, the rest on the CU-masked update stream while
    // the first panel (2.3 ms on 32 CUs at N=16384, nothing else to do) already runs.
    const int64_t mn = std::min(m, n);
    int64_t W0 = blocksize == 0 ? default_blocksize(mn) : blocksize;
    W0 = W0 > 0 ? round_up(W0, NB) : 0;
    const bool tail_overlap = h->tune.tail_overlap != 0;
    if (tail_overlap && !(h->prof || h->prof_one_stream) && h->num_cus == 256 && mn >= 12288 && W0 > 0 && W0 < mn && n - W0 >= 4096) {
        if (!h->tail_event_obj) {
            RFLU_HIP(hipEventCreateWithFlags(&h->tail_event_obj, hipEventDisableTiming));
            RFLU_HIP(hipEventCreateWithFlags(&h->tail_fork_obj, hipEventDisableTiming));
        }
        hipStream_t U0, user = h->stream;
        RFLU_TRY(get_ustream(h, 32, &U0));
        {   // settle which streams the schedules will use BEFORE work goes onto one of them (validate_queues may replace a stream)
            hipStream_t s64;
            RFLU_TRY(get_ustream(h, 64, &s64));
            if (!h->mask_failed) RFLU_TRY(validate_queues(h));
            RFLU_TRY(get_ustream(h, 32, &U0));
        }
        RFLU_HIP(hipEventRecord(h->tail_fork_obj, user));            // whatever produced A on the caller's stream
        RFLU_HIP(hipStreamWaitEvent(U0, h->tail_fork_obj, 0));
        RFLU_TRY(launch_transpose<T>(h, m, W0, A, lda, R, ldr));
        h->stream = U0;
        const int rc = launch_transpose<T>(h, m, n - W0, A + W0 * lda, lda, R + W0, ldr);
        h->stream = user;
        RFLU_TRY(rc);
        RFLU_HIP(hipEventRecord(h->tail_event_obj, U0));
        h->tail_event = h->tail_event_obj;
    } else {
        RFLU_TRY(launch_transpose<T>(h, m, n, A, lda, R, ldr));
    }
    const int rc_f = getrf_rm<T>(h, m, n, R, ldr, ipiv, pivot, blocksize, info);
    if (h->tail_event_obj && rc_f != RFLU_OK) {
        // error before the tail event was consumed: the layout change may still be reading A / writing the workspace on the
        // update stream -- do not hand either back to the caller while it runs
        hipStream_t U0 = nullptr;
        if (get_ustream(h, 32, &U0) == RFLU_OK && U0) (void)hipStreamSynchronize(U0);
    }
    h->tail_event = nullptr;
    RFLU_TRY(rc_f);
    if (!h->out_done) RFLU_TRY(launch_transpose<T>(h, n, m, R, ldr, A, lda));   // (the host entry has taken the factors out piece by piece)
    RFLU_HIP(hipStreamSynchronize(h->stream));
    return RFLU_OK;
}

// ---- host entry through the update engine: the way in overlaps the factorization (round 5) -----------------------------------------
// The reference's boundary is a host array (src/lu.jl:116-121).  Round 3 overlapped the way BACK with the factorization; the way in
// (38 ms of PCIe for a 16384^2 Float64 matrix) still preceded everything, because the stream schedules' first update touches every
// column.  The engine's per-column-block dataflow does not: a column block's operations become eligible when its columns have
// arrived, so the matrix is fed in block column by block column (a second host thread: copies from pageable memory block their
// caller) -- copy, layout change, a word that says how many columns are in place -- while the critical-path stream, which waits on the
// same word, factors what is there; finished block rows leave as before, told by a host-visible word the engine keeps
// (EngArgs::rows_final) instead of events.  Every block column goes through the engine here (no hand-over to the streams).
// *handled = false: not a case for this path (the caller falls back to getrf_host's sequence), nothing has been touched.
template <typename T>
static int getrf_host_engine(Handle* h, int64_t m, int64_t n, T* A, int64_t lda, int64_t* ipiv, int pivot, int64_t blocksize,
                             int64_t* info, bool* handled)
{
    *handled = false;
    const int64_t mn = std::min(m, n);
    const int64_t chunk = h->tune.host_early_out;
    const int64_t W = round_up(blocksize == 0 ? default_blocksize(mn) : blocksize, NB);
    // With pivoting only (round 6: Float32 too -- its device entry takes the engine at the headline size as well; a Float32 pivot search may
    // answer the engine's summation order with another, equally valid pivot sequence than a stream schedule's from a near-tie on, which is
    // why tests hold Float32 to the residual and a floor of equal leading pivots).  An unpivoted factorization shows another summation order
    // in visibly other digits and measures slower through the engine (N=16384: 68.9 vs 64.1 ms): it keeps the stream path, whose host entry
    // is bit-identical to the device entry
    if (!pivot) return RFLU_OK;
    if (!h->tune.engine_host || h->prof || h->prof_one_stream || h->num_cus != 256 || !h->tune.leafwise || h->tune.schedule_events ||
        chunk < 64 || mn < 8192 || m > 32 * (int64_t)PANEL_THREADS || m < n || W < 2 * NB || W > 512 || W % 128 != 0 || W >= mn ||
        (blocksize == 0 && mn >= 20480))
        return RFLU_OK;
    const int64_t ldr = workspace_ld(h, n);
    hipStream_t E, IN, OUT;
    RFLU_TRY(get_ustream(h, 32, &E));
    RFLU_TRY(get_ustream(h, 64, &E));    // (what getrf_rm creates before it settles the queues: nothing new appears afterwards)
    RFLU_TRY(get_pstream(h, 32, &IN));   // streams confined to the CUs the resident engine leaves free: anything else would wait for it
    RFLU_TRY(get_pstream(h, 64, &OUT));
    if (h->mask_failed) return RFLU_OK;
    RFLU_TRY(validate_queues(h));
    RFLU_TRY(get_pstream(h, 32, &IN));
    RFLU_TRY(get_pstream(h, 64, &OUT));
    // buffers: device copy of the input (kept intact for a failed call), row-major workspace, staging + pinned bounce buffers of the way back
    RFLU_TRY(ensure_buffer(&h->hostA_dev, &h->hostA_bytes, (size_t)m * (size_t)n * sizeof(T)));
    RFLU_TRY(ensure_buffer(&h->work, &h->work_bytes, (size_t)m * (size_t)ldr * sizeof(T)));
    if ((size_t)mn > h->ipiv_cap) {
        if (h->ipiv_dev) RFLU_HIP(hipFree(h->ipiv_dev));
        h->ipiv_dev = nullptr;
        h->ipiv_cap = 0;
        RFLU_HIP(hipMalloc((void**)&h->ipiv_dev, (size_t)mn * sizeof(int64_t)));
        h->ipiv_cap = (size_t)mn;
    }
    const size_t bounce_bytes = (size_t)std::min(chunk, m) * (size_t)n * sizeof(T);
    RFLU_TRY(ensure_buffer(&h->out_stage, &h->out_stage_bytes, 2 * bounce_bytes));
    if (h->bounce_bytes < bounce_bytes) {
        for (int i = 0; i < 2; ++i) {
            if (h->bounce[i]) RFLU_HIP(hipHostFree(h->bounce[i]));
            h->bounce[i] = nullptr;
        }
        h->bounce_bytes = 0;
        for (int i = 0; i < 2; ++i)
            if (hipHostMalloc(&h->bounce[i], bounce_bytes) != hipSuccess) {
                (void)hipGetLastError();
                for (int k = 0; k < 2; ++k) { if (h->bounce[k]) (void)hipHostFree(h->bounce[k]); h->bounce[k] = nullptr; }
                return RFLU_OK;   // no pinned memory to be had: the plain sequence needs none
            }
        h->bounce_bytes = bounce_bytes;
    }
    RFLU_TRY(ensure_engine_state(h));
    if (!h->eng_rows_final) {
        void* p = nullptr;
        if (hipHostMalloc(&p, 64, hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return RFLU_OK; }
        h->eng_rows_final = static_cast<unsigned long long*>(p);
        void* d = nullptr;
        RFLU_HIP(hipHostGetDevicePointer(&d, p, 0));
        h->eng_rows_final_dev = static_cast<unsigned long long*>(d);
    }
    *handled = true;
    *info = 0;
    EngState* est = static_cast<EngState*>(h->eng_state);
    T* dA = static_cast<T*>(h->hostA_dev);
    T* R = static_cast<T*>(h->work);
    const hipStream_t user = h->stream;
    const bool trace = h->tune.host_trace != 0;
    const auto t_call = std::chrono::steady_clock::now();
    auto since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    __atomic_store_n(h->eng_rows_final, 0ull, __ATOMIC_RELEASE);
    RFLU_HIP(hipMemsetAsync(&est->arrived, 0, sizeof(unsigned long long), user));
    RFLU_HIP(hipStreamSynchronize(user));   // (whatever the caller had in flight on this stream is done, the arrival word reads 0)
    // ---- the way in: a thread of its own (a copy from pageable memory returns when the data has left the host)
    std::atomic<int> feed_status{RFLU_OK};
    std::atomic<bool> feed_stop{false};
    const int device = h->device;
    const int64_t in_cols = std::max<int64_t>(W, 512);
    std::thread feeder([&, device]() {
        if (hipSetDevice(device) != hipSuccess) { feed_status = RFLU_ERR_HIP; return; }
        for (int64_t c0 = 0; c0 < n && !feed_stop.load(); c0 += in_cols) {
            const int64_t nc = std::min(in_cols, n - c0);
            if (hipMemcpy2DAsync(dA + c0 * m, (size_t)m * sizeof(T), A + c0 * lda, (size_t)lda * sizeof(T), (size_t)m * sizeof(T), (size_t)nc,
                                 hipMemcpyHostToDevice, IN) != hipSuccess ||
                launch_transpose_on<T>(IN, m, nc, dA + c0 * m, m, R + c0, ldr) != RFLU_OK ||
                launch_gate_signal_on(IN, &est->arrived, (unsigned long long)(c0 + nc)) != RFLU_OK) {
                feed_status = RFLU_ERR_HIP;
                return;
            }
        }
    });
    struct Join { std::thread& t; std::atomic<bool>& stop; ~Join() { stop = true; if (t.joinable()) t.join(); } } join{feeder, feed_stop};
    // ---- the way back: runs on this thread once the factorization is enqueued (Handle::before_sync), chunk by chunk as the engine
    // reports block rows final
    bool scattered = false;
    size_t ev_used = 0;
    auto new_event = [h, &ev_used](hipEvent_t* e) -> int {
        if (ev_used == h->out_events.size()) {
            hipEvent_t x;
            RFLU_HIP(hipEventCreateWithFlags(&x, hipEventDisableTiming));
            h->out_events.push_back(x);
        }
        *e = h->out_events[ev_used++];
        return RFLU_OK;
    };
    struct Reset { Handle* h; ~Reset() { h->before_sync = nullptr; h->out_done = false; h->eng_host_mode = false; } } reset{h};
    h->eng_host_mode = true;
    h->before_sync = [&]() -> int {
        hipEvent_t all_done;
        RFLU_TRY(new_event(&all_done));
        RFLU_HIP(hipEventRecord(all_done, user));   // the whole factorization (the critical-path stream joins the engine's at its end)
        if (trace) fprintf(stderr, "[rflu] host entry (engine): enqueue done %.1f ms after the call\n", since(t_call));
        std::vector<int64_t> ends;
        for (int64_t r = 0; r < m;) { r = std::min(m, r + chunk); ends.push_back(r); }
        const size_t nchunks = ends.size();
        std::vector<hipEvent_t> landed(nchunks);
        auto start_of = [&](size_t k) { return k == 0 ? (int64_t)0 : ends[k - 1]; };
        bool everything = false;
        auto wait_final = [&](int64_t r1) -> int {   // rows [0, r1) final: the engine's word, or the end of everything
            const auto t0 = std::chrono::steady_clock::now();
            while (!everything && (int64_t)__atomic_load_n(h->eng_rows_final, __ATOMIC_ACQUIRE) < r1) {
                const hipError_t q = hipEventQuery(all_done);
                if (q == hipSuccess) { everything = true; break; }
                if (q != hipErrorNotReady) { set_error("hipEventQuery failed: %s", hipGetErrorString(q)); return RFLU_ERR_HIP; }
                if (feed_status.load() != RFLU_OK) { set_error("host entry: feeding the matrix to the device failed"); return RFLU_ERR_HIP; }
                if (since(t0) > 20000.0) { set_error("host entry: no progress for 20 s"); return RFLU_ERR_TIMEOUT; }
                std::this_thread::sleep_for(std::chrono::microseconds(20));
            }
            return RFLU_OK;
        };
        auto send = [&](size_t k) -> int {   // chunk k: final -> transpose into a contiguous piece of the staging copy -> bounce buffer
            const int64_t r0 = start_of(k), rows = ends[k] - r0;
            RFLU_TRY(wait_final(ends[k]));
            T* piece = static_cast<T*>(h->out_stage) + (k & 1) * (size_t)std::min(chunk, m) * (size_t)n;
            RFLU_TRY(launch_transpose_on<T>(OUT, n, rows, R + r0 * ldr, ldr, piece, rows));
            RFLU_HIP(hipMemcpyAsync(h->bounce[k & 1], piece, (size_t)rows * (size_t)n * sizeof(T), hipMemcpyDeviceToHost, OUT));
            RFLU_TRY(new_event(&landed[k]));
            RFLU_HIP(hipEventRecord(landed[k], OUT));
            return RFLU_OK;
        };
        const int nthreads = std::max(1, std::min(h->tune.host_threads, 64));
        for (size_t k = 0; k < std::min<size_t>(2, nchunks); ++k) RFLU_TRY(send(k));
        for (size_t k = 0; k < nchunks; ++k) {
            RFLU_HIP(hipEventSynchronize(landed[k]));
            const int64_t r0 = start_of(k), rows = ends[k] - r0;
            const T* src = static_cast<const T*>(h->bounce[k & 1]);
            scattered = true;
            auto scatter = [&](int64_t j0, int64_t j1) {
                for (int64_t j = j0; j < j1; ++j) memcpy(A + j * lda + r0, src + j * rows, (size_t)rows * sizeof(T));
            };
            if (nthreads == 1 || (size_t)rows * (size_t)n * sizeof(T) < ((size_t)8 << 20)) {
                scatter(0, n);
            } else {
                std::vector<std::thread> pool;
                const int64_t per = (n + nthreads - 1) / nthreads;
                int64_t done_to = std::min<int64_t>(n, per);
                try {
                    for (int t = 1; t < nthreads; ++t) {
                        pool.emplace_back(scatter, std::min<int64_t>(n, t * per), std::min<int64_t>(n, (t + 1) * per));
                        done_to = std::min<int64_t>(n, (t + 1) * per);
                    }
                } catch (...) {
                }
                scatter(0, std::min<int64_t>(n, per));
                if (done_to < n) scatter(done_to, n);
                for (std::thread& th : pool) th.join();
            }
            if (trace) fprintf(stderr, "[rflu] host entry (engine): rows [%lld, %lld) home at %.1f ms\n", (long long)r0, (long long)ends[k], since(t_call));
            if (k + 2 < nchunks) RFLU_TRY(send(k + 2));
        }
        h->out_done = true;
        return RFLU_OK;
    };
    int rc = getrf_rm<T>(h, m, n, R, ldr, (pivot || ipiv) ? h->ipiv_dev : nullptr, pivot, blocksize, info);
    feed_stop = true;
    if (feeder.joinable()) feeder.join();
    if (rc == RFLU_OK && feed_status.load() != RFLU_OK) { set_error("host entry: feeding the matrix to the device failed"); rc = feed_status.load(); }
    if (rc != RFLU_OK) {
        // a failed call leaves the caller's matrix as it was (the device copy of the input is never written by the factorization)
        (void)hipDeviceSynchronize();
        if (scattered)
            (void)hipMemcpy2D(A, (size_t)lda * sizeof(T), dA, (size_t)m * sizeof(T), (size_t)m * sizeof(T), (size_t)n, hipMemcpyDeviceToHost);
        return rc;
    }
    if (!h->out_done) {   // (cannot happen: before_sync either brings everything home or fails)
        set_error("host entry: the factors did not travel back");
        return RFLU_ERR_ARG;
    }
    if (ipiv) RFLU_HIP(hipMemcpyAsync(ipiv, h->ipiv_dev, (size_t)mn * sizeof(int64_t), hipMemcpyDeviceToHost, user));
    RFLU_HIP(hipStreamSynchronize(user));
    RFLU_HIP(hipStreamSynchronize(IN));
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
    {   // the way in overlapped with the factorization (the update engine's dataflow waits for columns; the stream schedules cannot)
        bool handled = false;
        const int rc = getrf_host_engine<T>(h, m, n, A, lda, ipiv, pivot, blocksize, info, &handled);
        if (handled || rc != RFLU_OK) return rc;
    }
    RFLU_TRY(ensure_buffer(&h->hostA_dev, &h->hostA_bytes, (size_t)m * (size_t)n * sizeof(T)));
    if ((size_t)mn > h->ipiv_cap) {
        if (h->ipiv_dev) RFLU_HIP(hipFree(h->ipiv_dev));
        h->ipiv_dev = nullptr;
        h->ipiv_cap = 0;
        RFLU_HIP(hipMalloc((void**)&h->ipiv_dev, (size_t)mn * sizeof(int64_t)));
        h->ipiv_cap = (size_t)mn;
    }
    T* dA = static_cast<T*>(h->hostA_dev);
    const auto t_call = std::chrono::steady_clock::now();
    RFLU_HIP(hipMemcpy2DAsync(dA, (size_t)m * sizeof(T), A, (size_t)lda * sizeof(T), (size_t)m * sizeof(T), (size_t)n,
                              hipMemcpyHostToDevice, h->stream));
    const auto t_in = std::chrono::steady_clock::now();
    const bool want_ipiv = (ipiv != nullptr);
    // ---- the way back, overlapped -------------------------------------------------------------------------------------------
    // Rows [j0, j0 + W) of the factors are final as soon as block column b has been applied everywhere (later interchanges only
    // touch rows below), and in the row-major workspace a block of rows is one contiguous piece.  The block-column schedules report
    // how far that has got while they enqueue (Handle::progress); per chunk of rows the events of every stream are kept.  When
    // everything is enqueued (Handle::before_sync: 7 ms into an 80 ms factorization at N=16384) the calling thread -- which would only
    // wait now -- brings the chunks home on a stream of its own: wait for the chunk's events, transpose its rows into a contiguous
    // column-major piece of the staging copy, one contiguous copy into a pinned bounce buffer of the handle, and from there into the
    // caller's columns with a few host threads while the next chunk is on the link.  Chunks are 2048 rows while the factorization
    // has far to go and one block column at the end, so that little is left when the last leaf finishes.
    // Two things this needs: (1) the fourth stream on a hardware pipe of its own like the other three (validate_queues; round 3
    // first built this without and measured +26 ms instead of -30); (2) our own bounce buffers: a device-to-host copy into PAGEABLE
    // memory issued next to the running factorization returns only when the factorization has finished (the same call next to a
    // single long kernel does not wait: scripts/probes/d2h_block.hip), so the runtime's staging is of no use here.
    // RFLU_HOST_EARLY_OUT=0: the round-2 sequence (everything after the factorization).
    const int64_t chunk = h->tune.host_early_out;
    const int64_t W0 = blocksize == 0 ? default_blocksize(mn) : blocksize;
    const bool early = chunk >= 64 && !(h->prof || h->prof_one_stream) && h->num_cus == 256 && mn >= 8192 && W0 > 0 && W0 < mn;
    struct Mark { int64_t r1; std::vector<hipEvent_t> ev; };
    std::vector<Mark> marks;
    size_t ev_used = 0;
    hipStream_t C = nullptr;
    bool scattered = false;   // finished rows have reached the caller's A (they are taken back if the factorization fails after all)
    struct Reset { Handle* h; ~Reset() { h->progress = nullptr; h->before_sync = nullptr; h->out_done = false; } } reset{h};
    if (early) {
        RFLU_TRY(get_ustream(h, 96, &C));   // a masked stream = a queue of its own that validate_queues can place
        if (h->mask_failed) C = nullptr;
    }
    if (early && C) {
        const size_t bounce_bytes = (size_t)std::min(chunk, m) * (size_t)n * sizeof(T);
        // outgoing chunks are laid out column-major in a device staging area of their own (two pieces, like the bounce buffers):
        // the device copy of the INPUT (dA) stays intact until the factorization has succeeded, so that a failure reported
        // after some rows have already gone home (a panel timeout or placement error is only known at the end) can give the
        // caller its matrix back, bit for bit
        if (ensure_buffer(&h->out_stage, &h->out_stage_bytes, 2 * bounce_bytes) != RFLU_OK) C = nullptr;
    }
    if (early && C) {
        const size_t bounce_bytes = (size_t)std::min(chunk, m) * (size_t)n * sizeof(T);
        if (h->bounce_bytes < bounce_bytes) {
            for (int i = 0; i < 2; ++i) {
                if (h->bounce[i]) RFLU_HIP(hipHostFree(h->bounce[i]));
                h->bounce[i] = nullptr;
            }
            h->bounce_bytes = 0;
            bool ok = true;
            for (int i = 0; i < 2 && ok; ++i) ok = hipHostMalloc(&h->bounce[i], bounce_bytes) == hipSuccess;
            if (ok) {
                h->bounce_bytes = bounce_bytes;
            } else {   // no pinned memory to be had: the plain sequence (everything after the factorization) needs none
                (void)hipGetLastError();
                for (int i = 0; i < 2; ++i) {
                    if (h->bounce[i]) (void)hipHostFree(h->bounce[i]);
                    h->bounce[i] = nullptr;
                }
                C = nullptr;
            }
        }
    }
    if (early && C) {
        const hipStream_t user = h->stream;
        auto new_event = [h, &ev_used](hipEvent_t* e) -> int {
            if (ev_used == h->out_events.size()) {
                hipEvent_t x;
                RFLU_HIP(hipEventCreateWithFlags(&x, hipEventDisableTiming));
                h->out_events.push_back(x);
            }
            *e = h->out_events[ev_used++];
            return RFLU_OK;
        };
        auto mark_all = [h, user, new_event](std::vector<hipEvent_t>& out) -> int {
            auto rec = [&](hipStream_t st) -> int {
                hipEvent_t e;
                RFLU_TRY(new_event(&e));
                RFLU_HIP(hipEventRecord(e, st));
                out.push_back(e);
                return RFLU_OK;
            };
            RFLU_TRY(rec(user));
            if (h->stream != user) RFLU_TRY(rec(h->stream));
            for (int r = 1; r < 8; ++r) {
                if (r != 3 && h->ustreams[r]) RFLU_TRY(rec(h->ustreams[r]));
                if (h->pstreams[r] && h->pstreams[r] != h->stream) RFLU_TRY(rec(h->pstreams[r]));
            }
            return RFLU_OK;
        };
        h->progress = [&marks, mark_all, m, chunk](int64_t r) -> int {
            const int64_t have = marks.empty() ? 0 : marks.back().r1;
            r = std::min(r, m);
            // far from the end: whole chunks; within one chunk of the end: every report (one block column at a time)
            if (r <= have || (r - have < chunk && r + chunk < m)) return RFLU_OK;
            marks.push_back(Mark{std::min(r, have + chunk), {}});
            return mark_all(marks.back().ev);
        };
        h->before_sync = [&, mark_all, new_event]() -> int {
            {   // whatever is left (rows below the square part of a tall matrix included) is final when everything is: in chunk-sized pieces
                int64_t have = marks.empty() ? 0 : marks.back().r1;
                if (have < m) {
                    std::vector<hipEvent_t> all;
                    RFLU_TRY(mark_all(all));
                    while (have < m) {
                        have = std::min(m, have + chunk);
                        marks.push_back(Mark{have, all});
                    }
                }
            }
            // the way-back stream as it is NOW: validate_queues (run by the factorization, after C was first taken) may have parked the
            // stream of this mask and put a fresh one, on a pipe of its own, in its place
            RFLU_TRY(get_ustream(h, 96, &C));
            const int64_t ldr = workspace_ld(h, n);
            const T* R = static_cast<const T*>(h->work);
            const hipStream_t saved = h->stream;
            const bool trace = h->tune.host_trace != 0;
            auto since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
            if (trace) fprintf(stderr, "[rflu] host entry: enqueue done %.1f ms after the call (%.1f after the copy in), %zu chunks\n", since(t_call), since(t_in), marks.size());
            const size_t nchunks = marks.size();
            std::vector<hipEvent_t> landed(nchunks);
            std::vector<int64_t> start(nchunks);
            {
                int64_t r0 = 0;
                for (size_t k = 0; k < nchunks; ++k) { start[k] = r0; r0 = marks[k].r1; }
            }
            auto send = [&](size_t k) -> int {   // chunk k: events -> transpose into a contiguous piece of the staging copy -> bounce buffer
                const int64_t r0 = start[k], rows = marks[k].r1 - r0;
                for (hipEvent_t e : marks[k].ev)
                    if (hipStreamWaitEvent(C, e, 0) != hipSuccess) { set_error("hipStreamWaitEvent failed"); return RFLU_ERR_HIP; }
                T* piece = static_cast<T*>(h->out_stage) + (k & 1) * (size_t)std::min(chunk, m) * (size_t)n;
                h->stream = C;
                const int rc = launch_transpose<T>(h, n, rows, R + r0 * ldr, ldr, piece, rows);
                h->stream = saved;
                RFLU_TRY(rc);
                RFLU_HIP(hipMemcpyAsync(h->bounce[k & 1], piece, (size_t)rows * (size_t)n * sizeof(T), hipMemcpyDeviceToHost, C));
                RFLU_TRY(new_event(&landed[k]));
                RFLU_HIP(hipEventRecord(landed[k], C));
                return RFLU_OK;
            };
            const int nthreads = std::max(1, std::min(h->tune.host_threads, 64));
            for (size_t k = 0; k < std::min<size_t>(2, nchunks); ++k) RFLU_TRY(send(k));
            for (size_t k = 0; k < nchunks; ++k) {
                RFLU_HIP(hipEventSynchronize(landed[k]));
                const int64_t r0 = start[k], rows = marks[k].r1 - r0;
                const T* src = static_cast<const T*>(h->bounce[k & 1]);
                scattered = true;
                auto scatter = [&](int64_t j0, int64_t j1) {
                    for (int64_t j = j0; j < j1; ++j) memcpy(A + j * lda + r0, src + j * rows, (size_t)rows * sizeof(T));
                };
                if (nthreads == 1 || (size_t)rows * (size_t)n * sizeof(T) < ((size_t)8 << 20)) {
                    scatter(0, n);
                } else {
                    std::vector<std::thread> pool;
                    const int64_t per = (n + nthreads - 1) / nthreads;
                    int64_t done_to = std::min<int64_t>(n, per);   // columns [per, done_to) have a thread; no exception leaves this C entry
                    try {
                        for (int t = 1; t < nthreads; ++t) {
                            pool.emplace_back(scatter, std::min<int64_t>(n, t * per), std::min<int64_t>(n, (t + 1) * per));
                            done_to = std::min<int64_t>(n, (t + 1) * per);
                        }
                    } catch (...) {
                    }
                    scatter(0, std::min<int64_t>(n, per));
                    if (done_to < n) scatter(done_to, n);           // the threads that could not be started
                    for (std::thread& th : pool) th.join();
                }
                if (trace) fprintf(stderr, "[rflu] host entry: rows [%lld, %lld) home at %.1f ms\n", (long long)r0, (long long)marks[k].r1, since(t_call));
                if (k + 2 < nchunks) RFLU_TRY(send(k + 2));   // its bounce buffer is free again
            }
            h->out_done = true;
            return RFLU_OK;
        };
    }
    {
        const int rc = getrf_cm_dev<T>(h, m, n, dA, m, (pivot || want_ipiv) ? h->ipiv_dev : nullptr, pivot, blocksize, info);
        if (rc != RFLU_OK) {
            // A failed call leaves the caller's matrix as it was: rows that went home early are overwritten with the input again
            // (dA is only ever written after success: the final layout change of getrf_cm_dev).  The error text of rc is kept.
            if (scattered) {
                (void)hipDeviceSynchronize();
                (void)hipMemcpy2D(A, (size_t)lda * sizeof(T), dA, (size_t)m * sizeof(T), (size_t)m * sizeof(T), (size_t)n, hipMemcpyDeviceToHost);
            }
            return rc;
        }
    }
    if (!h->out_done)
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
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int dev)
    {
        err = hipGetDevice(&prev);
        if (err != hipSuccess) { prev = -1; return; }
        if (prev != dev) err = hipSetDevice(dev);
    }
    // Restores UNCONDITIONALLY: the multi-GPU entry points switch devices inside loops after the guard was taken, so "did the
    // constructor switch?" says nothing about where the current device is when the function returns (or bails out early).
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// CHECK_HANDLE declares the guard object in the enclosing scope: it must be the FIRST statement of a function body (not the
// body of an unbraced if / else, not twice in one scope).
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
#ifdef RFLU_EXPERIMENTS
        if (env_str("RFLU_NO_PRIORITY")) hi = 0;
#endif
        RFLU_HIP(hipStreamCreateWithPriority(&h->own_stream, hipStreamDefault, hi));
    }
    h->stream = h->own_stream;
    RFLU_HIP(hipMalloc((void**)&h->info_dev, 2 * sizeof(int64_t)));
    if (const char* e = env_str("RFLU_DUMMY_QUEUES")) {
        // measurement / test hook (scripts/queue_collision.py, tests/test_gpu_queues.py): k extra streams created AND USED here,
        // before the update / side streams exist, so that those get other queue indices -- and with them other hardware pipes -- than
        // in a fresh process.  Without validate_queues (RFLU_QUEUE_CHECK=0): N=16384 80 ms or 108-113 ms depending on k.
        for (int i = 0; i < atoi(e); ++i) {
            hipStream_t d;
            RFLU_HIP(hipStreamCreateWithFlags(&d, hipStreamNonBlocking));
            RFLU_HIP(hipMemsetAsync(h->info_dev, 0, 8, d));
            RFLU_HIP(hipStreamSynchronize(d));
        }
    }
    RFLU_HIP(hipMalloc((void**)&h->gates, 8 * sizeof(unsigned long long)));
    RFLU_HIP(hipMemset(h->gates, 0, 8 * sizeof(unsigned long long)));
    for (int i = 0; i < 3; ++i) h->gate_ptr[i] = h->gates + i;
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
    load_handle_env(h);
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
    if (h->out_stage) (void)hipFree(h->out_stage);
    if (h->rhs_work) (void)hipFree(h->rhs_work);
    if (h->hostB_dev) (void)hipFree(h->hostB_dev);
    if (h->pm_cnt) (void)hipFree(h->pm_cnt);
    if (h->pm_dst) (void)hipFree(h->pm_dst);
    if (h->pm_src) (void)hipFree(h->pm_src);
    if (h->linv) (void)hipFree(h->linv);
    if (h->linv_tmp) (void)hipFree(h->linv_tmp);
    if (h->pscratch) (void)hipFree(h->pscratch);
    if (h->info_dev) (void)hipFree(h->info_dev);
    if (h->gates) (void)hipFree(h->gates);
    if (h->tail_event_obj) (void)hipEventDestroy(h->tail_event_obj);
    if (h->tail_fork_obj) (void)hipEventDestroy(h->tail_fork_obj);
    if (h->gate_stamps) (void)hipFree(h->gate_stamps);
    if (h->eng_state) (void)hipFree(h->eng_state);
    if (h->eng_host) (void)hipHostFree(h->eng_host);
    if (h->eng_rows_final) (void)hipHostFree(h->eng_rows_final);
    if (h->eng_trace_buf) (void)hipFree(h->eng_trace_buf);
    if (h->info_pinned) (void)hipHostFree(h->info_pinned);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    for (hipEvent_t e : h->out_events) (void)hipEventDestroy(e);
    for (void* b : h->bounce) if (b) (void)hipHostFree(b);
    for (hipStream_t ps : h->parked_streams)
        if (ps) (void)hipStreamDestroy(ps);
    if (h->qprobe_slots) (void)hipFree(h->qprobe_slots);
    for (hipStream_t us : h->ustreams)
        if (us) (void)hipStreamDestroy(us);
    for (hipStream_t ps : h->pstreams)
        if (ps) (void)hipStreamDestroy(ps);
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

int rflu_reload_tuning(rflu_handle_t handle)
{
    if (!handle) { set_error("null handle"); return RFLU_ERR_ARG; }
    load_handle_env(H(handle));
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

// measurement only (scripts/microbench_*): `usec` of register-only MFMA load on the CU-masked update stream, asynchronously
// measurement only: copy the RFLU_GATE_TRACE stamps (3 x 4096 wall-clock ticks, 100 MHz) to the host
int rflu_debug_gate_stamps(rflu_handle_t handle, long long* out)
{
    CHECK_HANDLE(handle);
    if (!H(handle)->gate_stamps) { set_error("no gate trace (set RFLU_GATE_TRACE=1)"); return RFLU_ERR_ARG; }
    RFLU_HIP(hipMemcpy(out, H(handle)->gate_stamps, 3 * 4096 * sizeof(long long), hipMemcpyDeviceToHost));
    return RFLU_OK;
}

int rflu_debug_engine_acct(rflu_handle_t handle, long long* out8)
{
    // RFLU_ENGINE_TRACE: the workgroup-time accumulators of the LAST engine launch on this handle (100 MHz ticks, summed over the
    // workgroups): [0] block-column tiles, [1] leaf-window tiles, [2] strips + solves, [3] deferred interchanges, [4] between units,
    // [5] (of 4) asleep, [6] (of 4) publications, [7] unused
    if (!handle || !out8) return RFLU_ERR_ARG;
    Handle* h = H(handle);
    DeviceGuard device_guard__(h->device);
    for (int i = 0; i < 8; ++i) out8[i] = 0;
    if (!h->eng_trace_buf) return RFLU_OK;
    RFLU_HIP(hipStreamSynchronize(h->stream));
    RFLU_HIP(hipMemcpy(out8, h->eng_trace_buf + 4096 * 4, 8 * sizeof(long long), hipMemcpyDeviceToHost));   // (the finer split behind them: driver.cpp's own print)
    return RFLU_OK;
}

int rflu_debug_heat(rflu_handle_t handle, double usec)
{
    CHECK_HANDLE(handle);
    Handle* h = H(handle);
    hipStream_t us, saved = h->stream;
    RFLU_TRY(get_ustream(h, 32, &us));
    h->stream = us;
    const int rc = launch_heat(h, 224, usec);
    h->stream = saved;
    return rc;
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
        return gemm_public<T>(H(handle), M, N, K, A, lda, B, ldb, C, ldc);                                            \
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
    int rflu_butterfly_mul_##SFX##_dev(rflu_handle_t handle, int64_t n, T* A, int64_t lda, const T* uv)               \
    {                                                                                                                 \
        CHECK_HANDLE(handle);                                                                                         \
        if (n > 0 && (A == nullptr || uv == nullptr)) { set_error("butterfly: null pointer"); return RFLU_ERR_ARG; }  \
        return launch_butterfly_mul<T>(H(handle), n, A, lda, uv);                                                     \
    }                                                                                                                 \
    int rflu_butterfly_vec_##SFX##_dev(rflu_handle_t handle, int64_t n, int64_t nrhs, T* X, int64_t ldx, const T* uv, \
                                       int transpose_u)                                                               \
    {                                                                                                                 \
        CHECK_HANDLE(handle);                                                                                         \
        if (n > 0 && nrhs > 0 && (X == nullptr || uv == nullptr)) { set_error("butterfly: null pointer"); return RFLU_ERR_ARG; } \
        return launch_butterfly_vec<T>(H(handle), n, nrhs, X, ldx, uv, transpose_u ? 0 : 1);                          \
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

}  // extern "C" (the multi-GPU internals below are C++)

/* =====================================================================================================================
 * Multi-GPU: 1-D block-column layout over the GPUs of one node (BASELINE configs 3-4), ONE process.
 * The reference has no distributed path; this is the partition the north star asks for:
 *   - block column b (width `block`, a multiple of 64) lives on logical device (b / run) % ndev as part of that device's
 *     row-major slab (all n rows x its local columns); `run` consecutive block columns share an owner;
 *   - per block column the owner factors the tall panel with the single-GPU recursion (the same Fact::rec as
 *     rflu_panel_rm_*), packs {L\U panel rows j0.. | ipiv segment} and ONE broadcast carries it to the other devices --
 *     ncclBroadcast (RCCL over xGMI) enqueued on the library's own panel streams, no host synchronisation anywhere in the
 *     loop; every device then applies laswp -> TRSM -> GEMM to its local columns;
 *   - one block column of lookahead: the owner of b+1 updates that slice first and factors it on its panel stream while
 *     all devices still run the bulk of update b on their (CU-masked) update streams.
 * "Fake multi-GPU" (SURVEY.md 4(iii)): the same logical device list may name ONE physical device several times; the
 * broadcast then is a device-to-device copy and the whole partition / message / ordering logic runs on a single GPU --
 * what the -m gpu tests exercise with k = 2, 4, 8.
 * ===================================================================================================================== */
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace rflu {

struct Rccl {   // resolved lazily: single-GPU users of librflu.so never load RCCL
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    int load()
    {
        if (lib) return RFLU_OK;
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) { set_error("cannot load RCCL (librccl.so): %s", dlerror()); return RFLU_ERR_HIP; }
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(lib, "ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
        Broadcast = reinterpret_cast<decltype(Broadcast)>(dlsym(lib, "ncclBroadcast"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(lib, "ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
        if (!CommInitAll || !CommDestroy || !Broadcast || !GroupStart || !GroupEnd || !GetErrorString) {
            set_error("librccl.so lacks an expected symbol");
            return RFLU_ERR_HIP;
        }
        return RFLU_OK;
    }
};

#define RFLU_NCCL(r, call)                                                                          \
    do {                                                                                            \
        ncclResult_t e__ = (call);                                                                  \
        if (e__ != ncclSuccess) {                                                                   \
            set_error("%s failed: %s (%s:%d)", #call, (r).GetErrorString(e__), __FILE__, __LINE__); \
            return RFLU_ERR_HIP;                                                                    \
        }                                                                                           \
    } while (0)

struct Mgpu {
    int ndev = 0;
    bool fake = false;                 // a physical device is named more than once: broadcast = device-to-device copy
    int64_t ncoll = 0;                 // ncclBroadcast calls enqueued (rflu_mgpu_collectives)
    bool force_rccl = false;           // RFLU_MGPU_FORCE_RCCL=1 with ONE device: a one-rank communicator and the grouped broadcast-to-self, so
                                       // that the collective's code path (dlopen, ncclCommInitAll, ncclBroadcast on the panel stream) can be
                                       // executed and tested on a box with a single GPU
    std::vector<int> devs;
    std::vector<Handle*> h;            // one handle (streams, workspaces, exchange scratch) per logical device
    std::vector<ncclComm_t> comms;     // real multi-GPU only
    Rccl rccl;
    // per logical device: packed panel buffers (double-buffered by block-column parity), pivot vector, stream pair, events
    std::vector<void*> pbuf[2];
    std::vector<size_t> pbuf_bytes[2];
    std::vector<int64_t*> meta[2];     // ipiv segment of the block column (wmax entries)
    std::vector<size_t> meta_cap;
    std::vector<int64_t*> ipiv;        // full pivot vector (n) on every device
    std::vector<size_t> ipiv_cap;
    std::vector<hipStream_t> U, P;
    std::vector<std::vector<hipEvent_t>> ev;   // ev[d]: reusable, timing disabled
};

struct BlockCol { int64_t j0, w; int owner; int64_t lc; };

static void mgpu_layout(int64_t n, int64_t block, int ndev, int64_t run, std::vector<BlockCol>& out, std::vector<int64_t>& local_cols)
{
    out.clear();
    local_cols.assign(ndev, 0);
    const int64_t nb = (n + block - 1) / block;
    for (int64_t b = 0; b < nb; ++b) {
        const int64_t j0 = b * block, w = std::min(block, n - j0);
        const int owner = (int)((b / std::max<int64_t>(run, 1)) % ndev);
        out.push_back({j0, w, owner, local_cols[owner]});
        local_cols[owner] += w;
    }
}

static int mgpu_event(Mgpu* g, int d, size_t idx, hipEvent_t* out)
{
    auto& v = g->ev[d];
    while (v.size() <= idx) {
        hipEvent_t e;
        RFLU_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        v.push_back(e);
    }
    *out = v[idx];
    return RFLU_OK;
}

// apply block column (j0, w), held packed in pb (rows j0.. x w, leading dimension w), to local columns [c0, c0+ncols) of R
template <typename T>
static int mgpu_update(Handle* h, int64_t n, T* R, int64_t ld, const T* pb, int64_t j0, int64_t w, int64_t c0, int64_t ncols,
                       int pivot)
{
    if (ncols <= 0) return RFLU_OK;
    if (pivot) RFLU_TRY(launch_laswp<T>(h, R, ld, c0, ncols, j0 / NB, (j0 + w + NB - 1) / NB));
    RFLU_TRY(trsm_public<T>(h, w, ncols, pb, w, R + j0 * ld + c0, ld));
    if (n > j0 + w) RFLU_TRY(launch_gemm<T>(h, n - j0 - w, ncols, w, pb + w * w, w, R + j0 * ld + c0, ld, R + (j0 + w) * ld + c0, ld));
    return RFLU_OK;
}

// CUs the next owner of a tall panel keeps away from its share of the update (mgpu_getrf); RFLU_MGPU_BIG_RESERVE overrides
static int64_t mgpu_big_reserve(const Handle* h)
{
    return std::min<int64_t>(224, std::max<int64_t>(0, h->tune.mgpu_big_reserve / 32 * 32));
}

template <typename T>
static int mgpu_getrf(Mgpu* g, int64_t n, T* const* slabs, const int64_t* lds, int64_t* ipiv_host, int pivot, int64_t block,
                      int64_t run, int64_t* info)
{
    if (n < 0 || slabs == nullptr || lds == nullptr || info == nullptr || block <= 0 || block % NB != 0 || run <= 0 ||
        (pivot && ipiv_host == nullptr && n > 0)) {
        set_error("mgpu getrf: bad arguments (block must be a positive multiple of %d)", NB);
        return RFLU_ERR_ARG;
    }
    *info = 0;
    if (n == 0) return RFLU_OK;
    const int D = g->ndev;
    std::vector<BlockCol> lay;
    std::vector<int64_t> ncols_loc;
    mgpu_layout(n, block, D, run, lay, ncols_loc);
    const int64_t nb = (int64_t)lay.size();
    const int64_t wmax = std::min(block, n);
    DeviceGuard guard(g->devs[0]);
    RFLU_HIP(guard.err);
    for (int d = 0; d < D; ++d) {
        if (lds[d] < std::max<int64_t>(ncols_loc[d], 1) || (ncols_loc[d] > 0 && slabs[d] == nullptr)) {
            set_error("mgpu getrf: slab %d needs %lld columns (ld %lld)", d, (long long)ncols_loc[d], (long long)lds[d]);
            return RFLU_ERR_ARG;
        }
        RFLU_HIP(hipSetDevice(g->devs[d]));
        Handle* h = g->h[d];
        RFLU_TRY(ensure_bookkeeping(h, n));
        for (int par = 0; par < 2; ++par) {
            RFLU_TRY(ensure_buffer(&g->pbuf[par][d], &g->pbuf_bytes[par][d], (size_t)n * (size_t)wmax * sizeof(T)));
            if ((size_t)wmax > g->meta_cap[d]) {
                if (g->meta[par][d]) RFLU_HIP(hipFree(g->meta[par][d]));
                g->meta[par][d] = nullptr;
                RFLU_HIP(hipMalloc((void**)&g->meta[par][d], (size_t)wmax * sizeof(int64_t)));
            }
        }
        g->meta_cap[d] = std::max(g->meta_cap[d], (size_t)wmax);
        if ((size_t)n > g->ipiv_cap[d]) {
            if (g->ipiv[d]) RFLU_HIP(hipFree(g->ipiv[d]));
            g->ipiv[d] = nullptr;
            RFLU_HIP(hipMalloc((void**)&g->ipiv[d], (size_t)n * sizeof(int64_t)));
            g->ipiv_cap[d] = (size_t)n;
        }
        RFLU_TRY(get_ustream(h, 32, &g->U[d]));   // the update stream a device starts on (the mask leaves 32 CUs to the panel)
        g->P[d] = h->own_stream;
        if (!g->fake && !h->mask_failed) {
            // the streams this device will run side by side -- panel stream, 32-CU mask and (next owner of a tall panel) the big
            // mask -- on different hardware pipes (validate_queues); logical devices on one GPU share its 4 pipes anyway
            hipStream_t big;
            if (D > 1 && mgpu_big_reserve(h) > 32) RFLU_TRY(get_ustream(h, (int)mgpu_big_reserve(h), &big));
            h->stream = g->P[d];
            RFLU_TRY(validate_queues(h));
            RFLU_TRY(get_ustream(h, 32, &g->U[d]));   // may have been replaced
        }
        h->last_path = RFLU_PATH_HIP_LOOKAHEAD;
        // start state on both streams of the device
        h->stream = g->U[d];
        RFLU_HIP(hipMemsetAsync(h->info_dev, 0, 2 * sizeof(int64_t), g->U[d]));
        if (!pivot) RFLU_TRY(launch_iota_ipiv(h, g->ipiv[d], 0, n));
        hipEvent_t e0;
        RFLU_TRY(mgpu_event(g, d, 0, &e0));
        RFLU_HIP(hipEventRecord(e0, g->U[d]));
        RFLU_HIP(hipStreamWaitEvent(g->P[d], e0, 0));
    }
    // events per device: 1 + 4*b + {0: packed/received, 1: slice ready, 2: update done}
    auto EV = [&](int d, int64_t b, int k, hipEvent_t* e) { return mgpu_event(g, d, (size_t)(1 + 4 * b + k), e); };
    const ncclDataType_t ntype = sizeof(T) == 8 ? ncclDouble : ncclFloat;

    // factor + pack block column b on its owner's panel stream, then carry it to every other device
    auto produce = [&](int64_t b) -> int {
        const BlockCol& c = lay[b];
        const int o = c.owner;
        const int par = (int)(b & 1);
        const int64_t rows = n - c.j0;
        hipEvent_t e;
        for (int d = 0; d < D; ++d) {   // the parity buffer was last read by update b-2 on every device
            if (b >= 2) {
                RFLU_HIP(hipSetDevice(g->devs[d]));
                RFLU_TRY(EV(d, b - 2, 2, &e));
                RFLU_HIP(hipStreamWaitEvent(g->P[d], e, 0));
                if (g->fake) {
                    // without a collective the receivers of b-2 copied straight out of ITS owner's buffer, each at its own
                    // pace: nobody may overwrite a parity buffer before every logical device has block column b-2
                    for (int d2 = 0; d2 < D; ++d2) {
                        if (d2 == d) continue;
                        RFLU_TRY(EV(d2, b - 2, 0, &e));
                        RFLU_HIP(hipStreamWaitEvent(g->P[d], e, 0));
                    }
                }
            }
        }
        {
            RFLU_HIP(hipSetDevice(g->devs[o]));
            Handle* h = g->h[o];
            if (b >= 1) {   // the owner's slice has received update b-1
                RFLU_TRY(EV(o, b - 1, 1, &e));
                RFLU_HIP(hipStreamWaitEvent(g->P[o], e, 0));
            }
            h->stream = g->P[o];
            T* R = slabs[o];
            Fact<T> f{h, R, lds[o], n, c.lc + c.w, g->ipiv[o], pivot};
            f.sw_lo = c.lc;
            f.sw_hi = c.lc + c.w;
            f.roff = c.j0 - c.lc;
            RFLU_TRY(f.rec(c.lc, c.lc + c.w));
            RFLU_HIP(hipMemcpy2DAsync(g->pbuf[par][o], (size_t)c.w * sizeof(T), R + c.j0 * lds[o] + c.lc, (size_t)lds[o] * sizeof(T),
                                      (size_t)c.w * sizeof(T), (size_t)rows, hipMemcpyDeviceToDevice, g->P[o]));
            RFLU_HIP(hipMemcpyAsync(g->meta[par][o], g->ipiv[o] + c.j0, (size_t)c.w * sizeof(int64_t), hipMemcpyDeviceToDevice, g->P[o]));
            RFLU_TRY(EV(o, b, 0, &e));
            RFLU_HIP(hipEventRecord(e, g->P[o]));
        }
        if (D == 1 && !g->force_rccl) return RFLU_OK;
        if (!g->fake) {   // the one exchange step of the path: ncclBroadcast of {panel, pivots} on the panel streams
            RFLU_NCCL(g->rccl, g->rccl.GroupStart());
            for (int d = 0; d < D; ++d) {
                RFLU_NCCL(g->rccl, g->rccl.Broadcast(g->pbuf[par][d], g->pbuf[par][d], (size_t)rows * (size_t)c.w, ntype, o, g->comms[d], g->P[d]));
                RFLU_NCCL(g->rccl, g->rccl.Broadcast(g->meta[par][d], g->meta[par][d], (size_t)c.w, ncclInt64, o, g->comms[d], g->P[d]));
                g->ncoll += 2;
            }
            RFLU_NCCL(g->rccl, g->rccl.GroupEnd());
            for (int d = 0; d < D; ++d) {
                if (d == o) continue;
                RFLU_HIP(hipSetDevice(g->devs[d]));
                RFLU_TRY(EV(d, b, 0, &e));
                RFLU_HIP(hipEventRecord(e, g->P[d]));
            }
        } else {          // logical devices on one physical device: the "broadcast" is a device-to-device copy
            hipEvent_t packed;
            RFLU_TRY(EV(o, b, 0, &packed));
            for (int d = 0; d < D; ++d) {
                if (d == o) continue;
                RFLU_HIP(hipSetDevice(g->devs[d]));
                RFLU_HIP(hipStreamWaitEvent(g->P[d], packed, 0));
                RFLU_HIP(hipMemcpyAsync(g->pbuf[par][d], g->pbuf[par][o], (size_t)rows * (size_t)c.w * sizeof(T), hipMemcpyDeviceToDevice, g->P[d]));
                RFLU_HIP(hipMemcpyAsync(g->meta[par][d], g->meta[par][o], (size_t)c.w * sizeof(int64_t), hipMemcpyDeviceToDevice, g->P[d]));
                RFLU_TRY(EV(d, b, 0, &e));
                RFLU_HIP(hipEventRecord(e, g->P[d]));
            }
        }
        return RFLU_OK;
    };

    RFLU_TRY(produce(0));
    // The NEXT owner factors block column b+1 next to its own share of update b, which is only 1/D of the bulk: it can afford
    // to leave the panel as many CUs as its cooperating workgroups need.  Up to 32 workgroups (16384 rows): the usual 32-CU
    // reservation; up to `big_reserve` CUs (default 128 = 65536 rows): that device runs update b on the update stream of the big
    // reservation (ONE extra stream per device: panel stream, 32-CU mask, big mask -- three streams on three of the four hardware
    // pipes, placed by validate_queues).  Only a panel that needs even more is factored BEFORE the owner's bulk update (round 2 did that from 16384
    // rows on: at N=65536 over 8 GPUs 96 of 128 block columns, ~0.3 s of un-overlapped panels).
    const Tune& tune0 = g->h[0]->tune;
    const int64_t big_reserve = mgpu_big_reserve(g->h[0]);
    // (one device has no big-reserve stream -- it is only created for D > 1: its panels taller than the usual 32-CU reservation
    //  holds are factored before the bulk update, as in round 2)
    int64_t tall_rows = (D > 1 ? std::max<int64_t>(32, big_reserve) : 32) * (int64_t)PANEL_THREADS;
    if (tune0.mgpu_tall_rows >= 0) tall_rows = tune0.mgpu_tall_rows;     // debugging knobs
    std::vector<hipStream_t> Ucur(g->U);   // the stream that carried each device's previous update
    const int dbg_sync = tune0.mgpu_sync;
    auto sync_all = [&]() -> int {
        for (int d = 0; d < D; ++d) { RFLU_HIP(hipSetDevice(g->devs[d])); RFLU_HIP(hipDeviceSynchronize()); }
        return RFLU_OK;
    };
    for (int64_t b = 0; b < nb; ++b) {
        if (dbg_sync & 1) RFLU_TRY(sync_all());
        const BlockCol& c = lay[b];
        const int par = (int)(b & 1);
        const int nxt_owner = (b + 1 < nb) ? lay[b + 1].owner : -1;
        const bool tall_next = nxt_owner >= 0 && (n - lay[b + 1].j0) > tall_rows;
        std::vector<int64_t> left_end(D, 0), right_start(D, 0), nxt_slice(D, 0);
        hipEvent_t e;
        // the update stream of every device for this block column: the next owner leaves its panel the CUs it needs
        std::vector<hipStream_t> Ub(g->U);
        if (nxt_owner >= 0 && !tall_next && D > 1) {
            const int64_t gw = panel_wgs(g->h[nxt_owner], n - lay[b + 1].j0, pivot, sizeof(T));
            if (gw > 32 && big_reserve > 32) {
                RFLU_HIP(hipSetDevice(g->devs[nxt_owner]));
                RFLU_TRY(get_ustream(g->h[nxt_owner], (int)big_reserve, &Ub[nxt_owner]));
            }
        }
        for (int d = 0; d < D; ++d) {
            if (Ub[d] == Ucur[d]) continue;   // a different mask = a different stream: order it behind the device's last update
            RFLU_HIP(hipSetDevice(g->devs[d]));
            if (b >= 1) RFLU_TRY(EV(d, b - 1, 2, &e)); else RFLU_TRY(mgpu_event(g, d, 0, &e));
            RFLU_HIP(hipStreamWaitEvent(Ub[d], e, 0));
            Ucur[d] = Ub[d];
        }
        // ---- phase 1: receive, pivots, and the slice of the next owner
        for (int d = 0; d < D; ++d) {
            RFLU_HIP(hipSetDevice(g->devs[d]));
            Handle* h = g->h[d];
            h->stream = Ub[d];
            RFLU_TRY(EV(d, b, 0, &e));                       // packed (owner) / received (others)
            RFLU_HIP(hipStreamWaitEvent(Ub[d], e, 0));
            const T* pb = static_cast<const T*>(g->pbuf[par][d]);
            if (d != c.owner)
                RFLU_HIP(hipMemcpyAsync(g->ipiv[d] + c.j0, g->meta[par][d], (size_t)c.w * sizeof(int64_t), hipMemcpyDeviceToDevice, Ub[d]));
            if (pivot) RFLU_TRY(launch_perm_build(h, g->ipiv[d], c.j0, c.j0 + c.w, n));
            for (int64_t q = 0; q < b; ++q) if (lay[q].owner == d) left_end[d] += lay[q].w;
            right_start[d] = left_end[d] + (d == c.owner ? c.w : 0);
            if (d == nxt_owner) {   // this device owns block column b+1: bring exactly those columns up to date first
                nxt_slice[d] = lay[b + 1].w;
                RFLU_TRY(mgpu_update<T>(h, n, slabs[d], lds[d], pb, c.j0, c.w, right_start[d], nxt_slice[d], pivot));
            }
            RFLU_TRY(EV(d, b, 1, &e));
            RFLU_HIP(hipEventRecord(e, Ub[d]));
        }
        if (dbg_sync & 2) RFLU_TRY(sync_all());
        if (tall_next) {
            RFLU_TRY(produce(b + 1));
            RFLU_HIP(hipSetDevice(g->devs[nxt_owner]));
            RFLU_TRY(EV(nxt_owner, b + 1, 0, &e));
            RFLU_HIP(hipStreamWaitEvent(Ub[nxt_owner], e, 0));
        }
        // ---- phase 2: interchanges on the finished columns to the left, bulk update of the rest
        for (int d = 0; d < D; ++d) {
            RFLU_HIP(hipSetDevice(g->devs[d]));
            Handle* h = g->h[d];
            h->stream = Ub[d];
            const T* pb = static_cast<const T*>(g->pbuf[par][d]);
            if (pivot && left_end[d] > 0)
                RFLU_TRY(launch_laswp<T>(h, slabs[d], lds[d], 0, left_end[d], c.j0 / NB, (c.j0 + c.w + NB - 1) / NB));
            RFLU_TRY(mgpu_update<T>(h, n, slabs[d], lds[d], pb, c.j0, c.w, right_start[d] + nxt_slice[d],
                                    ncols_loc[d] - right_start[d] - nxt_slice[d], pivot));
            RFLU_TRY(EV(d, b, 2, &e));
            RFLU_HIP(hipEventRecord(e, Ub[d]));
        }
        if (b + 1 < nb && !tall_next) RFLU_TRY(produce(b + 1));   // queued behind the slice update: overlaps with the bulk of update b
    }
    // drain, collect info (first zero pivot = smallest global index among the owners) and the error flags
    int64_t first = 0, flags = 0;
    for (int d = 0; d < D; ++d) {
        RFLU_HIP(hipSetDevice(g->devs[d]));
        Handle* h = g->h[d];
        RFLU_HIP(hipStreamSynchronize(g->P[d]));
        RFLU_HIP(hipStreamSynchronize(g->U[d]));
        if (Ucur[d] != g->U[d]) RFLU_HIP(hipStreamSynchronize(Ucur[d]));
        RFLU_HIP(hipMemcpy(h->info_pinned, h->info_dev, 2 * sizeof(int64_t), hipMemcpyDeviceToHost));
        if (h->info_pinned[0] != 0 && (first == 0 || h->info_pinned[0] < first)) first = h->info_pinned[0];
        flags |= h->info_pinned[1];
        h->stream = h->own_stream;
    }
    g->h[0]->info_pinned[1] = flags;
    RFLU_TRY(panel_flags_status(g->h[0]));
    if (ipiv_host) {
        RFLU_HIP(hipSetDevice(g->devs[0]));
        RFLU_HIP(hipMemcpy(ipiv_host, g->ipiv[0], (size_t)n * sizeof(int64_t), hipMemcpyDeviceToHost));
    }
    *info = first;
    return RFLU_OK;
}

template <typename T>
static int mgpu_fill(Mgpu* g, int64_t n, T* const* slabs, const int64_t* lds, int64_t block, int64_t run, uint64_t seed,
                     double diag_add)
{
    if (n <= 0 || block <= 0 || run <= 0) { set_error("mgpu fill: bad arguments"); return RFLU_ERR_ARG; }
    std::vector<BlockCol> lay;
    std::vector<int64_t> ncols_loc;
    mgpu_layout(n, block, g->ndev, run, lay, ncols_loc);
    DeviceGuard guard(g->devs[0]);
    RFLU_HIP(guard.err);
    for (const BlockCol& c : lay) {
        RFLU_HIP(hipSetDevice(g->devs[c.owner]));
        Handle* h = g->h[c.owner];
        h->stream = h->own_stream;
        RFLU_TRY(launch_fill_uniform<T>(h, slabs[c.owner] + c.lc, n, c.w, lds[c.owner], 1, seed, n, 0, c.j0, diag_add));
    }
    for (int d = 0; d < g->ndev; ++d) {
        RFLU_HIP(hipSetDevice(g->devs[d]));
        RFLU_HIP(hipStreamSynchronize(g->h[d]->own_stream));
    }
    return RFLU_OK;
}

}  // namespace rflu

extern "C" {

static Mgpu* MG(rflu_mgpu_t m) { return reinterpret_cast<Mgpu*>(m); }

int rflu_mgpu_create(rflu_mgpu_t* out, int ndev, const int* devs)
{
    if (out == nullptr || ndev < 1 || ndev > 64 || devs == nullptr) { set_error("mgpu create: bad arguments"); return RFLU_ERR_ARG; }
    *out = nullptr;
    Mgpu* g = new (std::nothrow) Mgpu();
    if (!g) { set_error("out of host memory"); return RFLU_ERR_ARG; }
    g->ndev = ndev;
    g->devs.assign(devs, devs + ndev);
    for (int i = 0; i < ndev; ++i)
        for (int j = 0; j < i; ++j)
            if (devs[i] == devs[j]) g->fake = true;
    for (int par = 0; par < 2; ++par) {
        g->pbuf[par].assign(ndev, nullptr);
        g->pbuf_bytes[par].assign(ndev, 0);
        g->meta[par].assign(ndev, nullptr);
    }
    g->meta_cap.assign(ndev, 0);
    g->ipiv.assign(ndev, nullptr);
    g->ipiv_cap.assign(ndev, 0);
    g->U.assign(ndev, nullptr);
    g->P.assign(ndev, nullptr);
    g->ev.resize(ndev);
    int rc = RFLU_OK;
    for (int d = 0; d < ndev && rc == RFLU_OK; ++d) {
        rflu_handle_t hh = nullptr;
        rc = rflu_create(&hh, devs[d]);
        if (rc == RFLU_OK) g->h.push_back(H(hh));
    }
    if (const char* e = env_str("RFLU_MGPU_FORCE_RCCL")) g->force_rccl = atoi(e) != 0 && ndev == 1;
    if (rc == RFLU_OK && (ndev > 1 || g->force_rccl) && !g->fake) {   // RCCL communicator over the distinct devices (single process)
        rc = g->rccl.load();
        if (rc == RFLU_OK) {
            g->comms.assign(ndev, nullptr);
            ncclResult_t e = g->rccl.CommInitAll(g->comms.data(), ndev, devs);
            if (e != ncclSuccess) { set_error("ncclCommInitAll failed: %s", g->rccl.GetErrorString(e)); g->comms.clear(); rc = RFLU_ERR_HIP; }
        }
    }
    if (rc != RFLU_OK) { (void)rflu_mgpu_destroy(reinterpret_cast<rflu_mgpu_t>(g)); return rc; }
    *out = reinterpret_cast<rflu_mgpu_t>(g);
    return RFLU_OK;
}

int rflu_mgpu_destroy(rflu_mgpu_t m)
{
    if (!m) return RFLU_OK;
    Mgpu* g = MG(m);
    int prev = 0;
    (void)hipGetDevice(&prev);
    for (size_t d = 0; d < g->h.size(); ++d) {
        (void)hipSetDevice(g->devs[d]);
        (void)hipDeviceSynchronize();
        for (int par = 0; par < 2; ++par) {
            if (g->pbuf[par][d]) (void)hipFree(g->pbuf[par][d]);
            if (g->meta[par][d]) (void)hipFree(g->meta[par][d]);
        }
        if (g->ipiv[d]) (void)hipFree(g->ipiv[d]);
        for (hipEvent_t e : g->ev[d]) (void)hipEventDestroy(e);
    }
    for (ncclComm_t c : g->comms)
        if (c) (void)g->rccl.CommDestroy(c);
    for (Handle* h : g->h) (void)rflu_destroy(reinterpret_cast<rflu_handle_t>(h));
    (void)hipSetDevice(prev);
    delete g;
    return RFLU_OK;
}

int rflu_mgpu_ndev(rflu_mgpu_t m) { return m ? MG(m)->ndev : 0; }
int rflu_mgpu_is_fake(rflu_mgpu_t m) { return m ? (MG(m)->fake ? 1 : 0) : 0; }
int64_t rflu_mgpu_collectives(rflu_mgpu_t m) { return m ? MG(m)->ncoll : 0; }

int rflu_mgpu_reload_tuning(rflu_mgpu_t m)
{
    if (!m) { set_error("null multi-GPU object"); return RFLU_ERR_ARG; }
    for (Handle* h : MG(m)->h) load_handle_env(h);   // the per-device handles read RFLU_* again, like rflu_reload_tuning
    return RFLU_OK;
}

int64_t rflu_mgpu_local_cols(int64_t n, int64_t block, int ndev, int64_t run, int d)
{
    if (n < 0 || block <= 0 || ndev < 1 || run < 1 || d < 0 || d >= ndev) return -1;
    std::vector<BlockCol> lay;
    std::vector<int64_t> loc;
    mgpu_layout(n, block, ndev, run, lay, loc);
    return loc[d];
}

int rflu_getrf_f64_mgpu(rflu_mgpu_t m, int64_t n, double* const* slabs, const int64_t* lds, int64_t* ipiv_host, int pivot,
                        int64_t block, int64_t run, int64_t* info)
{
    if (!m) { set_error("null multi-GPU handle"); return RFLU_ERR_ARG; }
    return mgpu_getrf<double>(MG(m), n, slabs, lds, ipiv_host, pivot, block, run, info);
}
int rflu_getrf_f32_mgpu(rflu_mgpu_t m, int64_t n, float* const* slabs, const int64_t* lds, int64_t* ipiv_host, int pivot,
                        int64_t block, int64_t run, int64_t* info)
{
    if (!m) { set_error("null multi-GPU handle"); return RFLU_ERR_ARG; }
    return mgpu_getrf<float>(MG(m), n, slabs, lds, ipiv_host, pivot, block, run, info);
}
int rflu_mgpu_fill_uniform_f64(rflu_mgpu_t m, int64_t n, double* const* slabs, const int64_t* lds, int64_t block, int64_t run,
                               uint64_t seed, double diag_add)
{
    if (!m) { set_error("null multi-GPU handle"); return RFLU_ERR_ARG; }
    return mgpu_fill<double>(MG(m), n, slabs, lds, block, run, seed, diag_add);
}
int rflu_mgpu_fill_uniform_f32(rflu_mgpu_t m, int64_t n, float* const* slabs, const int64_t* lds, int64_t block, int64_t run,
                               uint64_t seed, double diag_add)
{
    if (!m) { set_error("null multi-GPU handle"); return RFLU_ERR_ARG; }
    return mgpu_fill<float>(MG(m), n, slabs, lds, block, run, seed, diag_add);
}

int rflu_profile_enable(rflu_handle_t handle, int enable)
{
    CHECK_HANDLE(handle);
    Handle* h = H(handle);
    h->prof = enable == 1;
    h->prof_async = enable == 2 || enable == 3;
    h->prof_one_stream = enable == 3;
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
