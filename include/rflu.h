/*
 * rflu.h -- C ABI of librflu.so: MI355X-native (gfx950, hand-written HIP) recursive LU with partial pivoting.
 *
 * Drop-in boundary for RecursiveFactorization.jl's hot path (citations into /root/reference/):
 *   lu!(A, ipiv, pivot, thread; check, blocksize, threshold) -> LU(A, ipiv, info)     src/lu.jl:97-130
 *   recurse!/_recurse!/reckernel! and their four kernels                              src/lu.jl:132-338
 * The reference is pure Julia and has no FFI today; these entry points are what a `ccall` from its `lu!` method
 * binds (see INTEGRATION.md for the Julia stub).  Conventions follow what LinearAlgebra.LU / LAPACK getrf expect:
 *   - matrices are column-major with leading dimension lda >= m; factors overwrite A (strict lower = L with an
 *     implicit unit diagonal, upper = U);
 *   - ipiv is int64 (Julia BlasInt on ILP64), 1-based, sequential row interchanges, GLOBAL indices
 *     (the reference makes them global through `P2 .+= n1`, src/lu.jl:256-260);
 *   - *info = 0 or the 1-based index of the first exactly-zero pivot, POSITIVE convention; the factorization
 *     continues past it (src/lu.jl:321-334).  The Julia>=1.11 sign flip for NoPivot (src/lu.jl:25,250,324) and
 *     `check` -> SingularException (src/lu.jl:128) are applied by the host glue, not here;
 *   - pivot != 0 is Val(true)/RowMaximum(), pivot == 0 is Val(false)/NoPivot(); with pivot == 0 a non-NULL ipiv is
 *     filled with the identity 1..min(m,n) (src/lu.jl:111-113), NULL plays the role of NotIPIV (src/lu.jl:27-40).
 * Return value of every function: 0 on success, non-zero rflu_status on a runtime (HIP / argument) failure --
 * never a numerical condition.  rflu_last_error() returns a thread-local description of the last failure.
 * A handle owns one device, one stream and its workspaces; one handle is not thread-safe, distinct handles are.
 * No function falls back to a CPU implementation.
 */
#ifndef RFLU_H
#define RFLU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rflu_handle_s* rflu_handle_t;

enum rflu_status {
    RFLU_OK = 0,
    RFLU_ERR_ARG = 1,      /* bad argument (negative size, lda < m, NULL pointer, unsupported size) */
    RFLU_ERR_HIP = 2,      /* a HIP runtime call failed */
    RFLU_ERR_TIMEOUT = 3,  /* the cooperative panel kernel gave up waiting for a peer workgroup */
    RFLU_ERR_NODEVICE = 4, /* no usable gfx950 device */
    RFLU_ERR_PLACEMENT = 5 /* a workgroup of the XCD-local panel kernel ran on an unexpected XCD (results discarded) */
};

/* rflu_last_path values: which implementation served the last getrf call on this handle.  The analogue of the
 * reference's dispatch-routing tests (test/runtests.jl:86-114,162-192): tests assert the HIP path really ran. */
enum rflu_path {
    RFLU_PATH_NONE = 0,
    RFLU_PATH_HIP_RECURSIVE = 1, /* pure Toledo recursion on one stream */
    RFLU_PATH_HIP_BLOCKED = 2,   /* right-looking block columns on one stream (profiling modes, devices without 256 CUs) */
    RFLU_PATH_HIP_LOOKAHEAD = 3, /* block-column lookahead / leaf-wise schedules on CU-masked streams */
    RFLU_PATH_HIP_ENGINE = 4     /* the leaf-wise chain with every trailing update pulled by the persistent update engine (csrc/engine.hip) */
};

/* kernel classes for the built-in per-kernel timers (rflu_profile_*) */
enum rflu_kclass {
    RFLU_K_GEMM = 0,      /* schur_complement!  C -= A*B      (src/lu.jl:265-284), MFMA */
    RFLU_K_TRSM = 1,      /* ldiv!(UnitLowerTriangular(A11), A12) base blocks (src/lu.jl:235) */
    RFLU_K_LASWP = 2,     /* apply_permutation! (src/lu.jl:164-188) */
    RFLU_K_PANEL = 3,     /* _generic_lufact!   (src/lu.jl:290-338) cooperative leaf panel */
    RFLU_K_TRANSPOSE = 4, /* column-major <-> internal row-major layout change at the boundary */
    RFLU_K_MISC = 5,      /* pivot bookkeeping, fills */
    RFLU_K_GEMM_SMALL = 6, /* the same update for K < 256: per-leaf updates and small merges (latency / HBM bound) */
    RFLU_K_LASWP_WIDE = 7, /* apply_permutation! launches that move >= 32 MiB (trailing-update / finished-columns interchanges:
                              the bandwidth-bound share; the per-leaf launches stay in RFLU_K_LASWP and are latency-bound) */
    RFLU_K_COUNT = 8
};

/* ---- lifetime ---- */
int rflu_create(rflu_handle_t* handle, int device);
int rflu_destroy(rflu_handle_t handle);
/* The library's RFLU_* tuning / debugging variables (INTEGRATION.md lists them) are read from the environment once, by
 * rflu_create; this reads them again for an existing handle (a host that changes them between calls: the test-suite). */
int rflu_reload_tuning(rflu_handle_t handle);
const char* rflu_last_error(void);
int rflu_version(void);
/* Use an existing HIP stream (hipStream_t passed as void*); NULL restores the handle's own stream. */
int rflu_set_stream(rflu_handle_t handle, void* hip_stream);
int rflu_synchronize(rflu_handle_t handle);
int rflu_last_path(rflu_handle_t handle);
/* The handle's second stream (hipStream_t as void*): restricted by a CU mask to 224 of the 256 CUs so that cooperative
 * panel kernels issued on another stream always find 32 free CUs.  Used by the lookahead drivers (single- and
 * multi-GPU) for the bulk trailing updates. */
int rflu_update_stream(rflu_handle_t handle, void** hip_stream_out);
/* Measurement aid (scripts/microbench_*.py): about `usec` microseconds of register-only MFMA load, launched asynchronously on
 * the CU-masked update stream.  Not part of the reference interface. */
int rflu_debug_heat(rflu_handle_t handle, double usec);
/* Measurement aid (scripts/gate_trace.py): with RFLU_GATE_TRACE=1 in the environment the leaf-wise schedule stamps the wall
 * clock (100 MHz ticks) when each stream passes each leaf; copies the 3 x 4096 stamps to `out` (host). */
int rflu_debug_gate_stamps(rflu_handle_t handle, long long* out);
/* RFLU_ENGINE_TRACE=1 (measurement): workgroup-time accumulators of the last launch of the persistent update engine on this handle, in
 * 100 MHz ticks summed over its workgroups: out8[0] block-column tiles, [1] leaf-window tiles, [2] interchange strips + block-row solves,
 * [3] deferred interchanges on finished columns, [4] everything between two units, [5] (part of 4) asleep, [6] (part of 4) publications */
int rflu_debug_engine_acct(rflu_handle_t handle, long long* out8);

/* ---- the boundary: lu!(A, ipiv, pivot; blocksize) on HOST buffers (caller-owned, column-major) ----
 * Replaces src/lu.jl:114-126 (recursive path + unblocked fallback) for Float64 / Float32.
 * blocksize: < 0 = pure Toledo recursion on the whole matrix, one stream (the reference's structure, src/lu.jl:189-263);
 *            > 0 = width of the outer right-looking block column (`blocksize` re-read as the GPU panel
 *                  width, BASELINE config 2 sweeps 64/128/256; rounded up to a multiple of 64).  Tall, update-bound block
 *                  columns: factored by the same recursion with one block column of lookahead (the next block column is
 *                  updated and factored while the rest of the trailing update still runs on a second stream); from the first
 *                  panel of at most 8192 rows (Float32: 16384) on, for widths 128..512: leaf by leaf, right-looking, with only
 *                  the next leaf's 64 columns on the critical path;
 *            = 0 = library default, measured on MI355X: pure recursion below 1024 columns, then block columns of
 *                  256 (<= 11264 columns), 512 (<= 16384: pivoted, not fat and at most 16384 rows these go through the persistent
 *                  update engine, RFLU_PATH_HIP_ENGINE), 1024 (<= 24576), 2048 above. */
int rflu_getrf_f64(rflu_handle_t handle, int64_t m, int64_t n, double* A_host, int64_t lda, int64_t* ipiv_host,
                   int pivot, int64_t blocksize, int64_t* info);
int rflu_getrf_f32(rflu_handle_t handle, int64_t m, int64_t n, float* A_host, int64_t lda, int64_t* ipiv_host,
                   int pivot, int64_t blocksize, int64_t* info);

/* ---- same, DEVICE-resident (A_dev, ipiv_dev are device pointers on the handle's device; info is a host pointer).
 * Column-major in, column-major out; asynchronous work is completed before return. */
int rflu_getrf_f64_dev(rflu_handle_t handle, int64_t m, int64_t n, double* A_dev, int64_t lda, int64_t* ipiv_dev,
                       int pivot, int64_t blocksize, int64_t* info);
int rflu_getrf_f32_dev(rflu_handle_t handle, int64_t m, int64_t n, float* A_dev, int64_t lda, int64_t* ipiv_dev,
                       int pivot, int64_t blocksize, int64_t* info);

/* ---- the solve step that follows the path in every caller: ldiv!(F::LU, B), i.e. B <- U^-1 L^-1 P B ----
 * Reference: stdlib ldiv!(::LU) on the object lu! returns (LinearSolve's solve!), and the package's own ldiv! for
 * NotIPIV factors (src/lu.jl:60-64).  F/ipiv exactly as rflu_getrf_* left them (column-major packed L\U, 1-based
 * ipiv; ipiv == NULL plays NotIPIV); B is n x nrhs column-major, overwritten with the solution.  A singular U
 * (info != 0) yields Inf/NaN like LAPACK getrs; the host glue checks info first. */
int rflu_getrs_f64(rflu_handle_t handle, int64_t n, int64_t nrhs, const double* F_host, int64_t lda,
                   const int64_t* ipiv_host, double* B_host, int64_t ldb);
int rflu_getrs_f32(rflu_handle_t handle, int64_t n, int64_t nrhs, const float* F_host, int64_t lda,
                   const int64_t* ipiv_host, float* B_host, int64_t ldb);
int rflu_getrs_f64_dev(rflu_handle_t handle, int64_t n, int64_t nrhs, const double* F_dev, int64_t lda,
                       const int64_t* ipiv_dev, double* B_dev, int64_t ldb);
int rflu_getrs_f32_dev(rflu_handle_t handle, int64_t n, int64_t nrhs, const float* F_dev, int64_t lda,
                       const int64_t* ipiv_dev, float* B_dev, int64_t ldb);
/* same on row-major device data (F as left by rflu_getrf_rm_*; B is n x nrhs row-major with leading dimension ldb) */
int rflu_getrs_rm_f64_dev(rflu_handle_t handle, int64_t n, int64_t nrhs, const double* R_dev, int64_t ld,
                          const int64_t* ipiv_dev, double* B_dev, int64_t ldb);
int rflu_getrs_rm_f32_dev(rflu_handle_t handle, int64_t n, int64_t nrhs, const float* R_dev, int64_t ld,
                          const int64_t* ipiv_dev, float* B_dev, int64_t ldb);

/* ---- building blocks on the INTERNAL row-major layout: element (i,j) at R[i*ld + j] (device pointers).
 * These are the four kernels of the path plus the bookkeeping the multi-GPU block-column driver and the parity
 * tests need.  Pivot rows are GLOBAL 0-based row positions r0.. of the slab; ipiv entries are 1-based rows.
 * rflu_getrf_rm_*: factor the m x n row-major matrix in place (diagonal at (0,0)); ipiv_dev length min(m,n). */
int rflu_getrf_rm_f64_dev(rflu_handle_t handle, int64_t m, int64_t n, double* R_dev, int64_t ld, int64_t* ipiv_dev,
                          int pivot, int64_t blocksize, int64_t* info);
int rflu_getrf_rm_f32_dev(rflu_handle_t handle, int64_t m, int64_t n, float* R_dev, int64_t ld, int64_t* ipiv_dev,
                          int pivot, int64_t blocksize, int64_t* info);
/* Factor the tall panel rows [r0, m) x columns [c0, c0+w) whose diagonal block starts at (r0, c0); writes
 * ipiv_dev[r0 .. r0+w) (1-based global rows) and the row-interchange bookkeeping used by rflu_laswp_rm_*.
 * r0 must be a multiple of 64.  *info (host) gets 0 or r0 + k + 1 of the first zero pivot. */
int rflu_panel_rm_f64_dev(rflu_handle_t handle, int64_t m, int64_t r0, int64_t c0, int64_t w, double* R_dev,
                          int64_t ld, int64_t* ipiv_dev, int pivot, int64_t* info);
int rflu_panel_rm_f32_dev(rflu_handle_t handle, int64_t m, int64_t r0, int64_t c0, int64_t w, float* R_dev,
                          int64_t ld, int64_t* ipiv_dev, int pivot, int64_t* info);
/* apply_permutation!: apply the interchanges ipiv_dev[k0 .. k1) (row k <-> ipiv[k]-1, in order) to columns
 * [c0, c0+ncols) of R.  k0 must be a multiple of 64.  m = number of rows of R (bounds for the bookkeeping). */
int rflu_laswp_rm_f64_dev(rflu_handle_t handle, double* R_dev, int64_t ld, int64_t m, int64_t c0, int64_t ncols,
                          const int64_t* ipiv_dev, int64_t k0, int64_t k1);
int rflu_laswp_rm_f32_dev(rflu_handle_t handle, float* R_dev, int64_t ld, int64_t m, int64_t c0, int64_t ncols,
                          const int64_t* ipiv_dev, int64_t k0, int64_t k1);
/* B <- L^-1 B, L = unit lower triangle of the n x n block L_dev (row-major, ldl); B is n x nrhs (row-major, ldb). */
int rflu_trsm_rm_f64_dev(rflu_handle_t handle, int64_t n, int64_t nrhs, const double* L_dev, int64_t ldl,
                         double* B_dev, int64_t ldb);
int rflu_trsm_rm_f32_dev(rflu_handle_t handle, int64_t n, int64_t nrhs, const float* L_dev, int64_t ldl, float* B_dev,
                         int64_t ldb);
/* C <- C - A*B, all row-major: A is M x K (lda), B is K x N (ldb), C is M x N (ldc). */
int rflu_gemm_rm_f64_dev(rflu_handle_t handle, int64_t M, int64_t N, int64_t K, const double* A_dev, int64_t lda,
                         const double* B_dev, int64_t ldb, double* C_dev, int64_t ldc);
int rflu_gemm_rm_f32_dev(rflu_handle_t handle, int64_t M, int64_t N, int64_t K, const float* A_dev, int64_t lda,
                         const float* B_dev, int64_t ldb, float* C_dev, int64_t ldc);
/* Layout change: column-major (m x n, lda) <-> row-major (m x n, ldr). */
int rflu_cm_to_rm_f64_dev(rflu_handle_t handle, int64_t m, int64_t n, const double* A_cm, int64_t lda, double* R_rm,
                          int64_t ldr);
int rflu_rm_to_cm_f64_dev(rflu_handle_t handle, int64_t m, int64_t n, const double* R_rm, int64_t ldr, double* A_cm,
                          int64_t lda);
int rflu_cm_to_rm_f32_dev(rflu_handle_t handle, int64_t m, int64_t n, const float* A_cm, int64_t lda, float* R_rm,
                          int64_t ldr);
int rflu_rm_to_cm_f32_dev(rflu_handle_t handle, int64_t m, int64_t n, const float* R_rm, int64_t ldr, float* A_cm,
                          int64_t lda);

/* ---- randomized butterfly pre-transform (the reference's 🦋 solver, src/butterflylu.jl) ----
 * rflu_butterfly_mul_*: A <- U' A V in place (🦋mul!, src/butterflylu.jl:90-113), A column-major n x n on the device, n % 4
 * == 0 (pad! first, :180-197), uv = the 4n random diagonal entries in the reference's layout (:93-108).  One streaming
 * pass applies both butterfly levels.  After it a NoPivot factorization (rflu_getrf_*_dev with pivot = 0) is safe.
 * rflu_butterfly_vec_*: X <- U' X (transpose_u != 0) or X <- V X (transpose_u == 0) for nrhs column vectors -- the two
 * products around ldiv! in 🦋solve! (:50-52), applied as butterflies instead of dense matrices. */
int rflu_butterfly_mul_f64_dev(rflu_handle_t handle, int64_t n, double* A_dev, int64_t lda, const double* uv_dev);
int rflu_butterfly_mul_f32_dev(rflu_handle_t handle, int64_t n, float* A_dev, int64_t lda, const float* uv_dev);
int rflu_butterfly_vec_f64_dev(rflu_handle_t handle, int64_t n, int64_t nrhs, double* X_dev, int64_t ldx,
                               const double* uv_dev, int transpose_u);
int rflu_butterfly_vec_f32_dev(rflu_handle_t handle, int64_t n, int64_t nrhs, float* X_dev, int64_t ldx,
                               const float* uv_dev, int transpose_u);

/* ---- synthetic input on device (bench / tests; not part of the reference's surface) ----
 * Fill the m x n sub-block starting at global (i0, j0) of an M_global-row uniform[0,1) matrix:
 * element (i,j) = u01(seed, (j0+j)*M_global + (i0+i)) -- bit-identical to oracle/rflu_oracle.c:rfo_uniform01.
 * row_major = 0: A[i + j*ld]; row_major = 1: A[i*ld + j].  diag_add is added to global diagonal entries. */
int rflu_fill_uniform_f64_dev(rflu_handle_t handle, double* A_dev, int64_t m, int64_t n, int64_t ld, int row_major,
                              uint64_t seed, int64_t M_global, int64_t i0, int64_t j0, double diag_add);
int rflu_fill_uniform_f32_dev(rflu_handle_t handle, float* A_dev, int64_t m, int64_t n, int64_t ld, int row_major,
                              uint64_t seed, int64_t M_global, int64_t i0, int64_t j0, double diag_add);
/* ---- multi-GPU: 1-D block-column layout over the GPUs of one node, ONE process (BASELINE configs 3-4).
 * The reference has no distributed path; this is the partition the north star specifies.  Block column b (width `block`, a
 * multiple of 64) lives on logical device (b / run) % ndev inside that device's ROW-MAJOR slab (n rows x its local
 * columns, element (i, jl) at slab[i*ld + jl]); per block column the owner factors the panel with the single-GPU
 * recursion and ONE ncclBroadcast (RCCL, enqueued on the library's panel streams -- no host synchronisation per block
 * column) carries {L\U panel, pivot segment} to the other devices, which then apply laswp / TRSM / GEMM to their slabs;
 * one block column of lookahead.  devs[] may name ONE physical device several times ("fake multi-GPU"): the broadcast
 * then is a device-to-device copy, so the whole partition logic runs -- and is tested -- on a single GPU.
 * ipiv_host (length n, HOST memory, global 1-based rows) and *info as for rflu_getrf_*; ipiv_host may be NULL iff pivot == 0. */
typedef struct rflu_mgpu_s* rflu_mgpu_t;
int rflu_mgpu_create(rflu_mgpu_t* out, int ndev, const int* devs);
int rflu_mgpu_destroy(rflu_mgpu_t mgpu);
int rflu_mgpu_ndev(rflu_mgpu_t mgpu);
int rflu_mgpu_is_fake(rflu_mgpu_t mgpu);
/* number of ncclBroadcast calls this object has enqueued so far (0 in fake mode and with one device, unless
 * RFLU_MGPU_FORCE_RCCL=1 made a one-device object build a one-rank communicator and broadcast to itself: the way the
 * collective's code path is executed on a box with a single GPU) */
int64_t rflu_mgpu_collectives(rflu_mgpu_t mgpu);
/* The RFLU_* tuning variables are read ONCE per handle (rflu_create / rflu_mgpu_create).  A host that changes them between
 * calls asks for another read: rflu_reload_tuning for a single-GPU handle, this one for the per-device handles of a multi-GPU
 * object. */
int rflu_mgpu_reload_tuning(rflu_mgpu_t mgpu);
/* number of local columns of logical device d for an n-column matrix (-1 on bad arguments) */
int64_t rflu_mgpu_local_cols(int64_t n, int64_t block, int ndev, int64_t run, int d);
int rflu_getrf_f64_mgpu(rflu_mgpu_t mgpu, int64_t n, double* const* slabs_dev, const int64_t* lds, int64_t* ipiv_host,
                        int pivot, int64_t block, int64_t run, int64_t* info);
int rflu_getrf_f32_mgpu(rflu_mgpu_t mgpu, int64_t n, float* const* slabs_dev, const int64_t* lds, int64_t* ipiv_host,
                        int pivot, int64_t block, int64_t run, int64_t* info);
/* synthetic input (bench / tests): every slab receives its block columns of the n x n uniform[0,1) matrix of
 * rflu_fill_uniform_* (same seed -> the same matrix as on one GPU) */
int rflu_mgpu_fill_uniform_f64(rflu_mgpu_t mgpu, int64_t n, double* const* slabs_dev, const int64_t* lds, int64_t block,
                               int64_t run, uint64_t seed, double diag_add);
int rflu_mgpu_fill_uniform_f32(rflu_mgpu_t mgpu, int64_t n, float* const* slabs_dev, const int64_t* lds, int64_t block,
                               int64_t run, uint64_t seed, double diag_add);

/* ---- built-in per-kernel timers (hipEvents on the launch stream around every launch of a class) ----
 * enable = 1: synchronous mode -- every launch is bracketed and waited for; the factorization then runs the one-stream
 *             blocked schedule (each kernel alone on the GPU: the kernel's own roofline);
 * enable = 2: in-schedule mode -- event pairs are recorded on whatever stream a launch goes to and resolved when the
 *             timers are read; the default two-stream lookahead schedule is left untouched (what a kernel achieves next
 *             to the other stream's work);
 * enable = 3: the single-stream blocked schedule of mode 1 with the event pairs of mode 2: nothing is waited for between the
 *             launches, so the GPU does not go idle (and its power management does not lower the clock) behind every kernel --
 *             each kernel alone on the GPU at the clock of a busy GPU (bench.py's roofline pass);
 * enable = 0: off.  Enabling resets the timers.  rflu_profile_get returns accumulated milliseconds, launch count and the
 * algorithmic work (flops for GEMM/TRSM/PANEL, bytes for LASWP/TRANSPOSE) of class k since enabling. */
int rflu_profile_enable(rflu_handle_t handle, int enable);
int rflu_profile_get(rflu_handle_t handle, int kclass, double* ms, int64_t* launches, double* work);
/* algorithmic (minimum) HBM bytes of the launches of class k since enabling (GEMM: A and B once, C in and out) */
int rflu_profile_get_bytes(rflu_handle_t handle, int kclass, double* bytes);

#ifdef __cplusplus
}
#endif
#endif /* RFLU_H */
