// rflu_internal.hpp -- shared declarations of librflu.so (MI355X / gfx950 only; no other targets, no CPU fallback).
//
// Internal data layout ("R layout"): the matrix lives ROW-major in HBM, element (i,j) at R[i*ld + j], ld a multiple
// of 16 elements so every row starts on a 128-byte line.  Why: the row interchanges of partial pivoting
// (apply_permutation!, /root/reference/src/lu.jl:164-188) then move contiguous row segments (perfectly coalesced,
// 32 B of traffic per pivot per column, no cache-line amplification), and a panel row is one contiguous 512-byte run.
// The column-major boundary (rflu_getrf_*_dev) converts with two tiled transposes.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <functional>
#include <vector>

#include "../../include/rflu.h"

namespace rflu {

constexpr int NB = 64;            // leaf panel width == pivot chunk size (columns per cooperative panel kernel)
constexpr int PANEL_THREADS = 512;  // 8 waves = 2 per SIMD: the second wave fills the issue gaps of a latency-bound step
constexpr int MAX_PANEL_WGS = 256;  // one workgroup per CU at most: all must be co-resident (they spin on each other)

// ---- error plumbing -------------------------------------------------------------------------------------------------
void set_error(const char* fmt, ...);
#define RFLU_HIP(call)                                                                         \
    do {                                                                                       \
        hipError_t e__ = (call);                                                               \
        if (e__ != hipSuccess) {                                                               \
            ::rflu::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
            return RFLU_ERR_HIP;                                                               \
        }                                                                                      \
    } while (0)
#define RFLU_TRY(expr)            \
    do {                          \
        int s__ = (expr);         \
        if (s__ != RFLU_OK) return s__; \
    } while (0)

// ---- per-kernel-class timers ------------------------------------------------------------------------------------------
struct ProfSlot {
    double ms = 0.0;
    int64_t launches = 0;
    double work = 0.0;
    double bytes = 0.0;  // algorithmic (minimum) HBM bytes of the launches
};

// ---- tuning and debugging variables -----------------------------------------------------------------------------------
// Every RFLU_* environment variable of the library lives here: read ONCE per handle (rflu_create -> Tune::load_env) -- no
// getenv on the factorization paths -- and again only when the host asks for it (rflu_reload_tuning; the tests switch
// variables between calls).  INTEGRATION.md lists them.
struct Tune {
    // leaf panel
    int panel_pw = 0;                  // RFLU_PANEL_PW: floor for the row waves per cooperative workgroup (1|2|4|8)
    int panel_rpw = 0;                 // RFLU_PANEL_RPW: rows per workgroup of the cooperative leaf, forced (64|128|256|384|512; experiments)
    int panel_spare = 1;               // RFLU_PANEL_SPARE: 384-row workgroups whose communication wave has a SIMD for itself, for panels of ...
    int64_t panel_spare_min = 8192;    // RFLU_PANEL_SPARE_MIN: ... more than this many rows (and at most 32 * 384)
    int panel_maxg = 32;               // RFLU_PANEL_MAXG: most workgroups of the short-workgroup leaves
    int panel_ballast = 96 * 1024;     // RFLU_PANEL_BALLAST: dynamic LDS asked for by the 64- / 128-row workgroups (keeps them off shared CUs)
    int64_t panel_local_min = 256;     // RFLU_PANEL_LOCAL_MIN
    int64_t panel_local_rows = -1;     // RFLU_PANEL_LOCAL_ROWS: tallest XCD-local panel (-1: 4096 Float64 / 8192 Float32)
    int64_t panel_local_pw8_rows = 0;  // RFLU_PANEL_LOCAL_PW8_ROWS: XCD-local panels taller than this use 512-row workgroups (0: never)
    int poll_delay = 700;              // RFLU_POLL_DELAY: clocks between a workgroup's header publish and its poll round
    int poll_adapt = 1;                // RFLU_POLL_ADAPT: adapt that delay step by step
    // kernels
    int laswp_lpr = 0;                 // RFLU_LASWP_LPR
    int gemm_flags = 1;                // RFLU_GEMM_FLAGS
    int skinny_max_k = 128;            // RFLU_SKINNY_MAXK
    int skinny_wide = 0;               // RFLU_SKINNY_WIDE
    int64_t gemm_cfirst_below = (int64_t)1 << 40;   // RFLU_GEMM_CFIRST_BELOW
    int gemm_masked = -1;              // RFLU_GEMM_MASKED: stand-alone rflu_gemm_* calls on the CU-masked stream leaving this many CUs free
    int64_t ld_pad = 0;                // RFLU_LD_PAD
    int64_t trsv_max_rhs = 32;         // RFLU_TRSV_MAX_RHS
    int64_t trsm_chain_max_rhs = 320;  // RFLU_TRSM_CHAIN_MAX_RHS: up to this many right-hand sides the cooperative solve in passes of 64 (MFMA)
    int trsm_chain_cached = 1;         // RFLU_TRSM_CHAIN_CACHED=0: every workgroup fetches x_d with cache-bypassing loads (trsv.hip)
    int trsm_chain_split = 2;          // RFLU_TRSM_CHAIN_SPLIT: a pass of 64 right-hand sides as 0 = one chain of 64 columns, 1 = four of 16 in runs of two blocks,
                                       // 2 = two chains of 32 side by side (trsv.hip; n = 16384, 64 columns: 5.55 / 5.41 / 4.81 ms)
    // streams and queues
    int queue_check = 1;               // RFLU_QUEUE_CHECK
    int queue_trace = 0;               // RFLU_QUEUE_TRACE
    // block-column lookahead schedule
    int split_all = 0;                 // RFLU_SPLIT_ALL
    double split_share = 0.5;          // RFLU_SPLIT_SHARE
    double split_scale = 1.0;          // RFLU_SPLIT_SCALE
    int max_reserve = 64;              // RFLU_MAX_RESERVE
    int wide_narrow = 1;               // RFLU_WIDE_NARROW: default block width of matrices of 20480 columns and more = wide, then 512 for the last ...
    int64_t narrow_cols = 16384;       // RFLU_NARROW_COLS: ... this many columns
    int min_reserve = 32;              // RFLU_RESERVE_CUS
    int64_t confine_rows = (int64_t)1 << 40;   // RFLU_CONFINE_ROWS
    int64_t merge_rows = -1;           // RFLU_MERGE_ROWS (-1: 8192 Float64, never Float32)
    // leaf-wise schedule
    int leafwise = 1;                  // RFLU_LEAFWISE
    int64_t leafwise_rows = -1;        // RFLU_LEAFWISE_ROWS (-1: 8192 Float64 / 16384 Float32)
    int swap_su = -1;                  // RFLU_SWAP_SU (-1: by size)
    int gate_fold = 1;                 // RFLU_GATE_FOLD
    int gate_trace = 0;                // RFLU_GATE_TRACE
    int schedule_events = 0;           // RFLU_SCHEDULE=events: cross-stream edges by hipEvents only, no device-side gates (what a
                                       // counter-collection run needs: rocprofv3 --pmc runs one kernel at a time)
    int time_enqueue = 0;              // RFLU_TIME_ENQUEUE
    int tail_overlap = 1;              // RFLU_TAIL_OVERLAP
    // host-pointer entry
    int64_t host_early_out = 512;      // RFLU_HOST_EARLY_OUT
    int host_trace = 0;                // RFLU_HOST_TRACE
    int swap_late = 1;                 // RFLU_SWAP_LATE: leaf-wise schedule: side stream on the 224-CU stream (updates on the 192-CU one) from the first panel of at most swap_rows rows on, also behind a lookahead part (0: round 3's assignment)
    int64_t swap_rows = 8192;          // RFLU_SWAP_ROWS
    int leaf_fuse = 1;                 // RFLU_LEAF_FUSE: leaf-wise schedule: interchanges + diagonal inverse + 64-row solve of the next leaf's columns in ONE launch (leaf_la_kernel)
    int host_threads = 8;              // RFLU_HOST_THREADS
    // multi-GPU
    int64_t mgpu_big_reserve = 128;    // RFLU_MGPU_BIG_RESERVE
    int64_t mgpu_tall_rows = -1;       // RFLU_MGPU_TALL_ROWS
    int mgpu_sync = 0;                 // RFLU_MGPU_SYNC
    // fault injection (tests): the cooperative leaf launch with this sequence number inside a factorization waits for a
    // participant that does not exist, runs into its bounded spin and raises the timeout flag (RFLU_ERR_TIMEOUT at the end)
    int debug_ghost_leaf = -1;         // RFLU_DEBUG_GHOST_LEAF
    // persistent update engine (engine.hip): the default schedule where it measures faster than the streams, and the host entry
    int engine = -1;                   // RFLU_ENGINE: 1 = the side / update streams' work is pulled by the resident engine wherever it can be, 0 = never,
                                       // -1 (default) = where it measures faster: pivoted, default block width, 12288 < min(m, n), m <= 16384 (N=16384: 72 vs 75 ms, Float32 56.0 vs 58.8)
    int engine_policy = 0;             // RFLU_ENGINE_POLICY: 0 = leftmost column block first, 1 = oldest panel piece first
    int engine_wgs = 0;                // RFLU_ENGINE_WGS: resident workgroups (0: two per CU of the update mask)
    int64_t engine_rows = 0;           // RFLU_ENGINE_ROWS: block columns whose panels are taller than this go through the engine, the streams take over below (0: the engine
                                       // to the end -- a hand-over waits for the engine's backlog on the far right: N=16384 79.6 ms at 4096 against 75.3)
    int engine_wc = 512;               // RFLU_ENGINE_WC: width of the engine's column blocks (a multiple of 128 dividing the block-column width; N=16384: 128: 82.7 ms, 256: 80.5, 512: 79.6)
    int engine_retire = -1;            // RFLU_ENGINE_RETIRE: from the first panel of at most this many rows on the engine's workgroups on the chain's XCD are gone and the
                                       // leaves XCD-local (-1: 4096, 2048 for matrices of 16384 rows or more; 0: they stay to the end, every leaf any-placement)
    int engine_replay = 0;             // RFLU_ENGINE_REPLAY=1 (measurement): the engine ALONE on the image a previous factorization of the same shape left behind -- every
                                       // leaf counts as done from the start, no chain is launched (results are meaningless; this is what rocprofv3 --pmc, which runs
                                       // one kernel at a time, can collect the resident kernel's counters on: scripts/pmc_engine.sh)
    int engine_host = 1;               // RFLU_ENGINE_HOST: host-pointer entry (rflu_getrf_*) through the engine: the way in overlaps the factorization
    int engine_write_through = 1;      // RFLU_ENGINE_WRITE_THROUGH: Schur tiles stored write-through, no L2 write-back per tile (N=16384: 84 -> 79 ms; 0: release fence per tile)
    int engine_leaf_xcds = 1;          // RFLU_ENGINE_LEAF_XCDS / RFLU_ENGINE_LEAF_WGS: the workgroups with blockIdx % 8 in [1, xcds] and blockIdx / 8 < wgs serve the
    int engine_leaf_wgs = 16;          //   leaf windows (K = 64) only: what the chain waits for never queues behind 127-us tiles (N=16384: 97 -> 84 ms; xcds 0: none)
    int engine_ahead = 1;              // RFLU_ENGINE_AHEAD: a leaf is applied leaf by leaf (K = 64) to its own block column and to this many block columns right of it
                                       // (engine.hpp: EngGeo::ahead; 1 = the stream schedule's window: the chain then waits 0.4 .. 2.5 ms at the last leaf of every block column)
    int engine_solve_rl = 1;           // RFLU_ENGINE_SOLVE_RL (experiments build): the engine's block-row solves right-looking, the block row in registers (0: left-looking,
                                       // X read back from memory; same box, alternating: N=16384 72.0-72.3 vs 72.8-73.1 ms, bit-identical)
    int engine_host_lag = 5;           // RFLU_ENGINE_HOST_LAG: host entry: whole-block-column operations that lag the chain by this many block columns go first, so
                                       // that block rows become final -- and leave -- while the factorization runs (N=16384: 3: 123 ms, 5: 109-110, 8: 112, 0 = never: 118)
    void load_env();                   // driver.cpp
};

struct Handle {
    Tune tune;
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // second stream for the lookahead driver: the deferred trailing updates run here, restricted by a CU mask to
    // 224 of the 256 CUs so that the cooperative panel kernels of the critical path always find 32 free CUs
    hipStream_t ustreams[8] = {};     // ustreams[r]: CU mask leaving 32*r CUs to the critical path (r = 1..7)
    bool mask_failed = false;         // a CU-masked stream could not be created: plain streams may share a hardware queue, so the
                                      // device-side gates (which need the streams to run concurrently) are not used
    // host-pointer entry (getrf_host): finished block rows travel back while the rest is still being factored
    std::function<int(int64_t)> progress;      // called by the block-column schedules: every kernel that writes rows [0, r) is enqueued
    std::function<int()> before_sync;          // called by getrf_rm when the whole factorization is enqueued, before it waits for it
    bool out_done = false;                     // set by before_sync: the factors are already in the caller's buffer
    std::vector<hipEvent_t> out_events;        // event pool of that path
    void* bounce[2] = {};                      // pinned bounce buffers of that path (one chunk of rows each)
    void* out_stage = nullptr;                 // device staging of the outgoing chunks (two pieces), so that the input copy stays intact
    size_t out_stage_bytes = 0;
    size_t bounce_bytes = 0;
    std::vector<hipStream_t> parked_streams;   // masked streams that shared a pipe with a stream in use (validate_queues): kept idle
    long long* qprobe_slots = nullptr;         // device: 4 stamps of the pipe probe
    std::vector<hipStream_t> queues_ok_streams;   // validate_queues: caller streams the masked streams in use have been checked against
    int queues_ok_count = 0;                      // ... and how many masked streams existed then
    int queue_unresolved = 0;                     // validate_queues: checks that ended with a conflict it could not settle
    bool queue_giveup = false;                    // ... three of them: the placement is taken as it is from then on
    hipStream_t pstreams[8] = {};     // pstreams[r]: the complement -- exactly those 32*r CUs (critical path of the update-bound phase)
    bool panel_attr_set[2][2][2] = {};     // [any placement|XCD-local][Float64|Float32][64|128 rows]: dynamic-LDS attribute of the small-workgroup leaves
    hipEvent_t tail_event = nullptr;       // column-major entry: the columns right of the first block column are still being
                                           // transposed on the update stream; set = pending, consumed by getrf_rm
    hipEvent_t tail_event_obj = nullptr, tail_fork_obj = nullptr;
    unsigned long long* gates = nullptr;   // device: [0] critical path, [1] side stream 1, [2] side stream 2 (leaf counters)
    unsigned long long* gate_ptr[3] = {};  // the three counters
    unsigned long long gate_epoch = 0;     // counters only grow: leaf g of a factorization is gate_epoch + g + 1
    long long* gate_stamps = nullptr;      // RFLU_GATE_TRACE: wall-clock stamps of the signals, [3][4096] (scripts/gate_trace.py)
    std::vector<hipEvent_t> events;   // reusable, timing disabled
    int last_path = RFLU_PATH_NONE;
    int num_cus = 256;

    // workspaces (grown on demand, kept for reuse across calls -- LinearSolve-style cache reuse)
    void* work = nullptr;        // row-major copy of the matrix for the column-major entry points
    size_t work_bytes = 0;
    int64_t* ipiv_dev = nullptr; // device pivots for the host entry points
    size_t ipiv_cap = 0;
    void* hostA_dev = nullptr;   // device copy of the host matrix (host entry points)
    size_t hostA_bytes = 0;
    void* rhs_work = nullptr;    // row-major copy of the right-hand sides (getrs)
    size_t rhs_work_bytes = 0;
    void* hostB_dev = nullptr;   // device copy of host right-hand sides (getrs host entry)
    size_t hostB_bytes = 0;

    // pivot bookkeeping: for every chunk of NB pivots the list of (dst,src) row moves equivalent to its interchanges
    int* pm_cnt = nullptr;
    int* pm_dst = nullptr;
    int* pm_src = nullptr;
    int64_t pm_chunks = 0;
    void* linv = nullptr;        // inverses of the 64x64 diagonal blocks of L, one per leaf (pm_chunks x 64 x 64 elements)
    unsigned trsv_tag = 0;       // last tag used by the cooperative solve (trsv.hip) in its exchange area ...
    void* trsv_area = nullptr;   // ... which lives at this address inside linv_tmp
    void* linv_tmp = nullptr;    // same for stand-alone rflu_trsm_rm_* calls
    size_t linv_tmp_bytes = 0;

    // cooperative panel scratch: double-buffered granule records + status words
    unsigned long long* pscratch = nullptr;
    size_t pscratch_bytes = 0;
    unsigned epoch = 1;          // next unused granule tag
    int coop_leaf_seq = 0;       // cooperative leaf launches since the start of the current factorization (Tune::debug_ghost_leaf)
    int64_t* info_dev = nullptr; // [0] = info, [1] = panel error flags (bit0 timeout, bit1 XCD placement mismatch)
    // pipelined leaf kernel with a communication wave (panel_local.hip), RFLU_PANEL_LOCAL: 0 = off (the kernels of
    // panel.hip), 2 = on with sc1 records on any placement (default), 1 = all workgroups on XCD panel_xcc with plain-store
    // records through that XCD's L2 (needs that XCD free: experiments only)
    int panel_local = 2;
    int panel_single = 1;        // RFLU_PANEL_SINGLE: leaves of at most 512 rows in one workgroup, LDS only (panel_single.hip)
    int panel_blocked = 0;       // RFLU_PANEL_BLOCKED=1: full pivoted leaves by the sub-panel kernel with one chain wave per workgroup (panel_blocked.hip;
                                 // bit-identical, measured no faster than the default leaves: opt-in)
    int panel_local_maxg = 64;
    int panel_xcc = 0;
    int trsv_max_wgs[2] = {0, 0};    // same for the cooperative solve kernels (asked on first use, per element type: [Float64, Float32])
    int trsm32_per_cu[2] = {0, 0};
    int trsm16_per_cu[2] = {0, 0};   // workgroups of the 16-column block solve a CU holds (trsv.hip: chains side by side)
    int panel_max_wgs = 0;       // how many workgroups of the cooperative panel kernels the device holds at once (occupancy query)
    bool coop_launch = false;    // RFLU_COOP_LAUNCH=1: hipLaunchCooperativeKernel (launch-time residency check, +15-19 us each)
    bool la_attr_set[2] = {false, false};     // dynamic-LDS opt-in of leaf_la_kernel (f64, f32)
    bool gemm_attr_set[2] = {false, false};   // dynamic-LDS opt-in of the GEMM kernels done on this handle's device (f64, f32)
    bool eng_attr_set[2] = {false, false};    // same for the persistent update engine (engine.hip)
    void* eng_state = nullptr;                // device: EngState (engine.hpp)
    void* eng_host = nullptr;                 // pinned host image of its initial value
    long long* eng_trace_buf = nullptr;       // device: RFLU_ENGINE_TRACE stamps + workgroup-time accounting (measurement only)
    bool eng_active = false;                  // a factorization's engine is resident (factor_leafwise in engine mode .. the join with its stream)
    // host entry through the engine (driver.cpp: getrf_host_engine): the matrix arrives while it is being factored
    bool eng_host_mode = false;
    unsigned long long* eng_rows_final = nullptr;       // pinned host word (rows of the factors that are final), ...
    unsigned long long* eng_rows_final_dev = nullptr;   // ... its device address
    int64_t* info_pinned = nullptr;

    // timers
    bool prof = false;           // synchronous mode: every launch bracketed and waited for (switches to the one-stream schedule)
    bool prof_async = false;     // in-schedule mode: event pairs recorded on the launch streams, resolved by profile_get
    bool prof_one_stream = false;   // mode 3: the one-stream schedule of the synchronous mode with the event pairs of the in-schedule mode
                                    // (nothing waited for between launches: the GPU does not idle -- and drop its clock -- behind every kernel)
    ProfSlot slots[RFLU_K_COUNT];
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    struct AsyncRec { int k; hipEvent_t a, b; double work, bytes; };
    std::vector<AsyncRec> async_recs;      // pending event pairs of the in-schedule mode
    std::vector<hipEvent_t> async_pool;    // timing events kept for reuse
};

struct ProfScope {
    Handle* h;
    int k;
    double work;
    double bytes;
    bool on;
    hipEvent_t ea = nullptr, eb = nullptr;
    static hipEvent_t take(Handle* h) {
        if (!h->async_pool.empty()) { hipEvent_t e = h->async_pool.back(); h->async_pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    ProfScope(Handle* h_, int k_, double work_, double bytes_ = 0.0) : h(h_), k(k_), work(work_), bytes(bytes_), on(h_->prof) {
        if (on) (void)hipEventRecord(h->ev0, h->stream);
        else if (h->prof_async) {   // no wait: the pair is resolved when the timers are read
            ea = take(h);
            eb = take(h);
            if (ea) (void)hipEventRecord(ea, h->stream);
        }
    }
    ~ProfScope() {
        if (ea && eb) {
            (void)hipEventRecord(eb, h->stream);
            h->async_recs.push_back({k, ea, eb, work, bytes});
        }
        if (on) {
            (void)hipEventRecord(h->ev1, h->stream);
            (void)hipEventSynchronize(h->ev1);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, h->ev0, h->ev1);
            h->slots[k].ms += ms;
            h->slots[k].launches += 1;
            h->slots[k].work += work;
            h->slots[k].bytes += bytes;
        }
    }
};

int ensure_bookkeeping(Handle* h, int64_t rows);
int ensure_buffer(void** ptr, size_t* cap, size_t need);  // grow-only device buffer (hipFree + hipMalloc)

// ---- kernel launchers (each returns an rflu_status); all pointers are device pointers in R layout -----------------------
int panel_local_rows_per_wg(const Handle* h, int64_t rows, size_t esize);   // panel_local.hip: rows per workgroup of a pivoted leaf
// panel.hip: does a full pivoted leaf of `rows` rows go to the sub-panel kernel (panel_blocked.hip), and the workgroups the leaf
// kernel of launch_panel's choice takes (what a schedule has to keep free)
bool panel_use_blocked(const Handle* h, int64_t rows, size_t esize);
int64_t panel_plan_wgs(const Handle* h, int64_t rows, size_t esize, int pivot);
int launch_heat(Handle* h, int cus, double usec);   // gemm.hip: clock keeper
// optional early-completion signal of a GEMM launch: the tiles of the first `first_cols` columns of C are computed first and
// the last of them publishes `val` in *flag (a stream gate, see LaswpGate); cnt: zero-initialised wrapping counter
struct GemmSignal {
    int64_t first_cols = 0;
    unsigned long long* flag = nullptr;
    unsigned long long val = 0;
    unsigned* cnt = nullptr;
};
template <typename T>
int launch_gemm(Handle* h, int64_t M, int64_t N, int64_t K, const T* A, int64_t lda, const T* B, int64_t ldb, T* C,
                int64_t ldc, GemmSignal sig = GemmSignal{});
template <typename T>
int launch_trsm_base(Handle* h, int64_t nb, int64_t nrhs, const T* L, int64_t ldl, T* B, int64_t ldb);
// fused strip TRSM (n <= 256) on pre-inverted 64x64 diagonal blocks, and the batched inversion of those blocks
template <typename T>
int launch_trsm_fused(Handle* h, int64_t n, int64_t nrhs, const T* L, int64_t ldl, const T* Linv, T* B, int64_t ldb);
template <typename T>
int launch_triu_base(Handle* h, int64_t nb, int64_t nrhs, const T* U, int64_t ldu, T* B, int64_t ldb);
template <typename T>
int launch_diag_inv(Handle* h, int64_t n, const T* L, int64_t ldl, T* Linv);
template <typename T>
int launch_trsm_inv64(Handle* h, int64_t n, int64_t nrhs, const T* Linv, T* B, int64_t ldb);   // one block, LDS-free
// cooperative solve for few right-hand sides (trsv.hip): B <- U^-1 L^-1 B, interchanges already applied
template <typename T>
int launch_trsv_coop(Handle* h, int64_t n, int64_t nrhs, const T* R, int64_t ld, T* B, int64_t ldb, bool wide = false);
// stream gates folded into an interchange launch (laswp.hip; used by factor_leafwise): hold the launch until *wait_flag >=
// wait_val, and let its last workgroup publish signal_val (signal_cnt: a zero-initialised counter that wraps by itself)
struct LaswpGate {
    const unsigned long long* wait_flag = nullptr;
    unsigned long long wait_val = 0;
    unsigned long long* signal_flag = nullptr;
    unsigned long long signal_val = 0;
    unsigned* signal_cnt = nullptr;
    int64_t* info = nullptr;   // timeout flag (info[1] bit 0)
};
template <typename T>
int launch_laswp(Handle* h, T* R, int64_t ld, int64_t c0, int64_t ncols, int64_t chunk0, int64_t chunk1);
// apply chunks [chunk0, chunk1) to two column ranges at once: [c0, c0+ncolsA) and [c1, c1+ncolsB)
template <typename T>
int launch_laswp3(Handle* h, T* R, int64_t ld, int64_t c0, int64_t ncolsA, int64_t c1, int64_t ncolsB, int64_t c2,
                  int64_t ncolsC, int64_t chunk0, int64_t chunk1, int64_t inv_nb, int64_t inv_cnt, const T* inv_L,
                  T* inv_out, LaswpGate gate = LaswpGate{});
template <typename T>
int launch_laswp2(Handle* h, T* R, int64_t ld, int64_t c0, int64_t ncolsA, int64_t c1, int64_t ncolsB, int64_t chunk0,
                  int64_t chunk1, int64_t inv_nb = 0, const T* inv_L = nullptr, T* inv_out = nullptr,
                  LaswpGate gate = LaswpGate{});
// leaf-wise schedule, one launch behind a full leaf (rows / columns c0.., move list `chunk`): the leaf's interchanges on the next
// leaf's 64 columns [la0, la0 + 64), the inverse of its diagonal block (-> inv_out) and the solve of those columns' top 64 rows
template <typename T>
int launch_leaf_la(Handle* h, T* R, int64_t ld, int64_t la0, int64_t chunk, int64_t c0, const T* inv_L, T* inv_out, LaswpGate gate);
// fold the interchanges ipiv[k0..k1) (k0 a multiple of NB) into per-chunk row-move lists
int launch_perm_build(Handle* h, const int64_t* ipiv, int64_t k0, int64_t k1, int64_t m);
size_t panel_scratch_bytes();
int panel_resident_limit(int num_cus);   // min over the cooperative leaf kernels of (resident workgroups per CU) * num_cus
size_t panel_trace_offset_bytes();
size_t panel_trace_all_offset_bytes();
size_t panel_trace_all_words();
template <typename T>
int launch_panel(Handle* h, T* R, int64_t ld, int64_t m, int64_t r0, int64_t c0, int64_t w, int64_t* ipiv, int pivot);
template <typename T>
int launch_panel_pair(Handle* h, T* R, int64_t ld, int64_t m, int64_t r0, int64_t c0, int64_t* ipiv);
template <typename T>
int launch_transpose(Handle* h, int64_t rows_out, int64_t cols_out, const T* in, int64_t ld_in, T* out, int64_t ld_out);
// the same on an explicit stream, touching nothing of the handle (a second host thread feeds the matrix in: getrf_host_engine)
template <typename T>
int launch_transpose_on(hipStream_t st, int64_t rows_out, int64_t cols_out, const T* in, int64_t ld_in, T* out, int64_t ld_out);
int launch_gate_signal_on(hipStream_t st, unsigned long long* flag, unsigned long long value);
template <typename T>
int launch_fill_uniform(Handle* h, T* A, int64_t m, int64_t n, int64_t ld, int row_major, uint64_t seed,
                        int64_t M_global, int64_t i0, int64_t j0, double diag_add);
int launch_iota_ipiv(Handle* h, int64_t* ipiv, int64_t k0, int64_t n);
int queue_probe(hipStream_t a, hipStream_t b, long long* slots, long long* host3, int* shared);   // laswp.hip: do two streams share a pipe?
int queue_probe_rate(hipStream_t a, hipStream_t b, int n, long long* slots, double* us_per_kernel);
int launch_gate_signal(Handle* h, unsigned long long* flag, unsigned long long value, long long* stamp = nullptr);   // laswp.hip: device-side stream gates
int launch_gate_wait(Handle* h, const unsigned long long* flag, unsigned long long value);
// butterfly.hip: A <- U' A V (column-major, in place) and x <- U' x (mode 0) / x <- V x (mode 1)
template <typename T>
int launch_butterfly_mul(Handle* h, int64_t n, T* A, int64_t lda, const T* uv);
template <typename T>
int launch_butterfly_vec(Handle* h, int64_t n, int64_t nrhs, T* X, int64_t ldx, const T* uv, int mode);

}  // namespace rflu
