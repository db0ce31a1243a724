// engine.hpp -- the persistent update engine (engine.hip): state shared between the host schedule (driver.cpp: factor_engine)
// and the device.  See engine.hip for the design.
#pragma once

#include <stdint.h>

#include "rflu_internal.hpp"

namespace rflu {

constexpr int ENG_MAX_CB = 512;                       // column blocks a factorization may have
constexpr unsigned ENG_SEQ_DONE = 0x7fffffffu;        // claim word of a column block that has received everything

// One per column block (64 bytes).  A "sequence" is one stage of one block column's update: seq = 2*b + stage, stage 0 = the
// interchanges of panel b on this column block + the block-row solve, stage 1 = the Schur update tiles.
struct EngCB {
    unsigned long long claim;    // (seq << 32) | next unclaimed unit of that sequence
    unsigned long long done;     // finished units of the current sequence
    unsigned long long lclaim;   // deferred interchanges of LATER panels on this (finished) column block: (b << 32) | next unit
    unsigned long long ldone;
    unsigned long long ready;    // != 0: every update of the panels in front of this column block has been applied
    long long t_ready;           // wall clock (100 MHz) when `ready` was raised / when the panel of this block column was
    long long t_panel;           // published (rflu_debug_engine_times)
    unsigned long long pad[1];
};

struct EngState {
    unsigned long long panel_done;   // panels [0, panel_done) are factored (written by the critical-path stream)
    unsigned long long remaining;    // column-block sequences (main and left) still unfinished: the engine exits at 0
    unsigned long long abort;        // != 0: leave (timeout somewhere)
    unsigned long long pad[5];
    EngCB cb[ENG_MAX_CB];
};

template <typename T>
struct EngArgs {
    T* R;
    int64_t ld;
    int m, n, mn;
    int W;          // block-column width (a multiple of 128)
    int nbp;        // the engine applies panels [0, nbp)
    int ncb;        // column blocks of width W covering [0, n)
    int pivot;
    int policy;     // 0: oldest panel first (right-looking order); 1: leftmost column block first
    const T* linv;  // inverses of the 64x64 diagonal blocks, one per 64 rows
    const int* pm_cnt;
    const int* pm_dst;
    const int* pm_src;
    EngState* st;
    int64_t* info;  // info[1] bit 0: timeout
    int gemm_flags;
    int x[8];       // experiment switches (Tune::engine_x)
};

// host-side mirror of the device's unit counts (driver.cpp initialises the state with them)
inline int eng_nseq(int cb, int nbp) { return 2 * (cb < nbp ? cb : nbp); }

template <typename T>
int launch_engine(Handle* h, hipStream_t stream, const EngArgs<T>& a, int wgs);
int launch_eng_signal(Handle* h, unsigned long long* flag, unsigned long long value, long long* stamp = nullptr);
int launch_eng_wait(Handle* h, const unsigned long long* flag, unsigned long long value);
size_t engine_lds_bytes(size_t esize);

}  // namespace rflu
