// engine.hpp -- the persistent update engine (engine.hip): state and work description shared between the host schedule
// (driver.cpp: factor_leafwise in engine mode) and the device.  See engine.hip for the design.
#pragma once

#include <stdint.h>

#include "rflu_internal.hpp"

#if defined(__HIPCC__)
#define RFLU_HD __host__ __device__ __forceinline__
#else
#define RFLU_HD inline
#endif

namespace rflu {

constexpr int ENG_MAX_CB = 512;                       // column blocks a factorization may have
constexpr unsigned ENG_SEQ_DONE = 0x7fffffffu;        // claim word of a column block that has received everything
constexpr int ENG_PREP_COLS = 32;                     // columns of a stage-0 unit

// One per column block (64 bytes).  A column block receives a fixed SEQUENCE OF OPERATIONS (EngOp below), each in two stages:
// seq = 2 * op + stage, stage 0 = interchanges + block-row solve (units of 32 columns), stage 1 = Schur update (128 x 128 tiles).
struct EngCB {
    unsigned long long claim;    // (seq << 32) | next unclaimed unit of that sequence
    unsigned long long done;     // finished units of the current sequence
    unsigned long long lclaim;   // deferred interchanges on this (finished) column block: (left op << 32) | next unit
    unsigned long long ldone;
    unsigned long long prog;     // 2 * (operations completed) + (the first tile column of the current leaf window is complete): the critical-path
                                 // stream waits for prog >= 2 * op + 1 in front of a leaf's lookahead columns (the leftmost of that window)
    unsigned long long leftdone; // (first column block of a block column) own << 32 | left, see "order between the interchanges ..." below
    unsigned long long lprog;    // left operations completed (lprog >= 1: the block column's own later interchanges have reached this column block)
    unsigned long long bigdone;  // (first column block of a block column) column blocks that have completed BIG(this block column)
};

struct EngState {
    unsigned long long arrived;      // host entry: columns [0, arrived) of the row-major workspace are in place (written by the feeding stream)
    unsigned long long remaining;    // column-block sequences (main and left) still unfinished: the engine exits at 0
    unsigned long long abort;        // != 0: leave (timeout somewhere)
    unsigned long long epoch;        // bumped whenever the engine publishes something that may make a unit eligible: an idle workgroup
                                     // watches this word and the critical path's leaf counter instead of sweeping every claim word
    unsigned long long xcc_wgs[8];   // workgroups of the engine that started on XCC x (counted at their start)
    unsigned long long retired;      // workgroups that have left the chain's XCD for good (EngArgs::retire_leaf): the critical path waits for
                                     // retired == xcc_wgs[that XCD] in front of its first XCD-local leaf
    unsigned long long pad[3];
    EngCB cb[ENG_MAX_CB];
};

// Geometry of the schedule (plain integers: host and device compute the same operation lists from it)
struct EngGeo {
    int m, n, mn;
    int W;          // width of a block column = of a panel (a multiple of Wc)
    int Wc;         // width of a COLUMN BLOCK, the unit of the dataflow (a multiple of 128 that divides W): the columns of a block column are
                    // brought up to date in pieces of their own, so that a leaf's window does not wait for the previous leaf's last columns
    int nbp;        // the engine serves the leaves of block columns [0, nbp)
    int ncb;        // column blocks of width Wc covering [0, n)
    int pivot;
    int ahead;      // a leaf is applied leaf by leaf (K = 64) to its own block column and to the `ahead` block columns right of it (>= 1); a block
                    // column as a whole (K = W) to everything further right.  1 is the leaf-wise stream schedule's window.  With 1 the block column
                    // right of the panel receives BIG(b - 1) only once block column b - 1 is complete, and then has to catch up, one leaf window
                    // after the other (each two dependent stages), with the leaves of block column b factored meanwhile -- the chain stood at the
                    // LAST leaf of every block column for 0.4 .. 2.5 ms (N=16384: all of its 12 ms of waiting, profiles/r06*_leaves.txt); with 2
                    // that column block has everything but the current block column's leaves a whole block column earlier
};

enum { ENG_OP_BIG = 0, ENG_OP_LEAF = 1 };

// One operation on a column block: apply the pivot block [j0, j0 + jb) x [j0, j0 + jb) (+ the rows below) to columns [c_lo, c_lo + nc)
struct EngOp {
    int type;
    int j0, jb;     // pivot rows / columns of the applied panel piece
    int c_lo, nc;   // columns that receive it (nc <= 0: nothing to do -- the operation completes by itself)
    int need;       // leaves [0, need) must be factored AND their lookahead launch must have run (diagonal inverse, move list)
    int chunk0, chunk1;   // pivot chunks (64 pivots each) whose interchanges belong to it
};

RFLU_HD int eng_leaves_of_block(const EngGeo& g, int b)
{
    const int j0 = b * g.W;
    const int jb = g.mn - j0 < g.W ? g.mn - j0 : g.W;
    return jb <= 0 ? 0 : (jb + NB - 1) / NB;
}
RFLU_HD int eng_pb(const EngGeo& g, int cb) { return (cb * g.Wc) / g.W; }          // the block column a column block belongs to
RFLU_HD int eng_first_cb(const EngGeo& g, int b) { return b * (g.W / g.Wc); }      // first column block of block column b ...
RFLU_HD int eng_cbs_of_block(const EngGeo& g, int b)                                // ... and how many it has
{
    const int first = eng_first_cb(g, b);
    int cnt = g.ncb - first;
    const int r = g.W / g.Wc;
    if (cnt > r) cnt = r;
    return cnt > 0 ? cnt : 0;
}

// operations of column block cb (in block column pb), in order:
//   BIG(b),  b = 0 .. min(pb - ahead, nbp) - 1 : block column b as a whole (K = W) -- block columns more than `ahead` to the left
//   LEAF(g), g = leaves of block columns pb - ahead .. pb - 1 (those that are served): K = 64, the leaf windows of the block columns in front
//   LEAF(g), g = leaves of block column pb itself but its last (if served): K = 64 on the columns right of the leaf's lookahead strip
RFLU_HD int eng_ahead(const EngGeo& g) { return g.ahead >= 1 ? g.ahead : 1; }
RFLU_HD int eng_nbig(const EngGeo& g, int cb)
{
    int v = eng_pb(g, cb) - eng_ahead(g);
    if (v < 0) v = 0;
    return v < g.nbp ? v : g.nbp;
}
// LEAF ops of cb that belong to the block columns in front of its own, up to (not including) block column `upto` (<= pb)
RFLU_HD int eng_nleaf_front(const EngGeo& g, int cb, int upto)
{
    const int pb = eng_pb(g, cb);
    int lo = pb - eng_ahead(g);
    if (lo < 0) lo = 0;
    int hi = upto < pb ? upto : pb;
    if (hi > g.nbp) hi = g.nbp;
    int cnt = 0;
    for (int b = lo; b < hi; ++b) cnt += eng_leaves_of_block(g, b);
    return cnt;
}
RFLU_HD int eng_nleafn(const EngGeo& g, int cb) { return eng_nleaf_front(g, cb, eng_pb(g, cb)); }
RFLU_HD int eng_nleafo(const EngGeo& g, int cb)
{
    const int pb = eng_pb(g, cb);
    if (pb >= g.nbp) return 0;
    const int nl = eng_leaves_of_block(g, pb);
    return nl > 1 ? nl - 1 : 0;
}
RFLU_HD int eng_nops(const EngGeo& g, int cb) { return eng_nbig(g, cb) + eng_nleafn(g, cb) + eng_nleafo(g, cb); }

RFLU_HD EngOp eng_op(const EngGeo& g, int cb, int k)
{
    EngOp o;
    const int LPB = g.W / NB;
    const int pb = eng_pb(g, cb);
    const int cb0 = cb * g.Wc;
    const int cb_end = cb0 + g.Wc < g.n ? cb0 + g.Wc : g.n;
    const int nbig = eng_nbig(g, cb);
    if (k < nbig) {
        o.type = ENG_OP_BIG;
        o.j0 = k * g.W;
        o.jb = g.mn - o.j0 < g.W ? g.mn - o.j0 : g.W;
        o.c_lo = cb0;
        o.nc = cb_end - cb0;
        o.chunk0 = o.j0 / NB;
        o.chunk1 = (o.j0 + o.jb + NB - 1) / NB;
        o.need = o.chunk1;
        return o;
    }
    const int nln = eng_nleafn(g, cb);
    int leaf;
    if (k - nbig < nln) {   // the block columns in front are full ones (only the last block column of a matrix can be short, and it has nothing behind it)
        int lo = pb - eng_ahead(g);
        if (lo < 0) lo = 0;
        leaf = lo * LPB + (k - nbig);
    } else leaf = pb * LPB + (k - nbig - nln);
    o.type = ENG_OP_LEAF;
    o.j0 = leaf * NB;
    o.jb = g.mn - o.j0 < NB ? g.mn - o.j0 : NB;
    const int la1 = o.j0 + o.jb + NB;   // the leaf's lookahead strip [j0 + jb, la1) is the critical-path stream's own business
    o.c_lo = la1 > cb0 ? la1 : cb0;
    o.nc = cb_end - o.c_lo;             // (<= 0 for the column blocks left of the strip: the operation completes by itself)
    o.chunk0 = leaf;
    o.chunk1 = leaf + 1;
    o.need = leaf + 1;
    return o;
}

// index of LEAF(leaf) in column block cb's operation list (cb in the leaf's own block column or the one right of it)
RFLU_HD int eng_leaf_op_index(const EngGeo& g, int cb, int leaf)
{
    const int LPB = g.W / NB;
    const int lb = leaf / LPB;
    if (lb < eng_pb(g, cb)) return eng_nbig(g, cb) + eng_nleaf_front(g, cb, lb) + (leaf - lb * LPB);
    return eng_nbig(g, cb) + eng_nleafn(g, cb) + (leaf - lb * LPB);
}

// ---- deferred interchanges on finished column blocks (block column pb < nbp): left op 0 = the later leaves of the block column
// itself on the columns of its earlier leaves (one unit per 64-column strip of the column block), left op i >= 1 = block column
// pb + i as a whole
RFLU_HD int eng_nleft(const EngGeo& g, int cb)
{
    const int pb = eng_pb(g, cb);
    if (!g.pivot || pb >= g.nbp) return 0;
    return 1 + (g.nbp - 1 - pb);
}
RFLU_HD int eng_left_need(const EngGeo& g, int cb, int lk)
{
    const int b = eng_pb(g, cb) + lk;
    return b * (g.W / NB) + eng_leaves_of_block(g, b);
}
template <typename T>
RFLU_HD int eng_left_units(const EngGeo& g, int cb, int lk)
{
    const int cb0 = cb * g.Wc;
    const int nc = (cb0 + g.Wc < g.n ? cb0 + g.Wc : g.n) - cb0;
    if (lk == 0) {
        if (eng_leaves_of_block(g, eng_pb(g, cb)) <= 1) return 0;
        return (nc + NB - 1) / NB;   // (a strip whose leaf is the block column's last has nothing to receive: its unit returns at once)
    }
    constexpr int SC4 = 4 * 8 * (16 / (int)sizeof(T));   // four wave strips of one 128-byte line per row
    return (nc + SC4 - 1) / SC4;
}

RFLU_HD int eng_units_of(const EngOp& o, int stage, int m)
{
    if (o.nc <= 0) return 0;
    if (stage == 0) return (o.nc + ENG_PREP_COLS - 1) / ENG_PREP_COLS;
    const int rows = m - (o.j0 + o.jb);
    if (rows <= 0) return 0;
    return ((rows + 127) / 128) * ((o.nc + 127) / 128);
}

// ---- order between the interchanges and the readers of a panel ------------------------------------------------------------------
// The leaves of block column b are applied one by one (LEAF ops on the column blocks of block columns b and b + 1) with the rows of L
// in the order of THAT leaf; the block column as a whole (BIG(b), block columns b + 2 ...) needs L with all of the block column's
// interchanges applied (left op 0 of every column block of b), and the interchanges of later block columns may permute L(b) only
// when nobody reads it any more:
//   left op 0 of b     after  every LEAF op of block column b is complete (column blocks of b and of b + 1)
//   BIG(b) anywhere    after  left op 0 on every column block of b              (own count of b == eng_cbs_of_block(b))
//   left op >= 1 of b  after  BIG(b) is complete on every column block that has it   (bigdone[b] == eng_big_users(b))
// The per-block-column counters live in the entry of the block column's FIRST column block: `bigdone`, and `leftdone` = (column
// blocks of this block column whose left op 0 is complete) << 32 | (column blocks to the left that have received this block
// column's interchanges).
RFLU_HD bool eng_big_waits_for_left(const EngGeo& g, int b) { return g.pivot && eng_leaves_of_block(g, b) > 1; }
RFLU_HD int eng_big_users(const EngGeo& g, int b)
{
    const int v = g.ncb - eng_first_cb(g, b + eng_ahead(g) + 1);
    return (b < g.nbp && v > 0) ? v : 0;
}
// ops of cb up to and including its LEAF ops of block column b (a block column in front of cb's own)
RFLU_HD int eng_ops_through_block(const EngGeo& g, int cb, int b) { return eng_nbig(g, cb) + eng_nleaf_front(g, cb, b + 1); }

template <typename T>
struct EngArgs {
    T* R;
    int64_t ld;
    EngGeo g;
    int policy;     // 0: leftmost column block first; 1: oldest panel piece first
    const T* linv;  // inverses of the 64x64 diagonal blocks, one per 64 rows
    const int* pm_cnt;
    const int* pm_dst;
    const int* pm_src;
    EngState* st;
    const unsigned long long* leaf_gate;   // the critical-path stream's counter: gate_base + (leaves completed incl. their lookahead launch)
    unsigned long long gate_base;
    int64_t* info;  // info[1] bit 0: timeout
    // host entry (driver.cpp: getrf_host_engine): the matrix arrives block column by block column while the factorization runs, and
    // finished block rows leave the same way
    const unsigned long long* arrived;     // columns [0, *arrived) of R are in place (nullptr: all of them)
    unsigned long long* rows_final;        // host-visible word: rows [0, *rows_final) of the factors are final (nullptr: nobody asks)
    int gemm_flags;
    long long* trace;   // measurement (RFLU_ENGINE_TRACE): per leaf g four wall-clock stamps of LEAF(g) on the column block of its first columns
    // (named defaults, Tune: RFLU_ENGINE_WRITE_THROUGH / _LEAF_XCDS / _LEAF_WGS / _HOST_LAG)
    int write_through;   // Schur tiles stored write-through (sc1): no agent-scope release -- an L2 write-back -- behind a tile
    int leaf_xcds;       // the workgroups with blockIdx % 8 in [1, leaf_xcds] and blockIdx / 8 < leaf_wgs serve the leaf windows (K = 64) only
    int leaf_wgs;
    int solve_rl;        // block-row solves of up to 512 rows right-looking with the block row in registers (engine.hip: eng_prep_unit)
    int host_lag;        // host entry: whole-block-column operations that lag the chain by this many block columns go first (0: never)
    int retire_xcc; // the workgroups on this XCC leave once retire_leaf leaves are done (-1: nobody retires): the short panels at the end, which
    int retire_leaf; // the engine has little to do for, get the XCD their XCD-local exchange needs (driver.cpp: factor_leafwise)
};

template <typename T>
int launch_engine(Handle* h, hipStream_t stream, const EngArgs<T>& a, int wgs);
int launch_eng_wait(Handle* h, const unsigned long long* flag, unsigned long long value);
int launch_eng_wait_retired(Handle* h, int xcc);   // the engine's workgroups on that XCC have all left
size_t engine_lds_bytes(size_t esize);

}  // namespace rflu
