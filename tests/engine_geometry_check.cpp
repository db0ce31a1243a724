// Host-side check of the persistent update engine's operation lists (csrc/engine.hpp -- the functions the host builds the initial state with and
// the device walks): over a grid of geometries, every column receives every earlier leaf exactly once and in order (the Schur updates of
// /root/reference/src/lu.jl:233-240, :265-284 in the engine's decomposition; the leaf right in front of a column reaches it on the critical-path
// stream), the leaf counts an operation waits for are sufficient and monotone, an operation's columns lie inside its column block, a leaf's
// operation index is where eng_leaf_op_index says, and a stage with columns has units.  Compiled and run by tests/test_engine_geometry.py (no GPU).
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "engine.hpp"
using namespace rflu;
int main(int argc, char** argv)
{
    int bad = 0, cases = 0;
    const int Ws[] = {128, 256, 512};
    for (int W : Ws)
        for (int Wc : {128, 256, 512}) {
            if (Wc > W || W % Wc) continue;
            for (int m : {1024, 2048, 2112, 5000, 6144}) for (int n : {1024, 2048, 2112, 5000, 6144}) {
                if (m < n && m % W) continue;   // (engine_usable)
                EngGeo g{};
                g.m = m; g.n = n; g.mn = m < n ? m : n; g.W = W; g.Wc = Wc; g.pivot = 1;
                const int nblk = (g.mn + W - 1) / W;
                g.ncb = (n + Wc - 1) / Wc;
                if (g.ncb > ENG_MAX_CB) continue;
                for (int ahead : {1, 2, 3})
                for (int nbp : {nblk, nblk > 2 ? nblk / 2 : nblk}) {
                    g.nbp = nbp;
                    g.ahead = ahead;   // leaf windows over the own block column and `ahead` block columns right of it (engine.hpp)
                    ++cases;
                    const int served_leaves = (std::min(nbp * W, g.mn) + NB - 1) / NB;
                    // the per-block-column counts the interchange ordering rests on (engine.hpp: "order between the interchanges ..."): who has
                    // BIG(b), and where the LEAF ops of block column b end in the lists of the column blocks in front of which it lies
                    for (int b = 0; b < nbp; ++b) {
                        int users = 0;
                        for (int cb = 0; cb < g.ncb; ++cb) users += eng_nbig(g, cb) > b;
                        if (users != eng_big_users(g, b)) { if (bad++ < 10) printf("big users W=%d Wc=%d m=%d n=%d ahead=%d b=%d: %d vs %d\n", W, Wc, m, n, ahead, b, users, eng_big_users(g, b)); }
                        for (int cb = 0; cb < g.ncb; ++cb) {
                            const int pb = eng_pb(g, cb);
                            int last = -1;   // last op of cb that applies a leaf of block column b leaf by leaf
                            for (int k = 0; k < eng_nops(g, cb); ++k) {
                                const EngOp o = eng_op(g, cb, k);
                                if (o.type == ENG_OP_LEAF && o.j0 / W == b) last = k;
                            }
                            if (pb > b && pb <= b + ahead) {
                                if (eng_ops_through_block(g, cb, b) != last + 1 && !(last < 0 && eng_leaves_of_block(g, b) == 0)) { if (bad++ < 10) printf("ops through block W=%d Wc=%d m=%d n=%d ahead=%d b=%d cb=%d: %d vs %d\n", W, Wc, m, n, ahead, b, cb, eng_ops_through_block(g, cb, b), last + 1); }
                            } else if (pb != b && last >= 0) { if (bad++ < 10) printf("leaf op outside the window W=%d Wc=%d m=%d n=%d ahead=%d b=%d cb=%d\n", W, Wc, m, n, ahead, b, cb); }
                        }
                    }
                    for (int cb = 0; cb < g.ncb; ++cb) {
                        const int c_first = cb * Wc, c_last = std::min(n, c_first + Wc);
                        std::vector<std::vector<int>> got(c_last - c_first);
                        int prev_need = 0, prev_j0 = -1;
                        for (int k = 0; k < eng_nops(g, cb); ++k) {
                            const EngOp o = eng_op(g, cb, k);
                            if (o.need < prev_need) { if (bad++ < 10) printf("need not monotone W=%d Wc=%d m=%d n=%d cb=%d k=%d\n", W, Wc, m, n, cb, k); }
                            prev_need = o.need;
                            if (o.j0 <= prev_j0) { if (bad++ < 10) printf("pivot blocks out of order W=%d Wc=%d m=%d n=%d cb=%d k=%d\n", W, Wc, m, n, cb, k); }
                            prev_j0 = o.j0;
                            if (o.need * NB < o.j0 + o.jb && o.need < served_leaves) { if (bad++ < 10) printf("need too small W=%d Wc=%d m=%d n=%d cb=%d k=%d need=%d j0=%d jb=%d\n", W, Wc, m, n, cb, k, o.need, o.j0, o.jb); }
                            if (o.type == ENG_OP_LEAF && eng_leaf_op_index(g, cb, o.j0 / NB) != k) { if (bad++ < 10) printf("leaf op index W=%d Wc=%d m=%d n=%d cb=%d k=%d\n", W, Wc, m, n, cb, k); }
                            if (o.nc > 0 && eng_units_of(o, 0, g.m) <= 0) { if (bad++ < 10) printf("no stage-0 units W=%d Wc=%d m=%d n=%d cb=%d k=%d\n", W, Wc, m, n, cb, k); }
                            if (o.nc > 0 && g.m > o.j0 + o.jb && eng_units_of(o, 1, g.m) <= 0) { if (bad++ < 10) printf("no stage-1 units W=%d Wc=%d m=%d n=%d cb=%d k=%d\n", W, Wc, m, n, cb, k); }
                            if (o.nc <= 0) continue;
                            if (o.c_lo < c_first || o.c_lo + o.nc > c_last) { if (bad++ < 10) printf("columns outside the block W=%d Wc=%d m=%d n=%d cb=%d k=%d\n", W, Wc, m, n, cb, k); continue; }
                            for (int c = o.c_lo; c < o.c_lo + o.nc; ++c)
                                for (int j = o.j0; j < o.j0 + o.jb; j += NB) got[c - c_first].push_back(j / NB);
                        }
                        for (int c = c_first; c < c_last; ++c) {
                            // column c (leaf lc) owes the engine the leaves 0 .. min(lc - 1, served) - 1: leaf lc - 1 reaches the 64 columns behind
                            // it on the critical-path stream (the lookahead strip)
                            const int lc = c / NB;
                            int want = std::min(lc - 1, served_leaves);
                            if (want < 0) want = 0;
                            const std::vector<int>& v = got[c - c_first];
                            bool ok = (int)v.size() == want;
                            for (int i = 0; ok && i < want; ++i) ok = v[i] == i;
                            if (!ok && bad++ < 10) {
                                printf("W=%d Wc=%d m=%d n=%d nbp=%d column %d (leaf %d): wants leaves 0..%d, got %zu:", W, Wc, m, n, nbp, c, lc, want - 1, v.size());
                                for (size_t i = 0; i < v.size() && i < 12; ++i) printf(" %d", v[i]);
                                printf("\n");
                            }
                        }
                    }
                }
            }
        }
    printf("%d geometries, %d violations\n", cases, bad);
    return bad ? 1 : 0;
}
