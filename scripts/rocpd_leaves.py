"""The chain of leaves, leaf by leaf, from a rocprofv3 rocpd trace of bench.py: per block column the average duration of the leaf kernel,
of the lookahead launch behind it (leaf_la_kernel: its own ~21 us of work + the folded wait for the update engine / side stream) and of
the K = 64 update of the next leaf's columns, the leaf-to-leaf period, and what the period leaves unexplained (launch gaps).
usage: python scripts/rocpd_leaves.py trace.db [step index] [leaves per block column]"""
import sqlite3, re, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
which = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lpb = int(sys.argv[3]) if len(sys.argv) > 3 else 8
rows = cur.execute("select name, start, end, queue_id from kernels order by start").fetchall()
fills = [i for i, r in enumerate(rows) if 'fill_uniform' in r[0]]
a = fills[which] + 1
b = fills[which + 1] - 1 if which + 1 < len(fills) else len(rows) - 1
while 'transpose' not in rows[b][0]: b -= 1
seg = rows[a:b + 1]
t0, t1 = seg[0][1], seg[-1][2]
short = lambda n: re.sub(r"<.*", "", re.sub(r"\(.*", "", n.replace("void rflu::", "")))
panel_q = next(q for n, s, e, q in seg if 'panel_' in n)
chain = [(short(n), s, e) for n, s, e, q in seg if q == panel_q]
leaves = []   # (start of the leaf, leaf us, la us, skinny us, other us)
cur_leaf = None
for n, s, e in chain:
    d = (e - s) / 1e3
    if n.startswith('panel_') and 'nopivot_rows' not in n:
        if cur_leaf: leaves.append(cur_leaf)
        cur_leaf = [s, d, 0.0, 0.0, 0.0]
    elif cur_leaf:
        if 'leaf_la' in n or 'laswp' in n: cur_leaf[2] += d
        elif 'gemm_skinny' in n or 'gemm_sub' in n: cur_leaf[3] += d
        else: cur_leaf[4] += d
if cur_leaf: leaves.append(cur_leaf)
print("factorization wall: %.2f ms, %d leaves on queue %d" % ((t1 - t0) / 1e6, len(leaves), panel_q))
print("block column: leaves | leaf us | lookahead launch us (work + folded wait) | K=64 update us | other kernels us | period us | gaps us | block column ms")
tot = [0.0] * 6
for i in range(0, len(leaves), lpb):
    blk = leaves[i:i + lpb]
    nxt = leaves[i + lpb][0] if i + lpb < len(leaves) else t1
    per = (nxt - blk[0][0]) / 1e3 / len(blk)
    av = [sum(x[k] for x in blk) / len(blk) for k in (1, 2, 3, 4)]
    gaps = per - sum(av)
    print("  %3d: %d | %6.1f | %6.1f | %5.1f | %5.1f | %6.1f | %5.1f | %.2f" % (i // lpb, len(blk), av[0], av[1], av[2], av[3], per, gaps, per * len(blk) / 1e3))
    for k in range(4): tot[k] += av[k] * len(blk)
    tot[4] += per * len(blk); tot[5] += gaps * len(blk)
if len(sys.argv) > 4:   # per-leaf rows for leaves [a, b)
    a_, b_ = (int(x) for x in sys.argv[4].split(":"))
    print("leaf: start ms | leaf us | lookahead launch us | K=64 update us | period us")
    for g in range(a_, min(b_, len(leaves))):
        nxt = leaves[g + 1][0] if g + 1 < len(leaves) else t1
        print("  %3d: %7.3f | %6.1f | %6.1f | %5.1f | %6.1f" % (g, (leaves[g][0] - t0) / 1e6, leaves[g][1], leaves[g][2], leaves[g][3], (nxt - leaves[g][0]) / 1e3))
print("sums (ms): leaf %.2f, lookahead launch %.2f, K=64 update %.2f, other %.2f, periods %.2f, gaps %.2f" % tuple(x / 1e3 for x in tot))
