"""Per-queue (stream) breakdown of one factorization in a rocprofv3 rocpd trace of bench.py (step index via argv[2])."""
import sqlite3, collections, re, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
which = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = cur.execute("select name, start, end, queue_id, grid_x, workgroup_x from kernels order by start").fetchall()
fills = [i for i, r in enumerate(rows) if 'fill_uniform' in r[0]]   # bench.py refills the input before every lu!
a = fills[which] + 1
b = fills[which + 1] - 1 if which + 1 < len(fills) else len(rows) - 1
while 'transpose' not in rows[b][0]: b -= 1
seg = rows[a:b + 1]
t0, t1 = seg[0][1], seg[-1][2]
print("factorization wall: %.2f ms" % ((t1 - t0) / 1e6))
short = lambda n: re.sub(r"<.*", "", re.sub(r"\(.*", "", n.replace("void rflu::", "")))
byq = collections.defaultdict(list)
for n, s, e, q, gx, wx in seg: byq[q].append((s, e, short(n), gx // max(wx, 1)))
for q, lst in byq.items():
    busy = sum(e - s for s, e, _, _ in lst)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for s, e, n, g in lst:
        agg[n][0] += 1; agg[n][1] += (e - s) / 1e6
    gaps = sum(max(0, lst[i + 1][0] - lst[i][1]) for i in range(len(lst) - 1))
    print(f"queue {q}: {len(lst)} kernels, busy {busy/1e6:.2f} ms, gaps {gaps/1e6:.2f} ms")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"    {n:24s} calls {c:5d} total {t:8.2f} ms avg {1e3*t/c:8.1f} us")
    if len(sys.argv) > 3:
        g2 = collections.defaultdict(lambda: [0, 0.0])
        for s, e, n, g in lst:
            if n == 'gemm_sub_kernel': g2[g][0] += 1; g2[g][1] += (e - s) / 1e6
        for g, (c, t) in sorted(g2.items(), key=lambda kv: -kv[1][1])[:12]:
            print(f"        gemm tiles={g:5d} calls {c:4d} total {t:7.2f} ms avg {1e3*t/c:8.1f} us")
