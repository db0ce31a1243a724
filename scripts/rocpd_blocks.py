"""Progress of one factorization in a rocprofv3 rocpd trace of bench.py: when every block column's first leaf starts on the
critical-path queue, and how busy every queue is in windows of 5 ms (overlap of the pivot chain with the bulk update).
usage: python scripts/rocpd_blocks.py trace.db [step index] [leaves per block column]"""
import sqlite3, collections, re, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
which = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lpb = int(sys.argv[3]) if len(sys.argv) > 3 else 8
rows = cur.execute("select name, start, end, queue_id, grid_x, workgroup_x from kernels order by start").fetchall()
fills = [i for i, r in enumerate(rows) if 'fill_uniform' in r[0]]   # bench.py refills the input before every lu!
a = fills[which] + 1
b = fills[which + 1] - 1 if which + 1 < len(fills) else len(rows) - 1
while 'transpose' not in rows[b][0]: b -= 1
seg = rows[a:b + 1]
t0, t1 = seg[0][1], seg[-1][2]
print("factorization wall: %.2f ms" % ((t1 - t0) / 1e6))
short = lambda n: re.sub(r"<.*", "", re.sub(r"\(.*", "", n.replace("void rflu::", "")))
panels = [(s, e) for n, s, e, q, gx, wx in seg if 'panel_pivot' in n]
print("block column: first leaf starts at (ms) / time since the previous block column (ms)")
prev = None
line = []
for i in range(0, len(panels), lpb):
    t = (panels[i][0] - t0) / 1e6
    line.append("%2d:%6.2f(+%.2f)" % (i // lpb, t, t - prev if prev is not None else 0.0))
    prev = t
for i in range(0, len(line), 6): print("   " + "  ".join(line[i:i + 6]))
byq = collections.defaultdict(list)
for n, s, e, q, gx, wx in seg: byq[q].append((s, e, short(n)))
WIN = 5e6
nwin = int((t1 - t0) / WIN) + 1
for q, lst in sorted(byq.items()):
    busy = [0.0] * nwin; wait = [0.0] * nwin; gemm = [0.0] * nwin
    for s, e, n in lst:
        for w in range(int((s - t0) / WIN), min(nwin - 1, int((e - t0) / WIN)) + 1):
            lo, hi = max(s, t0 + w * WIN), min(e, t0 + (w + 1) * WIN)
            if hi > lo:
                (wait if 'gate_wait' in n else busy)[w] += (hi - lo)
                if n == 'gemm_sub_kernel': gemm[w] += (hi - lo)
    print(f"queue {q}: busy % per 5 ms window (gate waits excluded): " + " ".join("%3d" % round(100 * x / WIN) for x in busy))
    print(f"          of which gemm_sub_kernel:                       " + " ".join("%3d" % round(100 * x / WIN) for x in gemm))
    if any(wait): print(f"          gate_wait %:                                    " + " ".join("%3d" % round(100 * x / WIN) for x in wait))
