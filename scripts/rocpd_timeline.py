"""Kernel-by-kernel timeline of one stretch of the critical-path queue in a rocprofv3 rocpd trace of bench.py:
python scripts/rocpd_timeline.py <db> <factorization index> <first panel-kernel ordinal> <count>  (durations / gaps in us)."""
import sqlite3, re, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
which, first, count = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
rows = cur.execute("select name, start, end, queue_id, grid_x, workgroup_x from kernels order by start").fetchall()
fills = [i for i, r in enumerate(rows) if 'fill_uniform' in r[0]]   # bench.py refills the input before every lu!
a0 = fills[which] + 1
b0 = fills[which + 1] - 1 if which + 1 < len(fills) else len(rows) - 1
while 'transpose' not in rows[b0][0]: b0 -= 1
seg = rows[a0:b0 + 1]
short = lambda n: re.sub(r"<.*", "", re.sub(r"\(.*", "", n.replace("void rflu::", "")))
import collections
qP = collections.Counter(r[3] for r in seg if 'panel_pivot' in r[0]).most_common(1)[0][0]   # the critical-path queue
if len(sys.argv) > 5 and sys.argv[5] == 'U':   # the update queue instead: ordinals count its laswp kernels
    qP = collections.Counter(r[3] for r in seg if 'gemm_sub' in r[0] and r[3] != qP).most_common(1)[0][0]
P = [r for r in seg if r[3] == qP]
others = [r for r in seg if r[3] != qP]
if len(sys.argv) > 5 and sys.argv[5] == 'U':
    for i, r in enumerate(P[:int(sys.argv[4])]):
        prev = P[i - 1][2] if i else r[1]
        print(f"{(r[1] - seg[0][1]) / 1e3:9.1f} us  gap {(r[1] - prev) / 1e3:7.1f}  dur {(r[2] - r[1]) / 1e3:7.1f}  wgs {r[4] // max(r[5], 1):6d}  {short(r[0])}")
    sys.exit(0)
idx = [i for i, r in enumerate(P) if 'panel_pivot' in r[0]]
a = idx[first]; b = idx[min(first + count, len(idx) - 1)]
t0 = P[a][1]
prev_end = P[a - 1][2] if a > 0 else t0
busy = gaps = 0.0
agg = {}
for r in P[a:b]:
    n, s, e = short(r[0]), r[1], r[2]
    ov = sum(max(0, min(e, o[2]) - max(s, o[1])) for o in others if o[2] > s and o[1] < e)
    print(f"{(s - t0) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f}  wgs {r[4] // max(r[5], 1):5d}  U-overlap {ov / max(e - s, 1):4.2f}  {n}")
    busy += (e - s) / 1e3; gaps += (s - prev_end) / 1e3
    x = agg.setdefault(n, [0, 0.0, 0.0]); x[0] += 1; x[1] += (e - s) / 1e3; x[2] += (s - prev_end) / 1e3
    prev_end = e
print(f"stretch: {(P[b][1] - t0) / 1e3:.1f} us, busy {busy:.1f}, gaps {gaps:.1f}")
for n, (c, d, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"   {n:28s} x{c:3d}  busy {d:8.1f}  avg {d / c:7.1f}   gap before (avg) {g / c:6.1f}")
