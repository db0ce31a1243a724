"""Summarise a rocprofv3 rocpd .db (kernel trace): per-kernel calls / total / avg, plus busy time and idle gaps."""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
def short(n):
    n = re.sub(r"\(.*", "", n); n = n.replace("void ", "").replace("rflu::", "")
    return n[:70]
agg = {}
for n, s, e in rows:
    a = agg.setdefault(short(n), [0, 0])
    a[0] += 1; a[1] += e - s
tot = sum(v[1] for v in agg.values())
print(f"{'kernel':72s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'%':>6s}")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:72s} {c:7d} {t/1e6:10.3f} {t/c/1e3:9.2f} {100*t/tot:6.2f}")
span = rows[-1][2] - rows[0][1]
gaps = sum(max(0, rows[i+1][1] - rows[i][2]) for i in range(len(rows)-1))
print(f"kernels busy {tot/1e6:.3f} ms; first-start..last-end {span/1e6:.3f} ms; sum of gaps between consecutive kernels {gaps/1e6:.3f} ms")
