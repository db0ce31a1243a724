"""Busy spans / idle gaps per queue for one factorization of a rocprofv3 rocpd trace of bench.py.
usage: rocpd_timeline.py trace.db [step] [gap_us]"""
import sqlite3, re, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
which = int(sys.argv[2]) if len(sys.argv) > 2 else 1
gap_us = float(sys.argv[3]) if len(sys.argv) > 3 else 150.0
rows = cur.execute("select name, start, end, queue_id, grid_x, workgroup_x from kernels order by start").fetchall()
tr = [i for i, r in enumerate(rows) if 'transpose' in r[0]]
a, b = tr[2 * which], tr[2 * which + 1]
seg = rows[a:b + 1]
t0 = seg[0][1]
short = lambda n: re.sub(r"<.*", "", re.sub(r"\(.*", "", n.replace("void rflu::", "")))
byq = collections.defaultdict(list)
for n, s, e, q, gx, wx in seg: byq[q].append((s, e, short(n), gx // max(wx, 1)))
for q, lst in sorted(byq.items()):
    print(f"== queue {q}: {len(lst)} kernels")
    span_s = lst[0][0]; prev_e = lst[0][1]; names = collections.Counter(); names[lst[0][2]] += 1
    for s, e, n, g in lst[1:]:
        if s - prev_e > gap_us * 1e3:
            print(f"  busy {(span_s-t0)/1e6:8.2f} .. {(prev_e-t0)/1e6:8.2f} ms ({(prev_e-span_s)/1e6:6.2f})  then idle {(s-prev_e)/1e6:6.2f} ms   {dict(names.most_common(3))}")
            span_s = s; names = collections.Counter()
        names[n] += 1
        prev_e = max(prev_e, e)
    print(f"  busy {(span_s-t0)/1e6:8.2f} .. {(prev_e-t0)/1e6:8.2f} ms ({(prev_e-span_s)/1e6:6.2f})  end   {dict(names.most_common(3))}")
