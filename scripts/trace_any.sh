#!/bin/bash
# rocprofv3 kernel trace of an arbitrary command ($@): prints the longest kernels' individual durations
set -e
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/trace_any; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace -d $OUT/trace -- "$@" > $OUT/stdout.txt 2>$OUT/trace.err
DB=$(find $OUT/trace -name "*.db" | head -1)
python - $DB <<'PY'
import sqlite3, sys, re
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, start, end, queue_id, grid_x, workgroup_x from kernels order by start").fetchall()
big = [r for r in rows if r[4] // max(r[5], 1) > 3000]
for r in big[-26:]:
    print(f"{(r[2]-r[1])/1e3:9.1f} us  wgs {r[4]//r[5]:6d}  queue {r[3]}  {re.sub(r'<.*','',r[0])[:40]}")
PY
rm -rf $OUT/trace
