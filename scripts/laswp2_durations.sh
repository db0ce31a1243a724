#!/bin/bash
# rocprofv3 kernel trace of one bench run -> durations of the per-leaf interchange launches (laswp_kernel with 2 workgroups) and of the
# two kernels behind them on the critical-path queue
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/lw2_$1; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace -d $OUT/trace -- python bench.py --size 16384 --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-extras > $OUT/bench.json 2>$OUT/trace.err
DB=$(find $OUT/trace -name "*.db" | head -1)
python - "$DB" <<'PY'
import sqlite3, sys, statistics
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = cur.execute("select name, start, end, queue_id, grid_x, workgroup_x from kernels order by start").fetchall()
for key, cond in (("laswp 2 wgs", lambda r: 'laswp_kernel' in r[0] and r[4] // max(r[5], 1) == 2), ("laswp 8 wgs", lambda r: 'laswp_kernel' in r[0] and r[4] // max(r[5], 1) == 8),
                  ("trsm_inv64 1 wg", lambda r: 'trsm_inv64' in r[0] and r[4] // max(r[5], 1) == 1), ("gemm_skinny", lambda r: 'gemm_skinny' in r[0])):
    d = sorted((r[2] - r[1]) / 1e3 for r in rows if cond(r))
    if d: print(f"{key:18s} n={len(d):5d}  p10 {d[len(d)//10]:6.1f}  median {statistics.median(d):6.1f}  p90 {d[9*len(d)//10]:6.1f} us")
PY
rm -rf $OUT/trace
