#!/bin/bash
# one rocprofv3 kernel trace of bench.py at $1 (size) -> timeline of the panel kernels [$2, $2+$3) of the P queue
set -e
SIZE=${1:-16384}; FIRST=${2:-160}; COUNT=${3:-8}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/timeline; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace -d $OUT/trace -- python bench.py --size $SIZE --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-extras > $OUT/bench.json 2>$OUT/trace.err
DB=$(find $OUT/trace -name "*.db" | head -1)
python scripts/rocpd_timeline.py $DB 1 $FIRST $COUNT > $OUT/timeline.txt
python scripts/rocpd_queues.py $DB 1 > $OUT/queues.txt
python scripts/rocpd_timeline.py $DB 1 0 40 U > $OUT/timeline_U.txt
python scripts/rocpd_timeline.py $DB 1 0 16 > $OUT/timeline_P0.txt
rm -rf $OUT/trace
tail -12 $OUT/timeline.txt
