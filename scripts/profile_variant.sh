#!/bin/bash
# kernel-trace stats of one bench variant: scripts/profile_variant.sh <tag> <bench args...>   (summary only, raw trace removed)
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-extras "$@" > $OUT/bench_trace.json 2>$OUT/trace.err
DB=$(find $OUT/trace -name "*.db" | head -1)
python scripts/rocpd_summary.py $DB > $OUT/kernel_stats.txt
python scripts/rocpd_queues.py $DB 1 >> $OUT/kernel_stats.txt
rm -rf $OUT/trace
cat $OUT/kernel_stats.txt
