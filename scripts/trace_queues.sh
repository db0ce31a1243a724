#!/bin/bash
# rocprofv3 kernel trace of bench.py at one size, then the per-queue breakdown of one timed factorization with the bulk GEMM
# launches listed by tile count (scripts/rocpd_queues.py): usage scripts/trace_queues.sh <size> [extra bench args]
set -e
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
SIZE=${1:-16384}; shift || true
OUT=gpurun_out/trace_queues; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace -d $OUT/trace -- python bench.py --size $SIZE --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-extras "$@" > $OUT/stdout.txt 2>$OUT/trace.err || true
DB=$(find $OUT/trace -name "*.db" | head -1)
python scripts/rocpd_queues.py $DB 1 x > $OUT/queues_$SIZE.txt 2>&1 || true
python scripts/rocpd_blocks.py $DB 1 > $OUT/blocks_$SIZE.txt 2>&1 || true
python scripts/rocpd_summary.py $DB > $OUT/summary_$SIZE.txt 2>&1 || true
rm -rf $OUT/trace
cat $OUT/queues_$SIZE.txt
