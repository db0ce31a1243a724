#!/bin/bash
# rocprofv3 evidence for the default bench workload: kernel-trace stats (two runs: the default two-stream schedule) and three
# separate PMC passes (counters never combined with hip/hsa/sys tracing).  Output: gpurun_out/prof_<tag>/...
set -e
TAG=${1:-r03}
SIZE=${2:-16384}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
CMD="python bench.py --size $SIZE --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-extras"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -- $CMD > $OUT/bench_trace.json 2>$OUT/trace.err
DB=$(find $OUT/trace -name "*.db" | head -1)
python scripts/rocpd_summary.py $DB > $OUT/kernel_stats.txt
python scripts/rocpd_queues.py $DB 1 >> $OUT/kernel_stats.txt
echo "# the profiled single-stream pass (4th factorization of the run): bench.py's roofline.avg_launch_ms is the gemm_sub_kernel average of THIS pass" >> $OUT/kernel_stats.txt
python scripts/rocpd_queues.py $DB 3 >> $OUT/kernel_stats.txt
# counter collection runs ONE kernel at a time across all queues: device-side gates cannot make progress, so the counter passes ask
# for the schedule whose cross-stream edges are hipEvents (same kernels, same shapes for the bulk GEMM)
export RFLU_SCHEDULE_PMC=events
CMD1="env RFLU_SCHEDULE=events python bench.py --size $SIZE --steps 1 --warmup 0 --no-cpu-baseline --no-check --no-extras"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/pmc_fetch -- $CMD1 > /dev/null 2>$OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $OUT/pmc_write -- $CMD1 > /dev/null 2>$OUT/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES -f csv -d $OUT/pmc_mfma -- $CMD1 > /dev/null 2>$OUT/pmc_mfma.err
python scripts/pmc_summary.py $(find $OUT/pmc_fetch -name "*counter_collection.csv") $(find $OUT/pmc_write -name "*counter_collection.csv") $(find $OUT/pmc_mfma -name "*counter_collection.csv") > $OUT/pmc.txt
python scripts/rocpd_blocks.py $DB 1 > $OUT/blocks.txt 2>/dev/null || true
python scripts/rocpd_blocks.py $DB 2 > $OUT/blocks2.txt 2>/dev/null || true
python scripts/rocpd_queues.py $DB 2 > $OUT/queues2.txt 2>/dev/null || true
python scripts/rocpd_leaves.py $DB 1 8 0:4096 > $OUT/leaves.txt 2>/dev/null || true
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma   # keep the summaries only (gpurun_out is size-limited)
cat $OUT/kernel_stats.txt | head -30
cat $OUT/pmc.txt | grep -E "gemm|==" 
