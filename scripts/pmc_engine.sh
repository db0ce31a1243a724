#!/bin/bash
# rocprofv3 counters of the resident engine_kernel, replayed ALONE (scripts/engine_replay.py): three separate --pmc passes with
# --kernel-trace only, plus the replay's own timing without a profiler.  usage (GPU box): bash scripts/pmc_engine.sh <tag> [n]
TAG=${1:-r06}; N=${2:-16384}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$TAG; mkdir -p $OUT
python scripts/engine_replay.py $N 3 > $OUT/engine_replay.txt 2>&1
CMD="python scripts/engine_replay.py $N 1"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/epmc_fetch -- $CMD > $OUT/epmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $OUT/epmc_write -- $CMD > $OUT/epmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES -f csv -d $OUT/epmc_mfma -- $CMD > $OUT/epmc_mfma.log 2>&1
python scripts/pmc_summary.py $(find $OUT/epmc_fetch -name "*counter_collection.csv") $(find $OUT/epmc_write -name "*counter_collection.csv") $(find $OUT/epmc_mfma -name "*counter_collection.csv") 2>&1 | grep -E "^==|engine_kernel" > $OUT/engine_pmc.txt
for d in epmc_fetch epmc_write epmc_mfma; do python scripts/rocpd_summary.py $(find $OUT/$d -name "*.db" | head -1) 2>/dev/null | grep -E "engine_kernel" >> $OUT/engine_pmc_durations.txt; done
rm -rf $OUT/epmc_fetch $OUT/epmc_write $OUT/epmc_mfma
cat $OUT/engine_replay.txt | tail -4; cat $OUT/engine_pmc.txt
