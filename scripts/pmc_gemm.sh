#!/bin/bash
# PMC passes over the GEMM micro-benchmark (own kernel vs vendor BLAS): where do the cycles of the K loop go?
# usage: scripts/pmc_gemm.sh <script> <args...>   (e.g. scripts/microbench_gemm_k.py 15872)
set -e
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_gemm
rm -rf $OUT; mkdir -p $OUT
run() { rocprofv3 --kernel-trace --pmc "$@" -f csv -d $OUT/p -- python $SCRIPT $ARGS > /dev/null 2>$OUT/err.txt; python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("$OUT/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if "gemm" in k.lower() or "Cijk" in k:
        print(k, {c: f"{v:.4g}" for c, v in d.items()}, "launches", max(cnt[(k, c)] for c in d))
PY
rm -rf $OUT/p; }
SCRIPT=$1; shift; ARGS="$@"
run SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY
run SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM
run SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU
