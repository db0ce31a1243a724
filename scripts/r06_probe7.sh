#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06p7; mkdir -p $O
for r in 1 2; do
timeout 300 python scripts/time_env.py 16384 4 "" "RFLU_ENGINE_BUNDLE_LEAF=2" "RFLU_ENGINE=0" >> $O/time_env.txt 2>&1
RFLU_LIB=$GRAFT_REPO_ROOT/recursivefactorization.jl_amd/librflu_oldloop.so timeout 300 python scripts/time_env.py 16384 4 "" >> $O/time_env.txt 2>&1
done; grep -v amdgpu.ids $O/time_env.txt
timeout 300 python scripts/engine_stress.py 8192 200 > $O/stress.txt 2>&1; tail -1 $O/stress.txt
timeout 300 python scripts/engine_stress.py 16384 60 >> $O/stress.txt 2>&1; tail -1 $O/stress.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $O/trace -- python bench.py --size 16384 --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-extras > $O/bench_trace.json 2>$O/trace.err
DB=$(find $O/trace -name "*.db" | head -1)
python scripts/rocpd_leaves.py $DB 1 8 0:48 > $O/leaves.txt 2>&1
python scripts/rocpd_leaves.py $DB 1 8 176:256 | grep -A100 "^leaf:" > $O/leaves_tail.txt 2>&1
rm -rf $O/trace
head -60 $O/leaves.txt | tail -52
