#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06p8; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_engine.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for r in 1 2; do
timeout 400 python scripts/time_env.py 16384 4 "" "RFLU_ENGINE_AHEAD=1" "RFLU_ENGINE_AHEAD=3" "RFLU_ENGINE_AHEAD=2,RFLU_ENGINE_RETIRE=4096" "RFLU_ENGINE_AHEAD=2,RFLU_ENGINE_RETIRE=0" "RFLU_ENGINE=0" >> $O/time_env.txt 2>&1
done; grep -v amdgpu.ids $O/time_env.txt
timeout 300 python scripts/time_env.py 12288 4 "RFLU_ENGINE=1" "RFLU_ENGINE=1,RFLU_ENGINE_AHEAD=1" "RFLU_ENGINE=0" > $O/time_12288.txt 2>&1; grep -v amdgpu.ids $O/time_12288.txt
timeout 300 python scripts/time_env.py 8192 4 "RFLU_ENGINE=1" "RFLU_ENGINE=1,RFLU_ENGINE_AHEAD=1" "RFLU_ENGINE=0" > $O/time_8192.txt 2>&1; grep -v amdgpu.ids $O/time_8192.txt
timeout 300 python scripts/engine_stress.py 8192 200 > $O/stress.txt 2>&1; tail -1 $O/stress.txt
timeout 300 python scripts/engine_stress.py 16384 60 >> $O/stress.txt 2>&1; tail -1 $O/stress.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $O/trace -- python bench.py --size 16384 --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-extras > $O/bench_trace.json 2>$O/trace.err
DB=$(find $O/trace -name "*.db" | head -1)
python scripts/rocpd_leaves.py $DB 1 8 0:256 > $O/leaves.txt 2>&1
rm -rf $O/trace
head -36 $O/leaves.txt; awk '/^leaf:/{f=1;next} f && $6+0 > 60 {print}' $O/leaves.txt | head -50
