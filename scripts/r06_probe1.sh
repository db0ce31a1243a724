#!/bin/bash
# round 6, first call: the build with the advisor's fixes -- engine tests, baseline, per-leaf profile of the chain, engine time accounting
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06p1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_host_entry.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
python bench.py --size 16384 --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_n16384.json 2>$O/bench.err; grep -o '"ms_per_step": [0-9.]*' $O/bench_n16384.json
timeout 600 python scripts/time_env.py 16384 4 "" "RFLU_ENGINE_RETIRE=0" "RFLU_ENGINE_RETIRE=4096" "RFLU_ENGINE_X6=8" "RFLU_ENGINE_X6=32" "RFLU_ENGINE_X5=2,RFLU_ENGINE_X6=8" "RFLU_ENGINE=0" > $O/time_env.txt 2>&1; cat $O/time_env.txt
RFLU_ENGINE_TRACE=1 timeout 300 python scripts/time_env.py 16384 2 "" > $O/engine_trace.txt 2>&1; grep "rflu\]\|leaf " $O/engine_trace.txt | tail -8
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $O/trace -- python bench.py --size 16384 --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-extras > $O/bench_trace.json 2>$O/trace.err
DB=$(find $O/trace -name "*.db" | head -1)
python scripts/rocpd_leaves.py $DB 1 > $O/leaves.txt 2>&1; cat $O/leaves.txt
python scripts/rocpd_summary.py $DB > $O/kernel_stats.txt 2>&1
python scripts/rocpd_queues.py $DB 1 >> $O/kernel_stats.txt 2>&1
rm -rf $O/trace
