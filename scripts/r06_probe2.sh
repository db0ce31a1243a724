#!/bin/bash
# round 6: the engine's two-round-trip scheduling loop -- parity, liveness, time, accounting
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06p2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_host_entry.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
timeout 300 python scripts/engine_stress.py 8192 300 > $O/stress.txt 2>&1; tail -2 $O/stress.txt
timeout 300 python scripts/engine_stress.py 5000 200 >> $O/stress.txt 2>&1; tail -1 $O/stress.txt
timeout 300 python scripts/engine_stress.py 16384 100 >> $O/stress.txt 2>&1; tail -1 $O/stress.txt
timeout 600 python scripts/time_env.py 16384 5 "" "RFLU_ENGINE=0" "RFLU_ENGINE_RETIRE=0" "RFLU_ENGINE_RETIRE=4096" "RFLU_ENGINE_LEAF_WGS=8" "RFLU_ENGINE_LEAF_WGS=32" > $O/time_env.txt 2>&1; cat $O/time_env.txt
timeout 300 python scripts/time_env.py 12288 4 "RFLU_ENGINE=1" "RFLU_ENGINE=0" >> $O/time_env.txt 2>&1; tail -2 $O/time_env.txt
timeout 300 python scripts/time_env.py 8192 4 "RFLU_ENGINE=1" "RFLU_ENGINE=0" >> $O/time_env.txt 2>&1; tail -2 $O/time_env.txt
RFLU_ENGINE_TRACE=1 timeout 300 python scripts/time_env.py 16384 2 "" > $O/engine_trace.txt 2>&1; grep "rflu\]\|leaf " $O/engine_trace.txt | tail -8
python bench.py --size 16384 --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_n16384.json 2>$O/bench.err; grep -o '"ms_per_step": [0-9.]*' $O/bench_n16384.json
