#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06p9; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_engine.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for r in 1 2; do
timeout 600 python scripts/time_env.py 16384 4 "" "RFLU_ENGINE_STICKY=0" "RFLU_ENGINE_LEAF_WGS=32" "RFLU_ENGINE_LEAF_WGS=56" "RFLU_ENGINE_LEAF_XCDS=2,RFLU_ENGINE_LEAF_WGS=32" "RFLU_ENGINE_LEAF_XCDS=0" "RFLU_ENGINE_AHEAD=2" "RFLU_ENGINE=0" >> $O/time_env.txt 2>&1
done; grep -v amdgpu.ids $O/time_env.txt
timeout 300 python scripts/time_env.py 12288 4 "RFLU_ENGINE=1" "RFLU_ENGINE=1,RFLU_ENGINE_STICKY=0" "RFLU_ENGINE=0" > $O/time_12288.txt 2>&1; grep -v amdgpu.ids $O/time_12288.txt
timeout 300 python scripts/time_env.py 8192 4 "RFLU_ENGINE=1" "RFLU_ENGINE=1,RFLU_ENGINE_STICKY=0" "RFLU_ENGINE=0" > $O/time_8192.txt 2>&1; grep -v amdgpu.ids $O/time_8192.txt
timeout 300 python scripts/engine_stress.py 8192 300 > $O/stress.txt 2>&1; tail -1 $O/stress.txt
timeout 300 python scripts/engine_stress.py 5000 200 >> $O/stress.txt 2>&1; tail -1 $O/stress.txt
timeout 300 python scripts/engine_stress.py 16384 100 >> $O/stress.txt 2>&1; tail -1 $O/stress.txt
