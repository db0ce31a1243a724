#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06p14; mkdir -p $O
for r in 1 2; do
timeout 600 python scripts/time_env.py 16384 4 "" "RFLU_ENGINE_SLACK=3" "RFLU_ENGINE_SLACK=5" "RFLU_ENGINE_SLACK=7" "RFLU_ENGINE_SLACK=10" "RFLU_ENGINE_SLACK=14" >> $O/time_env.txt 2>&1
done; grep -v amdgpu.ids $O/time_env.txt
timeout 300 python scripts/engine_stress.py 16384 60 > $O/stress.txt 2>&1; tail -1 $O/stress.txt
