#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06p15; mkdir -p $O
for n in 8192 10240 11264 12288 14336; do
timeout 300 python scripts/time_env.py $n 4 "" "BS=384,RFLU_ENGINE=1" "BS=512,RFLU_ENGINE=1" "BS=384,RFLU_ENGINE=0" "BS=256,RFLU_ENGINE=1" >> $O/time_env.txt 2>&1
done; grep -v amdgpu.ids $O/time_env.txt
