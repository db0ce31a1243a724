#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06p16; mkdir -p $O
timeout 600 python scripts/time_env.py 16384 3 "" "BS=768,RFLU_ENGINE=1" "BS=1024,RFLU_ENGINE=1" "BS=640,RFLU_ENGINE=1" > $O/time_env.txt 2>&1; grep -v amdgpu.ids $O/time_env.txt
