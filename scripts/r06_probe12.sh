#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06p12; mkdir -p $O
timeout 600 python scripts/time_env.py 16384 3 "" "BS=256,RFLU_ENGINE=1" "BS=384,RFLU_ENGINE=1" "BS=256,RFLU_ENGINE=0" "BS=384,RFLU_ENGINE=0" > $O/time_bs.txt 2>&1; grep -v amdgpu.ids $O/time_bs.txt
timeout 600 python scripts/time_env.py 16384 3 f32 "" "RFLU_ENGINE=1" > $O/time_f32.txt 2>&1; grep -v amdgpu.ids $O/time_f32.txt
timeout 600 python scripts/time_env.py 16384 3 f64 0 "" "RFLU_ENGINE=1" > $O/time_nopiv.txt 2>&1; grep -v amdgpu.ids $O/time_nopiv.txt
timeout 600 python scripts/time_env.py 12288 3 "" "BS=512,RFLU_ENGINE=1" "BS=512,RFLU_ENGINE=0" "BS=384,RFLU_ENGINE=1" > $O/time_12288.txt 2>&1; grep -v amdgpu.ids $O/time_12288.txt
