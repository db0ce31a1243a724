#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06p11; mkdir -p $O
for r in 1 2; do
timeout 600 python scripts/time_env.py 16384 4 "" "RFLU_ENGINE_WC=256" "RFLU_ENGINE_WC=128" "RFLU_ENGINE_WC=256,RFLU_ENGINE_LEAF_WGS=32" "RFLU_ENGINE_WC=128,RFLU_ENGINE_LEAF_WGS=32" "RFLU_ENGINE_WC=256,RFLU_ENGINE_AHEAD=2" >> $O/time_env.txt 2>&1
done; grep -v amdgpu.ids $O/time_env.txt
