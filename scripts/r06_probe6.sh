#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06p6; mkdir -p $O
{
echo "== replay (engine alone), bundles"
for cfg in "RFLU_ENGINE_BUNDLE_LEAF=1 RFLU_ENGINE_BUNDLE_BIG=1" "RFLU_ENGINE_BUNDLE_LEAF=4 RFLU_ENGINE_BUNDLE_BIG=1" "RFLU_ENGINE_BUNDLE_LEAF=8 RFLU_ENGINE_BUNDLE_BIG=1" "RFLU_ENGINE_BUNDLE_LEAF=4 RFLU_ENGINE_BUNDLE_BIG=2 RFLU_ENGINE_BUNDLE_FAR=-1000" "RFLU_ENGINE_BUNDLE_LEAF=4 RFLU_ENGINE_BUNDLE_BIG=4 RFLU_ENGINE_BUNDLE_FAR=-1000"; do
  echo "-- $cfg"; env $cfg timeout 200 python scripts/engine_replay.py 16384 2 2>&1 | grep engine_ms
done
} > $O/replay.txt 2>&1; cat $O/replay.txt
timeout 900 python scripts/time_env.py 16384 4 "" "RFLU_ENGINE_BUNDLE_LEAF=1,RFLU_ENGINE_BUNDLE_BIG=1" "RFLU_ENGINE_BUNDLE_LEAF=4,RFLU_ENGINE_BUNDLE_BIG=1" "RFLU_ENGINE_BUNDLE_LEAF=8,RFLU_ENGINE_BUNDLE_BIG=1" "RFLU_ENGINE_BUNDLE_LEAF=2,RFLU_ENGINE_BUNDLE_BIG=1" "RFLU_ENGINE_BUNDLE_BIG=2,RFLU_ENGINE_BUNDLE_FAR=2" "RFLU_ENGINE_BUNDLE_BIG=2,RFLU_ENGINE_BUNDLE_FAR=5" "RFLU_ENGINE_BUNDLE_BIG=3,RFLU_ENGINE_BUNDLE_FAR=4" "RFLU_ENGINE=0" > $O/time_env.txt 2>&1; cat $O/time_env.txt
RFLU_LIB=$GRAFT_REPO_ROOT/recursivefactorization.jl_amd/librflu_oldloop.so timeout 300 python scripts/time_env.py 16384 4 "" > $O/time_oldloop.txt 2>&1; cat $O/time_oldloop.txt
timeout 300 python scripts/time_env.py 16384 2 "RFLU_ENGINE_TRACE=1" > $O/engine_trace.txt 2>&1; grep "rflu\]" $O/engine_trace.txt | tail -4
timeout 300 python scripts/engine_stress.py 8192 200 > $O/stress.txt 2>&1; tail -1 $O/stress.txt
timeout 300 python scripts/engine_stress.py 16384 60 >> $O/stress.txt 2>&1; tail -1 $O/stress.txt
