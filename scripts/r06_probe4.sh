#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06p4; mkdir -p $O
for p in 0 4 128 132 64 32 68; do echo "== replay, RFLU_ENGINE_POLICY=$p"; RFLU_ENGINE_POLICY=$p timeout 200 python scripts/engine_replay.py 16384 2 2>&1 | grep engine_ms; done > $O/replay_policy.txt 2>&1; cat $O/replay_policy.txt
timeout 600 python scripts/time_env.py 16384 3 "" "RFLU_ENGINE_POLICY=4" "RFLU_ENGINE_POLICY=128" "RFLU_ENGINE_POLICY=132" "RFLU_ENGINE_POLICY=64" "RFLU_ENGINE_POLICY=68" "RFLU_ENGINE_POLICY=32" > $O/time_env.txt 2>&1; cat $O/time_env.txt
timeout 300 python scripts/time_env.py 16384 2 "RFLU_ENGINE_TRACE=1,RFLU_ENGINE_POLICY=132" > $O/engine_trace.txt 2>&1; grep "rflu\]" $O/engine_trace.txt | tail -3
