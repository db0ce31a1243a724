#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06p5; mkdir -p $O
timeout 300 python scripts/time_env.py 16384 2 "RFLU_ENGINE_TRACE=1" > $O/engine_trace.txt 2>&1; grep "rflu\]" $O/engine_trace.txt | tail -4
RFLU_ENGINE_TRACE=1 timeout 200 python scripts/engine_replay.py 16384 3 > $O/replay_trace.txt 2>&1; grep "rflu\]\|engine_ms" $O/replay_trace.txt | tail -9
