#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06p3; mkdir -p $O
timeout 300 python scripts/time_env.py 16384 3 "RFLU_ENGINE_TRACE=1" > $O/engine_trace.txt 2>&1; grep "rflu\]" $O/engine_trace.txt | tail -6
timeout 300 python scripts/time_env.py 8192 3 "RFLU_ENGINE_TRACE=1,RFLU_ENGINE=1" > $O/engine_trace_8192.txt 2>&1; grep "rflu\]" $O/engine_trace_8192.txt | tail -3
timeout 900 bash scripts/pmc_engine.sh r06a 16384 > $O/pmc_engine.log 2>&1; tail -12 $O/pmc_engine.log
