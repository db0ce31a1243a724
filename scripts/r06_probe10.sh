#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06p10; mkdir -p $O
timeout 300 python scripts/time_env.py 16384 2 "RFLU_ENGINE_TRACE=32:24" > $O/trace_a.txt 2>&1; grep "rflu\]\|   leaf" $O/trace_a.txt | tail -27
timeout 300 python scripts/time_env.py 16384 2 "RFLU_ENGINE_TRACE=184:24" > $O/trace_b.txt 2>&1; grep "   leaf" $O/trace_b.txt | tail -24
