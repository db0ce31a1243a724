#!/bin/bash
cd "$(dirname "$0")/.."
export RFLU_PANEL_LOCAL_ROWS=0
run() { python bench.py --size $1 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-check 2>&1 | grep -o '"ms_per_step": [0-9.]*\|timed out.*' | head -1; }
for n in 2560 3072 3584 4096; do echo -n "n=$n: "; run $n; echo; done
echo -n "4096 LEAFWISE=0: "; RFLU_LEAFWISE=0 run 4096; echo
echo -n "4096 SCHEDULE=events: "; RFLU_SCHEDULE=events run 4096; echo
echo -n "4096 blocksize -1 (pure recursion, one stream): "; python bench.py --size 4096 --blocksize -1 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-check 2>&1 | grep -o '"ms_per_step": [0-9.]*\|timed out.*' | head -1; echo
echo -n "4096 MAXG... reserve 64: "; RFLU_RESERVE_CUS=64 run 4096; echo
