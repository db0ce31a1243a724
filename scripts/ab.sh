#!/bin/bash
# A/B of environment settings on one box, alternating: scripts/ab.sh "<size> [bench args]" reps "ENV_A" "ENV_B" ...  (ms_per_step of bench.py)
cd $GRAFT_REPO_ROOT
ARGS=$1; REPS=$2; shift 2
for r in $(seq $REPS); do
  for cfg in "$@"; do
    v=$(env $cfg timeout 200 python bench.py --size $ARGS --steps 5 --warmup 1 --no-cpu-baseline --no-check --no-extras 2>&1 | grep -oE "\"ms_per_step\": [0-9.]+|status [0-9].*" | tr "\n" " ")
    echo "[$cfg] size $ARGS: $v"
  done
done
