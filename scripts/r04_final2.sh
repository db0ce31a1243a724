#!/bin/bash
# the rocprofv3 part of scripts/r04_final.sh alone, and the Float32 A/B of the 384-row workgroups
TAG=${1:-r04c}
cd "$(dirname "$0")/.."
O=gpurun_out/r04; mkdir -p $O
bash scripts/collect_profiles.sh $TAG 16384 > $O/collect_16384.log 2>&1
bash scripts/collect_profiles.sh ${TAG}_n4096 4096 > $O/collect_4096.log 2>&1
tail -3 $O/collect_16384.log
B="python bench.py --warmup 1 --no-cpu-baseline --no-extras --no-check --steps 5"
for rep in 1 2; do for v in 0 1; do echo -n "f32 n=16384 RFLU_PANEL_SPARE=$v: "; RFLU_PANEL_SPARE=$v $B --size 16384 --dtype f32 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; done; done
echo -n "f32 n=16384 LEAFWISE_ROWS default vs SPARE_MIN=16384 (off for f32): "; RFLU_PANEL_SPARE_MIN=16384 $B --size 16384 --dtype f32 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
for v in 0 1; do echo -n "f64 n=16384 RFLU_PANEL_SPARE=$v: "; RFLU_PANEL_SPARE=$v $B --size 16384 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; done
