#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r04; mkdir -p $O
export RFLU_PANEL_LOCAL_ROWS=0
{
for n in 2048 4096 8192 16384; do echo -n "nolocal n=$n "; python bench.py --size $n --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-check 2>$O/blk_err_$n.txt | grep -o '"ms_per_step": [0-9.]*'; grep -v amdgpu.ids $O/blk_err_$n.txt | tail -2; done
for n in 2048 4096 8192 16384; do echo -n "nolocal f32 n=$n "; python bench.py --size $n --dtype float32 --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-check 2>$O/blk_err_$n.txt | grep -o '"ms_per_step": [0-9.]*'; grep -v amdgpu.ids $O/blk_err_$n.txt | tail -2; done
export RFLU_PANEL_BLOCKED=0
for n in 2048 4096 8192 16384; do echo -n "old n=$n "; python bench.py --size $n --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-check 2>$O/blk_err_$n.txt | grep -o '"ms_per_step": [0-9.]*'; grep -v amdgpu.ids $O/blk_err_$n.txt | tail -2; done
} > $O/blk2_sizes.txt 2>&1
cat $O/blk2_sizes.txt
