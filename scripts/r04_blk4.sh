#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r04; mkdir -p $O
{
echo "== Float64"; PANEL_MODES=2,4 timeout 300 python scripts/panel_bench.py 64 100 448 449 1000 2048 4096 8192 14336 28672
echo "== Float64, XCD-local forced"; PANEL_MODES=5 timeout 300 python scripts/panel_bench.py 1024 4096
echo "== Float32"; PANEL_F32=1 PANEL_MODES=2,4 timeout 300 python scripts/panel_bench.py 64 513 2048 8192 16384
RFLU_PANEL_LOCAL_ROWS=0 timeout 300 python scripts/panel_blocked_trace.py 448 4096 14336
} 2>&1 | grep -v amdgpu.ids > $O/blk4_panel.txt
cat $O/blk4_panel.txt
export RFLU_PANEL_LOCAL_ROWS=0
for n in 2048 4096 8192 16384; do echo -n "nolocal n=$n "; timeout 120 python bench.py --size $n --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-check 2>$O/blk_err_$n.txt | grep -o '"ms_per_step": [0-9.]*'; grep -v amdgpu.ids $O/blk_err_$n.txt | tail -2; done > $O/blk4_sizes.txt 2>&1
cat $O/blk4_sizes.txt
timeout 900 python -m pytest tests/test_gpu_lu.py tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -5
