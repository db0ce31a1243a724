#!/bin/bash
# round 4: the sub-panel leaf (panel_blocked.hip) against the shipped leaves: bit identity and time per column, then the parity tests
cd "$(dirname "$0")/.."
O=gpurun_out/r04; mkdir -p $O
{
echo "== Float64"; PANEL_MODES=2,4 timeout 600 python scripts/panel_bench.py 64 100 448 449 512 1000 2048 4096 5000 8192 14336 16384 28672
echo "== Float64, XCD-local forced"; PANEL_MODES=2,5 timeout 300 python scripts/panel_bench.py 1024 4096 8192 14336
echo "== Float32"; PANEL_F32=1 PANEL_MODES=2,4 timeout 600 python scripts/panel_bench.py 64 513 2048 8192 16384
} 2>&1 | grep -v amdgpu.ids > $O/blk_panel.txt
cat $O/blk_panel.txt
timeout 900 python -m pytest tests/test_gpu_lu.py tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -15 > $O/blk_pytest.txt
cat $O/blk_pytest.txt
for n in 2048 4096 8192 16384; do echo -n "n=$n "; python bench.py --size $n --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-check 2>$O/blk_err_$n.txt | grep -o '"ms_per_step": [0-9.]*'; grep -v amdgpu.ids $O/blk_err_$n.txt | tail -3; done > $O/blk_sizes.txt 2>&1
cat $O/blk_sizes.txt
