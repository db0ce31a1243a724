#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r04; mkdir -p $O
{
echo "== RFLU_PANEL_SPARE=0 (round-3 routing)"; RFLU_PANEL_SPARE=0 PANEL_MODES=0,2 timeout 300 python scripts/panel_bench.py 4608 6144 8192 10240 12288
echo "== 384-row workgroups forced (RFLU_PANEL_RPW=384, any placement)"; RFLU_PANEL_LOCAL_ROWS=0 RFLU_PANEL_RPW=384 PANEL_MODES=0,2 timeout 300 python scripts/panel_bench.py 2048 4608 6144 8192 10240 12288
echo "== default"; PANEL_MODES=2 timeout 300 python scripts/panel_bench.py 4608 6144 8192 10240 12288
} 2>&1 | grep -v amdgpu.ids > $O/spare_panel.txt
cat $O/spare_panel.txt
for v in 0 1; do for n in 8192 12288 16384; do echo -n "RFLU_PANEL_SPARE=$v n=$n "; RFLU_PANEL_SPARE=$v python bench.py --size $n --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-check 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; done; done > $O/spare_sizes.txt 2>&1
for n in 8192 12288 16384; do echo -n "RFLU_PANEL_SPARE_MIN=4096 n=$n "; RFLU_PANEL_SPARE_MIN=4096 python bench.py --size $n --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-check 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; done >> $O/spare_sizes.txt 2>&1
cat $O/spare_sizes.txt
