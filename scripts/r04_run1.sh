#!/bin/bash
# round 4, first run of the new leaf kernels: bit-identity + per-column times, a slice of the GPU suite, the size table
cd "$(dirname "$0")/.."
O=gpurun_out/r04; mkdir -p $O
PANEL_MODES=0,2,1 python scripts/panel_bench.py 64 128 256 512 1024 4096 8192 16384 > $O/run1_panel_bench.txt 2>&1
PANEL_F32=1 PANEL_MODES=0,2 python scripts/panel_bench.py 256 512 4096 16384 > $O/run1_panel_bench_f32.txt 2>&1
for d in 0 400 700 1000 1400; do echo "== RFLU_POLL_DELAY=$d RFLU_POLL_ADAPT=0"; RFLU_POLL_DELAY=$d RFLU_POLL_ADAPT=0 PANEL_MODES=2 python scripts/panel_bench.py 4096 8192 16384; done > $O/run1_poll_delay.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_lu.py tests/test_gpu_kernels.py -x -q -m gpu > $O/run1_pytest.txt 2>&1
for n in 1024 2048 4096 8192 16384; do python bench.py --size $n --steps 5 --warmup 1 --no-cpu-baseline --no-extras; done > $O/run1_sizes.txt 2>&1
tail -n 40 $O/run1_panel_bench.txt; tail -n 12 $O/run1_panel_bench_f32.txt; cat $O/run1_poll_delay.txt | grep -v amdgpu.ids; tail -n 5 $O/run1_pytest.txt; grep -o '"ms_per_step": [0-9.]*' $O/run1_sizes.txt
