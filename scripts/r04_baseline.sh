#!/bin/bash
# round-4 baseline of the leaf kernel on this box: per-column times, phase stamps, skew, latency probes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r04
O=gpurun_out/r04
PANEL_MODES=2 python scripts/panel_bench.py 512 1024 4096 8192 16384 > $O/base_panel_bench.txt 2>&1
RFLU_PANEL_LOCAL=2 python scripts/panel_local_trace.py 512 4096 16384 > $O/base_panel_trace.txt 2>&1
python scripts/panel_skew_trace.py 16384 > $O/base_panel_skew.txt 2>&1
scripts/probes/bin/latprobe > $O/base_latprobe.txt 2>&1
scripts/probes/bin/xcdlocal > $O/base_xcdlocal.txt 2>&1
for n in 4096 8192 16384; do python bench.py --size $n --steps 5 --warmup 1 --no-cpu-baseline --no-extras; done > $O/base_sizes.txt 2>&1
tail -n 30 $O/base_panel_bench.txt $O/base_panel_trace.txt $O/base_panel_skew.txt $O/base_latprobe.txt $O/base_sizes.txt
