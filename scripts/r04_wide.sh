#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r04; mkdir -p $O
run() { python bench.py --size $1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | grep -o '"ms_per_step": [0-9.]*\|"residual[^,]*\|ipiv[^,]*\|Error.*' | tr '\n' ' '; echo; }
{
for n in 20480 24576 32768; do
  echo -n "n=$n default: "; run $n
  echo -n "n=$n RFLU_WIDE_NARROW=0: "; RFLU_WIDE_NARROW=0 run $n
done
echo -n "n=32768 narrow 12288: "; RFLU_NARROW_COLS=12288 run 32768
echo -n "n=32768 narrow 20480: "; RFLU_NARROW_COLS=20480 run 32768
echo -n "n=65536 default: "; run 65536
echo -n "n=65536 RFLU_WIDE_NARROW=0: "; RFLU_WIDE_NARROW=0 run 65536
} > $O/wide_sizes.txt 2>&1
cat $O/wide_sizes.txt
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "32768 or config3 or config4 or 65536" 2>&1 | tail -4
