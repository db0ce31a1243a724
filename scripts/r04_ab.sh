#!/bin/bash
# round 4: old (round-3 HEAD, scripts/probes/bin/librflu_r3.so) vs new library on the SAME box: leaf kernels alone and whole factorizations
cd "$(dirname "$0")/.."
O=gpurun_out/r04; mkdir -p $O
{
echo "== leaf alone, round-3 library"; RFLU_LIB=$PWD/scripts/probes/bin/librflu_r3.so PANEL_MODES=2 python scripts/panel_bench.py 512 1024 4096 8192 16384
echo "== leaf alone, this build";      PANEL_MODES=2 python scripts/panel_bench.py 512 1024 4096 8192 16384
echo "== leaf alone, this build, XCD-local forced"; PANEL_MODES=1 python scripts/panel_bench.py 4096 8192 16384
} 2>&1 | grep -v amdgpu.ids > $O/ab_panel.txt
for rep in 1 2; do
for lib in r3 new; do
  for n in 2048 4096 8192 16384; do
    if [ $lib = r3 ]; then export RFLU_LIB=$PWD/scripts/probes/bin/librflu_r3.so; else unset RFLU_LIB; fi
    echo -n "$lib n=$n "; python bench.py --size $n --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-check 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
  done
done
done > $O/ab_sizes.txt 2>&1
unset RFLU_LIB
{ for bs in 256 512; do for n in 4096 8192 12288; do echo -n "blocksize $bs n=$n "; python bench.py --size $n --blocksize $bs --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-check 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; done; done; } > $O/ab_blocksize.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_host_entry.py tests/test_gpu_lu.py -x -q -m gpu > $O/ab_pytest.txt 2>&1
cat $O/ab_panel.txt $O/ab_sizes.txt $O/ab_blocksize.txt; tail -5 $O/ab_pytest.txt
