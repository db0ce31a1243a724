#!/bin/bash
# round 5: everything the documents quote, from ONE box: size table, default bench line, microbenches, rocprofv3 summaries.
# usage (on the GPU box): bash scripts/r05_final.sh <tag>      then, here: python scripts/install_profiles.py <tag>
TAG=${1:-r05a}
cd "$(dirname "$0")/.."
O=gpurun_out/r05; mkdir -p $O
rm -f $O/bench_n*.json
B="python bench.py --warmup 1 --no-cpu-baseline --no-extras"
$B --size 4096 --steps 5 > $O/bench_n4096.json 2>/dev/null
$B --size 8192 --steps 5 > $O/bench_n8192.json 2>/dev/null
$B --size 12288 --steps 5 > $O/bench_n12288.json 2>/dev/null
$B --size 16384 --steps 5 > $O/bench_n16384.json 2>/dev/null
$B --size 16384 --steps 5 --dtype f32 > $O/bench_n16384_f32.json 2>/dev/null
$B --size 16384 --steps 5 --nopivot > $O/bench_n16384_nopivot.json 2>/dev/null
$B --size 32768 --steps 3 > $O/bench_n32768.json 2>/dev/null
$B --size 65536 --steps 3 --no-check > $O/bench_n65536.json 2>/dev/null
$B --size 65536 --steps 3 --no-check --dtype f32 > $O/bench_n65536_f32.json 2>/dev/null
python bench.py --steps 8 --warmup 2 > $O/bench_default.json 2>$O/bench_default.err
python scripts/microbench_gemm_sustained.py > $O/gemm_sustained.txt 2>&1
python scripts/microbench_gemm_sustained.py 15872 512 f32 > $O/gemm_sustained_f32.txt 2>&1
python scripts/microbench_gemm_k.py > $O/gemm_k.txt 2>&1
python scripts/microbench_laswp.py > $O/laswp_alone.txt 2>&1
python scripts/microbench_host_entry.py > $O/host_entry.txt 2>&1
PANEL_MODES=2 python scripts/panel_bench.py 64 512 1024 2048 4096 8192 12288 16384 > $O/panel_bench.txt 2>&1
for n in 4096 16384; do python scripts/microbench_getrs.py $n; done > $O/getrs.txt 2>&1
python scripts/getrs_check.py > $O/getrs_block.txt 2>&1
python scripts/engine_check.py time > $O/engine_time.txt 2>&1
PANEL_MODES=2,1 python scripts/panel_bench.py 4096 6144 8192 12288 16384 > $O/panel_bench_local.txt 2>&1
bash scripts/collect_profiles.sh $TAG 16384 > $O/collect_16384.log 2>&1
bash scripts/collect_profiles.sh ${TAG}_n4096 4096 > $O/collect_4096.log 2>&1
grep -o '"ms_per_step": [0-9.]*' $O/bench_n*.json $O/bench_default.json
tail -3 $O/collect_16384.log
