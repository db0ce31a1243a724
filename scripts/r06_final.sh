#!/bin/bash
# round 6: everything the documents quote, from ONE box: size table, default bench line, microbenches, rocprofv3 summaries, the engine's
# counters (replayed alone), the chain leaf by leaf, the engine's workgroup-time table.
# usage (on the GPU box): bash scripts/r06_final.sh <tag>      then, here: python scripts/install_profiles.py <tag>
TAG=${1:-r06d}
cd "$(dirname "$0")/.."
O=gpurun_out/r06; mkdir -p $O
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
python scripts/time_env.py 16384 4 "" "RFLU_ENGINE=0" "RFLU_ENGINE_AHEAD=2" "RFLU_ENGINE_RETIRE=0" "RFLU_ENGINE_RETIRE=4096" > $O/engine_time.txt 2>&1
python scripts/time_env.py 16384 4 f32 "" "RFLU_ENGINE=0" >> $O/engine_time.txt 2>&1
python scripts/time_env.py 16384 4 f64 0 "" "RFLU_ENGINE=1" >> $O/engine_time.txt 2>&1
python scripts/time_env.py 12288 4 "" "RFLU_ENGINE=0,BS=256" "RFLU_ENGINE=0" >> $O/engine_time.txt 2>&1
python scripts/time_env.py 12288 4 f32 "" "RFLU_ENGINE=0,BS=256" >> $O/engine_time.txt 2>&1
python scripts/time_env.py 11264 4 "" "RFLU_ENGINE=1,BS=512" >> $O/engine_time.txt 2>&1
python scripts/time_env.py 8192 4 "" "RFLU_ENGINE=1,BS=512" >> $O/engine_time.txt 2>&1
python scripts/tall_panel.py > $O/tall_panel.txt 2>&1
python scripts/time_env.py 16384 3 "RFLU_ENGINE_TRACE=72:24" > $O/engine_trace.txt 2>&1
python scripts/engine_stress.py 16384 200 > $O/engine_stress.txt 2>&1
python scripts/engine_stress.py 8192 500 >> $O/engine_stress.txt 2>&1
RFLU_ENGINE=-1 python scripts/engine_stress.py 12288 300 >> $O/engine_stress.txt 2>&1
python scripts/engine_stress.py 5000 300 >> $O/engine_stress.txt 2>&1
python scripts/host_entry_stress.py >> $O/engine_stress.txt 2>&1
bash scripts/collect_profiles.sh $TAG 16384 > $O/collect_16384.log 2>&1
bash scripts/collect_profiles.sh ${TAG}_n4096 4096 > $O/collect_4096.log 2>&1
bash scripts/pmc_engine.sh $TAG 16384 > $O/pmc_engine.log 2>&1
grep -o '"ms_per_step": [0-9.]*' $O/bench_n*.json $O/bench_default.json
tail -3 $O/collect_16384.log; tail -8 $O/pmc_engine.log
