#!/bin/bash
# samples the shader clock / power while the large-K GEMM microbenchmark runs (is the fp64 MFMA peak clock-limited?)
python scripts/microbench_gemm_sustained.py > gpurun_out/clk_gemm.log 2>&1 &
PID=$!
sleep 6
for i in $(seq 1 30); do
  /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo
  sleep 0.25
  kill -0 $PID 2>/dev/null || break
done
wait $PID
tail -5 gpurun_out/clk_gemm.log
