#!/bin/bash
# librflu_trace_all.so = librflu with wall-clock stamps from thread 0 of EVERY panel workgroup (scripts/panel_skew_trace.py)
set -e
cd "$(dirname "$0")/../recursivefactorization.jl_amd/csrc"
mkdir -p build_trace_all
for f in gemm.hip panel.hip panel_f32.hip panel_local.hip panel_local_f32.hip trsm.hip trsv.hip laswp.hip butterfly.hip driver.cpp; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -Wno-unused-result -DRFLU_PANEL_TRACE_ALL -c $f -o build_trace_all/${f%.*}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../librflu_trace_all.so build_trace_all/*.o
