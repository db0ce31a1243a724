#!/bin/bash
# librflu_trace.so = librflu with the RFLU_PANEL_TRACE clock stamps compiled in (experiments only; see scripts/panel_trace.py)
set -e
cd "$(dirname "$0")/../recursivefactorization.jl_amd/csrc"
mkdir -p build_trace
for f in gemm.hip panel.hip panel_f32.hip panel_local.hip panel_local_f32.hip trsm.hip trsv.hip laswp.hip butterfly.hip driver.cpp; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -Wno-unused-result -DRFLU_PANEL_TRACE -c $f -o build_trace/${f%.*}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../librflu_trace.so build_trace/*.o
