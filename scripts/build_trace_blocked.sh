#!/bin/bash
# librflu_trace.so = librflu with the RFLU_PANEL_TRACE clock stamps compiled into the sub-panel leaf only (scripts/panel_blocked_trace.py)
cd "$(dirname "$0")/../recursivefactorization.jl_amd/csrc"
mkdir -p build_trace
for f in panel_blocked.hip panel.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -Wno-unused-result -DRFLU_PANEL_TRACE -c $f -o build_trace/${f%.*}.o &
done
wait
objs=""
for o in build/*.o; do b=$(basename $o); if [ -f build_trace/$b ] && { [ $b = panel_blocked.o ] || [ $b = panel.o ]; }; then objs="$objs build_trace/$b"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../librflu_trace.so $objs && echo built ../librflu_trace.so
