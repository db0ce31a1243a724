#!/bin/bash
# librflu_gemmtrace.so = librflu with per-workgroup wall-clock stamps in gemm_sub_kernel (experiments only; scripts/gemm_phase_trace.py)
set -e
cd "$(dirname "$0")/../recursivefactorization.jl_amd/csrc"
mkdir -p build_gemmtrace
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -Wno-unused-result -DRFLU_GEMM_TRACE $EXTRA -c gemm.hip -o build_gemmtrace/gemm.o
objs=""
for f in build/*.o; do b=$(basename $f); [ "$b" = gemm.o ] || objs="$objs $f"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../librflu_gemmtrace.so build_gemmtrace/gemm.o $objs
