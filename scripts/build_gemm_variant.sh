#!/bin/bash
# librflu_<name>.so = librflu with gemm.hip (or the source given as $3) compiled with extra flags $2 (experiments only)
# usage: scripts/build_gemm_variant.sh name "-DRFLU_GEMM_TRACE -DRFLU_GEMM_GLOAD_AT=0" [source]
set -e
NAME=$1; EXTRA=$2; SRC=${3:-gemm.hip}
cd "$(dirname "$0")/../recursivefactorization.jl_amd/csrc"
mkdir -p build_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -Wno-unused-result $EXTRA -c $SRC -o build_variants/gemm_$NAME.o
objs=""
for f in build/*.o; do b=$(basename $f); [ "$b" = gemm.o ] || objs="$objs $f"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../librflu_$NAME.so build_variants/gemm_$NAME.o $objs
