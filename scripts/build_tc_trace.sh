#!/bin/bash
# librflu with the block solve's stage stamps compiled in (csrc/trsv.hip, -DRFLU_TC_TRACE): every wide rflu_getrs call prints, per
# triangle, the publish-to-publish times of a few blocks, what the owner of a block did between the previous publish and its own, and
# where a bystander workgroup spends a stage.  usage: scripts/build_tc_trace.sh && RFLU_LIB=$PWD/build_trace/librflu_trace.so python scripts/getrs_check.py
set -e
cd "$(dirname "$0")/.."
bash scripts/build.sh > /dev/null
mkdir -p build_trace
O=recursivefactorization.jl_amd/csrc/build
/opt/rocm/lib/llvm/bin/clang++ --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -Wno-unused-result -DRFLU_TC_TRACE \
    -c recursivefactorization.jl_amd/csrc/trsv.hip -o build_trace/trsv.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_trace/librflu_trace.so $(ls $O/*.o | grep -v "/trsv.o") build_trace/trsv.o
echo build_trace/librflu_trace.so
