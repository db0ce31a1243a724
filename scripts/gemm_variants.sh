#!/bin/bash
# compares the GEMM builds of scripts/build_gemm_variant.sh on one box: checksums, sustained rates, lone / shared workgroup efficiency
cd $GRAFT_REPO_ROOT
L=$PWD/recursivefactorization.jl_amd
for v in ${VARIANTS:-v0 v1 v2 v3 v4}; do
  echo "=== $v"
  RFLU_LIB=$L/librflu_$v.so python scripts/gemm_checksum.py 2>&1 | grep -v amdgpu.ids | md5sum
  RFLU_LIB=$L/librflu_$v.so python scripts/microbench_gemm_sustained.py 15872 512 2>&1 | grep -E "^ (20|60)"
  RFLU_LIB=$L/librflu_$v.so python scripts/microbench_gemm_sustained.py 15360 256 2>&1 | grep -E "^ 60"
  RFLU_LIB=$L/librflu_$v.so python scripts/microbench_gemm_sustained.py 15360 1024 2>&1 | grep -E "^ 20"
done
for v in ${TVARIANTS:-t1 t2 t3 t4}; do
  echo "=== $v"
  export RFLU_LIB=$L/librflu_$v.so WARM=6
  SN=2048 python scripts/gemm_phase_trace.py 2048 512 2>&1 | grep -E "^S=|prologue|shader clocks"
  SN=4096 python scripts/gemm_phase_trace.py 2048 512 2>&1 | grep -E "^S=|prologue|shader clocks"
  python scripts/gemm_phase_trace.py 15872 512 2>&1 | grep -E "^S=|prologue|shader clocks|CU-time|stores"
done
