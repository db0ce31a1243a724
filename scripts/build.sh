#!/bin/bash
# builds librflu.so in-tree (incremental)
cd "$(dirname "$0")/.." && python -c "
import importlib.util
spec=importlib.util.spec_from_file_location('b','recursivefactorization.jl_amd/build.py'); m=importlib.util.module_from_spec(spec); spec.loader.exec_module(m); print(m.build_librflu(verbose=False))"
