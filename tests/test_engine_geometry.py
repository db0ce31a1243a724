"""No GPU: the operation lists of the persistent update engine (csrc/engine.hpp: eng_op, eng_nops, eng_units_of, eng_leaf_op_index -- the
same functions the host builds the engine's initial state with and the device walks) cover the Schur updates of the factorization
(/root/reference/src/lu.jl:233-240, :265-284) exactly once and in order, for a grid of shapes, block widths and column-block widths.
The checker is host C++ (tests/engine_geometry_check.cpp), compiled here with g++ against the HIP headers."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None or not os.path.isdir("/opt/rocm/include"), reason="needs g++ and the HIP headers")
def test_engine_operation_lists_cover_every_update_once(tmp_path):
    exe = str(tmp_path / "engine_geometry_check")
    cmd = ["g++", "-std=c++17", "-O1", "-w", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "recursivefactorization.jl_amd", "csrc"),
           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "engine_geometry_check.cpp"), "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=300)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "0 violations" in r.stdout
