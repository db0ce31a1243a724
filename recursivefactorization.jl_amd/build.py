"""Builds librflu.so (hand-written HIP for gfx950) in-tree with hipcc.  No JIT cache, no other targets."""
from __future__ import annotations

import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build_exp" if os.environ.get("RFLU_EXPERIMENTS", "0") not in ("", "0") else "build")
LIB = os.path.join(HERE, "librflu_exp.so" if os.environ.get("RFLU_EXPERIMENTS", "0") not in ("", "0") else "librflu.so")
# RFLU_EXPERIMENTS=1 in the environment of the BUILD adds the kernels that were measured and lost (the sub-panel
# leaf of round 4) -- objects and library of their own (build_exp/, librflu_exp.so) so that the default build never contains them
EXPERIMENTS = os.environ.get("RFLU_EXPERIMENTS", "0") not in ("", "0")
SOURCES = ["gemm.hip", "engine.hip", "panel.hip", "panel_f32.hip", "panel_local.hip", "panel_local_f32.hip", "panel_local_xcd.hip", "panel_local_xcd_f32.hip", "panel_single.hip", "panel_single_f32.hip", "trsm.hip", "trsv.hip", "laswp.hip", "butterfly.hip", "driver.cpp"]
if EXPERIMENTS:
    SOURCES += ["panel_blocked.hip", "panel_blocked_f32.hip"]
HEADERS = ["rflu_internal.hpp", os.path.join("..", "..", "include", "rflu.h")]   # included by every source
ALL_HEADERS = HEADERS + ["panel_common.hpp", "panel_xchg.hpp", "trsm_row.hpp", "gemm_tile.hpp", "laswp_strip.hpp", "engine.hpp"]
_PANEL_H = ["panel_common.hpp", "panel_xchg.hpp", "trsm_row.hpp"]
EXTRA_DEPS = {"gemm.hip": ["gemm_tile.hpp"], "engine.hip": ["gemm_tile.hpp", "laswp_strip.hpp", "engine.hpp"], "driver.cpp": ["engine.hpp"],
              "laswp.hip": ["laswp_strip.hpp", "trsm_row.hpp"], "trsm.hip": ["trsm_row.hpp"], "trsv.hip": ["trsm_row.hpp"],
              "panel.hip": _PANEL_H, "panel_local.hip": _PANEL_H, "panel_single.hip": _PANEL_H, "panel_blocked.hip": _PANEL_H,
              "panel_f32.hip": ["panel.hip", *_PANEL_H], "panel_local_f32.hip": ["panel_local.hip", *_PANEL_H],
              "panel_local_xcd.hip": ["panel_local.hip", *_PANEL_H], "panel_local_xcd_f32.hip": ["panel_local.hip", *_PANEL_H],
              "panel_single_f32.hip": ["panel_single.hip", *_PANEL_H], "panel_blocked_f32.hip": ["panel_blocked.hip", *_PANEL_H]}
# (a source that #includes another source, or a header only some sources see: kept per source so that touching one
#  kernel family does not rebuild the others)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-Wno-unused-result"] + (["-DRFLU_EXPERIMENTS"] if EXPERIMENTS else [])


def _mtime(p):
    return os.path.getmtime(p) if os.path.exists(p) else 0.0


def _digest(paths):
    """Content hash of the sources (mtimes do not survive the copy to the GPU box)."""
    import hashlib

    h = hashlib.sha1(" ".join(FLAGS).encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def sources_digest() -> str:
    """One hash over every kernel / driver source and header: identifies the build a measurement belongs to."""
    return _digest([os.path.join(CSRC, f) for f in SOURCES] + [os.path.join(CSRC, h) for h in ALL_HEADERS])


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def build_librflu(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs, stamps = [], {}
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        stamp = op + ".sha1"
        want = _digest([sp, *hdrs, *[os.path.join(CSRC, d) for d in EXTRA_DEPS.get(src, [])]])
        have = open(stamp).read().strip() if os.path.exists(stamp) else ""
        if force or not os.path.exists(op) or have != want:
            jobs.append((sp, op))
            stamps[op] = (stamp, want)

    def compile_one(job):
        sp, op = job
        cmd = [_hipcc(), *FLAGS, "-c", sp, "-o", op]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {sp}:\n{r.stdout}\n{r.stderr}")
        stamp, want = stamps[op]
        with open(stamp, "w") as f:
            f.write(want)
        return op

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(compile_one, jobs))
    objs = [os.path.join(OBJ, os.path.splitext(s)[0] + ".o") for s in SOURCES]
    if force or jobs or not os.path.exists(LIB):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    import sys

    print(build_librflu(force="--force" in sys.argv, verbose=True))
