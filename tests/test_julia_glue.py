"""The Julia host glue (julia/RFLUAMD) cannot be executed here (no Julia in the image): check it mechanically instead.
Every `ccall((:symbol, librflu), Ret, (ArgTypes...), args...)` in the package is parsed and compared with the prototype of
that symbol in include/rflu.h: the symbol must exist, the arity must match (type tuple AND actual arguments), and every
Julia argument type must be the C type's FFI image."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL_DIR = os.path.join(ROOT, "julia", "RFLUAMD")

# Julia FFI type -> set of acceptable C parameter types (normalised: no names, no const, single spaces)
JL2C = {
    "Cint": {"int"},
    "Int64": {"int64_t"},
    "UInt64": {"uint64_t"},
    "Cdouble": {"double"},
    "Cstring": {"char*"},
    "Ptr{Cvoid}": {"rflu_handle_t", "void*", "rflu_mgpu_t"},
    "Ref{Ptr{Cvoid}}": {"rflu_handle_t*", "void**", "rflu_mgpu_t*"},
    "Ptr{Float64}": {"double*"},
    "Ptr{Float32}": {"float*"},
    "Ptr{Int64}": {"int64_t*"},
    "Ref{Int64}": {"int64_t*"},
    "Ptr{Cint}": {"int*"},
}


def c_prototypes():
    src = open(os.path.join(ROOT, "include", "rflu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(const char\*|int)\s+(rflu_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, params = m.group(1), m.group(2), m.group(3).strip()
        plist = []
        if params and params != "void":
            for p in params.split(","):
                p = re.sub(r"\bconst\b", "", p).strip()
                p = re.sub(r"\s*\*\s*", "* ", p)          # "double* A" / "double *A" -> "double* A"
                toks = p.split()
                ty = " ".join(toks[:-1]) if len(toks) > 1 else toks[0]
                plist.append(ty.replace(" ", ""))
        protos[name] = (ret.replace("const ", "").replace(" ", ""), plist)
    return protos


def split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def julia_ccalls():
    calls = []
    for dirpath, _, files in os.walk(JL_DIR):
        for fn in files:
            if not fn.endswith(".jl"):
                continue
            text = open(os.path.join(dirpath, fn)).read()
            text = re.sub(r"#[^\n]*", "", text)
            for m in re.finditer(r"ccall\(", text):
                i = m.end()
                depth, j = 1, i
                while depth:
                    depth += {"(": 1, ")": -1}.get(text[j], 0)
                    j += 1
                parts = split_top(text[i:j - 1])
                sym = re.match(r"\(\s*:(\w+)\s*,\s*librflu\s*\)", parts[0])
                assert sym, f"{fn}: ccall target {parts[0]!r} is not (:symbol, librflu)"
                argt = parts[2].strip()
                assert argt.startswith("(") and argt.endswith(")"), (fn, argt)
                types = split_top(argt[1:-1])
                calls.append((fn, sym.group(1), parts[1].strip(), types, parts[3:]))
    return calls


def test_glue_files_exist():
    for rel in ("Project.toml", "src/RFLUAMD.jl", "ext/RFLUAMDLinearSolveExt.jl", "test/runtests.jl"):
        assert os.path.exists(os.path.join(JL_DIR, rel)), rel


def test_every_ccall_matches_the_header():
    protos = c_prototypes()
    calls = julia_ccalls()
    assert len(calls) >= 9
    for fn, sym, ret, types, args in calls:
        assert sym in protos, f"{fn}: {sym} is not declared in include/rflu.h"
        cret, cparams = protos[sym]
        assert JL2C[ret] & {cret}, f"{fn}: {sym} returns {cret}, the ccall says {ret}"
        assert len(types) == len(cparams), f"{fn}: {sym} takes {len(cparams)} arguments, the ccall type tuple has {len(types)}"
        assert len(args) == len(types), f"{fn}: {sym}: {len(types)} types but {len(args)} actual arguments"
        for k, (jt, ct) in enumerate(zip(types, cparams)):
            assert jt in JL2C, f"{fn}: {sym}: unknown Julia FFI type {jt}"
            assert ct in JL2C[jt], f"{fn}: {sym}: argument {k + 1} is `{ct}` in rflu.h but `{jt}` in the ccall"


def test_boundary_symbols_are_bound():
    bound = {c[1] for c in julia_ccalls()}
    for need in ("rflu_create", "rflu_destroy", "rflu_last_error", "rflu_last_path", "rflu_getrf_f64", "rflu_getrf_f32",
                 "rflu_getrf_f64_dev", "rflu_getrf_f32_dev", "rflu_getrs_f64", "rflu_getrs_f32"):
        assert need in bound, need


def test_reference_signature_and_linear_solve_protocol():
    src = open(os.path.join(JL_DIR, "src", "RFLUAMD.jl")).read()
    # src/lu.jl:97-102: lu!(A, ipiv, pivot = Val(true), thread = Val(false); check, blocksize, threshold)
    assert re.search(r"function lu!\(A::AbstractMatrix\{T\}, ipiv::AbstractVector\{<:Integer\}, pivot = Val\(true\), "
                     r"thread = Val\(false\);\s*check::Union\{Bool, Val\{true\}, Val\{false\}\} = Val\(true\), "
                     r"blocksize::Integer = 0,\s*threshold::Integer = 0\)", src)
    assert "checknonsingular(info)" in src and "LU(A, ipiv, info)" in src and "GC.@preserve A ipiv" in src
    assert "copyto!(ipiv, 1:mnmin)" in src            # NoPivot identity fill, src/lu.jl:111-113
    assert "NOPIVOT_NEGATIVE_INFO" in src and "info = -info" in src
    ext = open(os.path.join(JL_DIR, "ext", "RFLUAMDLinearSolveExt.jl")).read()
    for needle in ("cache.isfresh", "check = false", "issuccess(fact)", "ReturnCode.Failure", "cache.cacheval = (fact, ipiv)",
                   "RFLUAMD.ldiv!"):
        assert needle in ext, needle
