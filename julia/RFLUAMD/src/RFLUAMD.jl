"""
    RFLUAMD

Julia host glue for `librflu.so`, the MI355X (gfx950) implementation of RecursiveFactorization.jl's `lu!` hot path
(`lu!` -> `recurse!` -> `reckernel!`, RecursiveFactorization `src/lu.jl:97-338`).  The C ABI is `include/rflu.h`; every
`ccall` below is checked mechanically against that header by `tests/test_julia_glue.py` (arity and argument types), because
Julia is not available in the image this repository is built in.

Surface (the reference's own, `src/lu.jl:19-21, 67-83, 97-130`):

    RFLUAMD.lu(A, pivot = Val(true), thread = Val(false); check, blocksize, threshold)
    RFLUAMD.lu!(A, pivot = Val(true), thread = Val(false); check, blocksize, threshold)
    RFLUAMD.lu!(A, ipiv, pivot = Val(true), thread = Val(false); check, blocksize, threshold)   # what RFLUFactorization calls

returning a genuine `LinearAlgebra.LU(A, ipiv, info)` that aliases the caller's arrays.  `Float64` / `Float32` strided
column-major matrices with at least `GPU_MIN_N[]` columns go to the GPU; everything else (other element types, non-strided
storage, small sizes, no device) goes to `RecursiveFactorization.lu!` when that package is loaded, else to
`LinearAlgebra.lu!`/`generic_lufact!` -- the same fall-back rules the reference applies (`src/lu.jl:74-77, 92-93, 114-126`).
"""
module RFLUAMD

using LinearAlgebra
using LinearAlgebra: BlasInt, LU, RowMaximum, NoPivot, checknonsingular
using Libdl

export RFLUAMDFactorization

const librflu = get(ENV, "RFLU_LIB", "librflu.so")
const HANDLE = Ref{Ptr{Cvoid}}(C_NULL)
const HANDLE_LOCK = ReentrantLock()
"below this many columns the CPU recursion wins (PCIe staging + launch latency); `ENV[\"RFLU_MIN_N\"]` overrides"
const GPU_MIN_N = Ref{Int}(parse(Int, get(ENV, "RFLU_MIN_N", "1024")))
"Julia >= 1.11 reports NoPivot failures with a negative info (RecursiveFactorization src/lu.jl:25)"
const NOPIVOT_NEGATIVE_INFO = VERSION >= v"1.11.0-DEV"   # as src/lu.jl:25

# rflu_status (include/rflu.h)
const RFLU_OK = Cint(0)

last_error() = unsafe_string(ccall((:rflu_last_error, librflu), Cstring, ()))

function handle()
    lock(HANDLE_LOCK) do
        if HANDLE[] == C_NULL
            st = ccall((:rflu_create, librflu), Cint, (Ref{Ptr{Cvoid}}, Cint), HANDLE, Cint(0))
            st == RFLU_OK || error("rflu_create failed: ", last_error())
            atexit() do
                ccall((:rflu_destroy, librflu), Cint, (Ptr{Cvoid},), HANDLE[])
                HANDLE[] = C_NULL
            end
        end
        HANDLE[]
    end
end

"`true` when librflu.so loads and finds a gfx950 device (no CPU fallback lives inside the library)"
function available()
    Libdl.dlopen(librflu; throw_error = false) === nothing && return false
    try
        return handle() != C_NULL
    catch
        return false
    end
end

"which implementation served the last factorization: 1 recursive, 2 blocked (profiling), 3 lookahead (rflu_path)"
last_path() = Int(ccall((:rflu_last_path, librflu), Cint, (Ptr{Cvoid},), handle()))

normalize_pivot(t::Val{T}) where {T} = t                     # RecursiveFactorization src/lu.jl:10-17
normalize_pivot(::RowMaximum) = Val(true)
normalize_pivot(::NoPivot) = Val(false)

"zero-storage identity pivots for NoPivot (RecursiveFactorization src/lu.jl:27-40); the C ABI takes NULL for it"
struct NotIPIV <: AbstractVector{BlasInt}
    len::Int
end
Base.size(A::NotIPIV) = (A.len,)
Base.getindex(::NotIPIV, i::Int) = i
Base.view(::NotIPIV, r::AbstractUnitRange) = NotIPIV(length(r))
init_pivot(::Val{false}, minmn) = NotIPIV(minmn)
init_pivot(::Val{true}, minmn) = Vector{BlasInt}(undef, minmn)

# ---- the C ABI, one method per element type (literal ccall tuples: tests/test_julia_glue.py parses them) ------------------
function getrf!(A::StridedMatrix{Float64}, ipiv::Ptr{Int64}, pivot::Bool, blocksize::Integer)
    m, n = size(A)
    info = Ref{Int64}(0)
    st = ccall((:rflu_getrf_f64, librflu), Cint,
               (Ptr{Cvoid}, Int64, Int64, Ptr{Float64}, Int64, Ptr{Int64}, Cint, Int64, Ref{Int64}),
               handle(), m, n, A, stride(A, 2), ipiv, Cint(pivot), blocksize, info)
    st == RFLU_OK || error("librflu: ", last_error())
    return BlasInt(info[])
end

function getrf!(A::StridedMatrix{Float32}, ipiv::Ptr{Int64}, pivot::Bool, blocksize::Integer)
    m, n = size(A)
    info = Ref{Int64}(0)
    st = ccall((:rflu_getrf_f32, librflu), Cint,
               (Ptr{Cvoid}, Int64, Int64, Ptr{Float32}, Int64, Ptr{Int64}, Cint, Int64, Ref{Int64}),
               handle(), m, n, A, stride(A, 2), ipiv, Cint(pivot), blocksize, info)
    st == RFLU_OK || error("librflu: ", last_error())
    return BlasInt(info[])
end

"device-resident variant for a matrix that already lives in HBM (e.g. the buffer of an AMDGPU.jl ROCArray)"
function getrf_dev!(A::Ptr{Float64}, m::Integer, n::Integer, lda::Integer, ipiv::Ptr{Int64}, pivot::Bool, blocksize::Integer)
    info = Ref{Int64}(0)
    st = ccall((:rflu_getrf_f64_dev, librflu), Cint,
               (Ptr{Cvoid}, Int64, Int64, Ptr{Float64}, Int64, Ptr{Int64}, Cint, Int64, Ref{Int64}),
               handle(), m, n, A, lda, ipiv, Cint(pivot), blocksize, info)
    st == RFLU_OK || error("librflu: ", last_error())
    return BlasInt(info[])
end

function getrf_dev!(A::Ptr{Float32}, m::Integer, n::Integer, lda::Integer, ipiv::Ptr{Int64}, pivot::Bool, blocksize::Integer)
    info = Ref{Int64}(0)
    st = ccall((:rflu_getrf_f32_dev, librflu), Cint,
               (Ptr{Cvoid}, Int64, Int64, Ptr{Float32}, Int64, Ptr{Int64}, Cint, Int64, Ref{Int64}),
               handle(), m, n, A, lda, ipiv, Cint(pivot), blocksize, info)
    st == RFLU_OK || error("librflu: ", last_error())
    return BlasInt(info[])
end

"`ldiv!(F, B)` on the GPU: B <- U^-1 L^-1 P B (stdlib `ldiv!(::LU, B)`; the package's own for NotIPIV, src/lu.jl:60-64)"
function getrs!(F::StridedMatrix{Float64}, ipiv::Ptr{Int64}, B::StridedVecOrMat{Float64})
    n = size(F, 1)
    st = ccall((:rflu_getrs_f64, librflu), Cint,
               (Ptr{Cvoid}, Int64, Int64, Ptr{Float64}, Int64, Ptr{Int64}, Ptr{Float64}, Int64),
               handle(), n, size(B, 2), F, stride(F, 2), ipiv, B, B isa AbstractVector ? n : stride(B, 2))
    st == RFLU_OK || error("librflu: ", last_error())
    return B
end

function getrs!(F::StridedMatrix{Float32}, ipiv::Ptr{Int64}, B::StridedVecOrMat{Float32})
    n = size(F, 1)
    st = ccall((:rflu_getrs_f32, librflu), Cint,
               (Ptr{Cvoid}, Int64, Int64, Ptr{Float32}, Int64, Ptr{Int64}, Ptr{Float32}, Int64),
               handle(), n, size(B, 2), F, stride(F, 2), ipiv, B, B isa AbstractVector ? n : stride(B, 2))
    st == RFLU_OK || error("librflu: ", last_error())
    return B
end

# ---- dispatch: who serves a call (RecursiveFactorization src/lu.jl:92-93, 114-126) -------------------------------------------
const GPUEltype = Union{Float32, Float64}
gpu_ok(A::StridedMatrix{<:GPUEltype}, ipiv) =
    stride(A, 1) == 1 && min(size(A)...) >= GPU_MIN_N[] && (ipiv isa Vector{Int64} || ipiv isa NotIPIV) && available()
gpu_ok(A, ipiv) = false

const CPU_FALLBACK = Ref{Any}(nothing)   # set to RecursiveFactorization.lu! by the user / an extension when that package is loaded
function cpu_lu!(A, ipiv, pivot, thread; check, kwargs...)
    f = CPU_FALLBACK[]
    f === nothing || return f(A, ipiv, pivot, thread; check = check, kwargs...)
    F = LinearAlgebra.lu!(A, pivot === Val(true) ? RowMaximum() : NoPivot(); check = check)
    ipiv isa AbstractVector && !(ipiv isa NotIPIV) && copyto!(ipiv, F.ipiv)
    return LU(F.factors, ipiv isa NotIPIV ? F.ipiv : ipiv, F.info)
end

"""
    lu!(A, ipiv, pivot = Val(true), thread = Val(false); check = Val(true), blocksize = 0, threshold = 0)

The method LinearSolve's `RFLUFactorization` calls (RecursiveFactorization `src/lu.jl:97-130`), served by the MI355X.
`blocksize`: 0 = library default, negative = pure Toledo recursion, 64/128/256... = outer block-column width (see rflu.h);
`threshold` and `thread` are accepted for signature parity (the GPU path has neither knob).
"""
function lu!(A::AbstractMatrix{T}, ipiv::AbstractVector{<:Integer}, pivot = Val(true), thread = Val(false);
             check::Union{Bool, Val{true}, Val{false}} = Val(true), blocksize::Integer = 0,
             threshold::Integer = 0) where {T}
    pivot = normalize_pivot(pivot)
    check isa Bool && (check = Val(check))
    gpu_ok(A, ipiv) || return cpu_lu!(A, ipiv, pivot, thread; check = check === Val(true))
    mnmin = min(size(A)...)
    if pivot === Val(false) && !(ipiv isa NotIPIV)
        copyto!(ipiv, 1:mnmin)                                  # src/lu.jl:111-113 (the library fills it as well)
    end
    p = ipiv isa NotIPIV ? Ptr{Int64}(C_NULL) : pointer(ipiv)
    info = GC.@preserve A ipiv getrf!(A, p, pivot === Val(true), blocksize)
    (pivot === Val(false) && NOPIVOT_NEGATIVE_INFO) && (info = -info)   # src/lu.jl:249-254, 323-326
    check === Val(true) && checknonsingular(info)                       # src/lu.jl:128
    return LU(A, ipiv, info)                                            # src/lu.jl:129
end

function lu!(A::AbstractMatrix, pivot = Val(true), thread = Val(false); check = Val(true), kwargs...)   # src/lu.jl:67-83
    npivot = normalize_pivot(pivot)
    return lu!(A, init_pivot(npivot, min(size(A)...)), npivot, thread; check = check, kwargs...)
end

lu(A::AbstractMatrix, pivot = Val(true), thread = Val(false); kwargs...) = lu!(copy(A), pivot, thread; kwargs...)   # :19-21

for (f, T) in [(:adjoint, :Adjoint), (:transpose, :Transpose)], lufn in (:lu, :lu!)      # src/lu.jl:85-87
    @eval $lufn(A::$T, args...; kwargs...) = $f($lufn(parent(A), args...; kwargs...))
end

"solve with the factors on the GPU when they are large enough, else stdlib `ldiv!`"
function ldiv!(F::LU{T, <:StridedMatrix{T}}, B::StridedVecOrMat{T}) where {T <: GPUEltype}
    # the same layout conditions as `gpu_ok` for lu!: getrs! passes stride(F, 2) / stride(B, 2) as leading dimensions, i.e. it
    # assumes unit row stride and a square factorization; anything else (a strided view, an LU from a non-unit-stride CPU
    # fallback, a B with the wrong number of rows) stays with the stdlib
    lay_ok = stride(F.factors, 1) == 1 && stride(B, 1) == 1 && size(F.factors, 1) == size(F.factors, 2) == size(B, 1)
    if lay_ok && size(F.factors, 1) >= GPU_MIN_N[] && available() && (F.ipiv isa Vector{Int64} || F.ipiv isa NotIPIV)
        p = F.ipiv isa NotIPIV ? Ptr{Int64}(C_NULL) : pointer(F.ipiv)
        GC.@preserve F B getrs!(F.factors, p, B)
        return B
    end
    return LinearAlgebra.ldiv!(F, B)
end

"""
    RFLUAMDFactorization(; pivot = Val(true), blocksize = 0)

LinearSolve.jl algorithm with the cache protocol of `RFLUFactorization{P,T}`: `cacheval = (fact, ipiv)`, `lu!(A, ipiv,
Val(P), Val(false); check = false)` when the cache is fresh, `issuccess(fact)` -> `ReturnCode.Failure`, then `ldiv!`.
The methods live in the package extension `ext/RFLUAMDLinearSolveExt.jl` (loaded with LinearSolve).
"""
struct RFLUAMDFactorization{P}
    blocksize::Int
    RFLUAMDFactorization(::Val{P}, blocksize::Integer = 0) where {P} = new{P}(Int(blocksize))
end
RFLUAMDFactorization(; pivot = Val(true), blocksize::Integer = 0) = RFLUAMDFactorization(normalize_pivot(pivot), blocksize)

end # module
