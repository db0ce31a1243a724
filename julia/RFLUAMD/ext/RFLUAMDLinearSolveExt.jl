# LinearSolve.jl extension: RFLUAMDFactorization as a drop-in for RFLUFactorization (same cache protocol).
# LinearSolve's own `solve!(cache, ::RFLUFactorization{P,T})` does, in order:
#     fact, ipiv = cacheval; if cache.isfresh: resize ipiv; fact = RecursiveFactorization.lu!(A, ipiv, Val(P), Val(T), check = false);
#     cache.cacheval = (fact, ipiv); !issuccess(fact) -> ReturnCode.Failure; y = ldiv!(cache.u, fact, cache.b)
# (RecursiveFactorization README.md:36-37 names it; the call shape is src/lu.jl:97-130 with check = false).
module RFLUAMDLinearSolveExt

using LinearAlgebra
using LinearSolve
using RFLUAMD
using RFLUAMD: RFLUAMDFactorization

# the algorithm type takes part in LinearSolve's factorization machinery
LinearSolve.needs_concrete_A(::RFLUAMDFactorization) = true

function LinearSolve.init_cacheval(alg::RFLUAMDFactorization{P}, A, b, u, Pl, Pr, maxiters::Int, abstol, reltol,
                                   verbose, assumptions::LinearSolve.OperatorAssumptions) where {P}
    A isa AbstractMatrix || return nothing
    ipiv = Vector{LinearAlgebra.BlasInt}(undef, min(size(A)...))
    # a well-typed placeholder factorization, like LinearSolve's ArrayInterface.lu_instance
    fact = LinearAlgebra.LU(similar(A, 0, 0), similar(ipiv, 0), zero(LinearAlgebra.BlasInt))
    return (fact, ipiv)
end

function LinearSolve.solve!(cache::LinearSolve.LinearCache, alg::RFLUAMDFactorization{P}; kwargs...) where {P}
    A = convert(AbstractMatrix, cache.A)
    fact, ipiv = LinearSolve.@get_cacheval(cache, :RFLUAMDFactorization)
    if cache.isfresh
        if length(ipiv) != min(size(A)...)
            ipiv = Vector{LinearAlgebra.BlasInt}(undef, min(size(A)...))
        end
        fact = RFLUAMD.lu!(A, ipiv, Val(P), Val(false); check = false, blocksize = alg.blocksize)
        cache.cacheval = (fact, ipiv)
        if !LinearAlgebra.issuccess(fact)
            return SciMLBase.build_linear_solution(alg, cache.u, nothing, cache; retcode = ReturnCode.Failure)
        end
        cache.isfresh = false
    end
    y = RFLUAMD.ldiv!(LinearSolve.@get_cacheval(cache, :RFLUAMDFactorization)[1], copyto!(cache.u, cache.b))
    return SciMLBase.build_linear_solution(alg, y, nothing, cache; retcode = ReturnCode.Success)
end

end # module
