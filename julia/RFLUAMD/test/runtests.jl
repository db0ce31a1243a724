# Mirrors RecursiveFactorization's test/runtests.jl:14-68 (info equality with LAPACK, residual bound, solve check) through
# RFLUAMD; sizes from GPU_MIN_N up take the MI355X path (asserted through rflu_last_path).  Needs Julia + a gfx950 device.
using Test, LinearAlgebra, Random
using RFLUAMD
Random.seed!(12)
const baselu = LinearAlgebra.lu
function testlu(A, MF, BF, p)
    @test MF.info == BF.info
    iszero(MF.info) || return
    E = 20size(A, 1) * eps(real(one(float(first(A)))))
    @test norm(MF.L * MF.U - A[MF.p, :], Inf) < (p ? E : 10sqrt(E))
end
@testset "RFLUAMD lu / lu!" begin
    RFLUAMD.GPU_MIN_N[] = 64
    for _p in (true, false), T in (Float64, Float32), s in (64, 130, 300, 1000)
        for m in (s, s + 2)
            A = rand(T, s, m)
            _p || (A = A + T(10) * Matrix{T}(I, s, m))          # the reference's NoPivot inputs are diagonally shifted
            MF = RFLUAMD.lu(A, Val(_p))
            RFLUAMD.available() && @test RFLUAMD.last_path() in (1, 3, 4)
            testlu(A, MF, baselu(A, _p ? RowMaximum() : NoPivot()), _p)
            At = permutedims(A)
            testlu(At, parent(RFLUAMD.lu(At', Val(_p))), baselu(At, _p ? RowMaximum() : NoPivot()), _p)
        end
    end
    A = rand(300, 300); A[:, 17] .= 0
    @test RFLUAMD.lu(A; check = false).info == baselu(A; check = false).info
    ipiv = fill(typemax(Int64) - 7, 300)
    F = RFLUAMD.lu!(rand(300, 300) + 10I, ipiv, Val(false), Val(false))
    @test ipiv == 1:300
end
