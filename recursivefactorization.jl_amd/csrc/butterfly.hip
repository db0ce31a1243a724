// butterfly.hip -- randomized butterfly pre-transform  A <- U' A V  and the matching vector transforms.
//
// Replaces 🦋mul! / 🦋mul_level! (/root/reference/src/butterflylu.jl:59-113) and the dense  mul!(tmp, U', b) / mul!(b, V, tmp)
// of 🦋solve! (:45-55), whose U, V are the products of two butterfly levels (materializeUV, :149-178):
//     B(a, b) = [ D_a  D_b ; D_a  -D_b ]     U = blockdiag(B(U1), B(U2)) * B(Uf),    V likewise,
// with the 4n random diagonal entries uv laid out as the reference lays them out (:63-78):
//     uv[0 : h) = U1   uv[h : n) = V1   uv[n : n+h) = U2   uv[n+h : 2n) = V2   uv[2n : 3n) = Uf   uv[3n : 4n) = Vf     (h = n/2)
// After the transform a NoPivot LU is safe (that is the point: no pivot search -> no latency chain on the GPU), and
//     x = V * (U' A V)^-1 * U' b.
//
// The reference makes two sweeps over A (level 2 on the four quadrants, then level 1).  Here ONE sweep does both: a thread
// owns the 4 x 4 set of entries {m, m+q, m+2q, m+3q} x {c, c+q, c+2q, c+3q} (q = n/4) that the two levels mix among
// themselves, applies the reference's expressions in the reference's order (bit-identical results) and writes them back:
// 16*sizeof(T) bytes of HBM traffic per 16 entries -- the algorithmic minimum of an in-place transform.
// Roofline: HBM; algorithmic bytes 2*sizeof(T)*n^2.
#include "rflu_internal.hpp"

// The fused sweep must round exactly like the reference's two sweeps: a product of level 2 feeds an addition of level 1, and
// hipcc's default contraction would fuse the two into one fma (one rounding instead of two).
#pragma STDC FP_CONTRACT OFF

namespace rflu {

// one 2 x 2 butterfly of 🦋mul_level! (src/butterflylu.jl:66-88), same operations in the same order
template <typename T>
__device__ __forceinline__ void bfly(T& a11, T& a21, T& a12, T& a22, T u1, T u2, T v1, T v2)
{
    const T t1 = a11 + a12, t2 = a21 + a22, t3 = a11 - a12, t4 = a21 - a22;
    const T c11 = t1 + t2, c21 = t1 - t2, c12 = t3 + t4, c22 = t3 - t4;
    a11 = u1 * c11 * v1;
    a21 = u2 * c21 * v1;
    a12 = u1 * c12 * v2;
    a22 = u2 * c22 * v2;
}

// A: column-major n x n (lda), n % 4 == 0.  Thread (m, c): rows m + i*q, columns c + j*q.
template <typename T>
__global__ void __launch_bounds__(256) butterfly_mul_kernel(T* __restrict__ A, int64_t lda, int n, const T* __restrict__ uv)
{
    const int q = n >> 2, h = n >> 1;
    const int m = blockIdx.x * 256 + threadIdx.x;   // fast index = row (contiguous in a column-major matrix)
    const int c = blockIdx.y;
    if (m >= q) return;
    T x[4][4];   // x[i][j] = A[m + i*q, c + j*q]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i][j] = A[(int64_t)(c + j * q) * lda + m + i * q];
    // level 2: each quadrant is its own (n/2 x n/2) butterfly with halves of size q; U1/U2 act on the top/bottom row half,
    // V1/V2 on the left/right column half (src/butterflylu.jl:95-105)
    const T* U1 = uv;
    const T* V1 = uv + h;
    const T* U2 = uv + n;
    const T* V2 = uv + n + h;
    bfly<T>(x[0][0], x[1][0], x[0][1], x[1][1], U1[m], U1[m + q], V1[c], V1[c + q]);   // rows top,    columns left
    bfly<T>(x[2][0], x[3][0], x[2][1], x[3][1], U2[m], U2[m + q], V1[c], V1[c + q]);   // rows bottom, columns left
    bfly<T>(x[0][2], x[1][2], x[0][3], x[1][3], U1[m], U1[m + q], V2[c], V2[c + q]);   // rows top,    columns right
    bfly<T>(x[2][2], x[3][2], x[2][3], x[3][3], U2[m], U2[m + q], V2[c], V2[c + q]);   // rows bottom, columns right
    // level 1 on the whole matrix: halves of size h; entry (m + i*q) pairs with (m + i*q + h) = index i + 2
    const T* Uf = uv + 2 * (int64_t)n;
    const T* Vf = uv + 3 * (int64_t)n;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            bfly<T>(x[i][j], x[i + 2][j], x[i][j + 2], x[i + 2][j + 2], Uf[m + i * q], Uf[m + i * q + h], Vf[c + j * q],
                    Vf[c + j * q + h]);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) A[(int64_t)(c + j * q) * lda + m + i * q] = x[i][j];
}

// x <- U' x (mode 0) or x <- V x (mode 1) for nrhs vectors (column-major n x nrhs, ldx); thread = 4 entries of one vector.
//   B(a,b)  x: y_i = a_i x_i + b_i x_{i+s},  y_{i+s} = a_i x_i - b_i x_{i+s}
//   B(a,b)' x: y_i = a_i (x_i + x_{i+s}),    y_{i+s} = b_i (x_i - x_{i+s})
// U' = B(Uf)' * blockdiag(B(U1)', B(U2)'),  V = blockdiag(B(V1), B(V2)) * B(Vf)   (materializeUV, src/butterflylu.jl:149-178)
template <typename T>
__global__ void __launch_bounds__(256) butterfly_vec_kernel(T* __restrict__ X, int64_t ldx, int n, const T* __restrict__ uv, int mode)
{
    const int q = n >> 2, h = n >> 1;
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= q) return;
    T* x = X + (int64_t)blockIdx.y * ldx;
    T x0 = x[m], x1 = x[m + q], x2 = x[m + h], x3 = x[m + h + q];
    if (mode == 0) {
        const T* U1 = uv;
        const T* U2 = uv + n;
        const T* Uf = uv + 2 * (int64_t)n;
        // level 2 transposed on each half (stride q), then level 1 transposed (stride h)
        T y0 = U1[m] * (x0 + x1), y1 = U1[m + q] * (x0 - x1);
        T y2 = U2[m] * (x2 + x3), y3 = U2[m + q] * (x2 - x3);
        x0 = Uf[m] * (y0 + y2);
        x2 = Uf[m + h] * (y0 - y2);
        x1 = Uf[m + q] * (y1 + y3);
        x3 = Uf[m + q + h] * (y1 - y3);
    } else {
        const T* V1 = uv + h;
        const T* V2 = uv + n + h;
        const T* Vf = uv + 3 * (int64_t)n;
        // level 1 forward (stride h), then level 2 forward on each half (stride q)
        T y0 = Vf[m] * x0 + Vf[m + h] * x2, y2 = Vf[m] * x0 - Vf[m + h] * x2;
        T y1 = Vf[m + q] * x1 + Vf[m + q + h] * x3, y3 = Vf[m + q] * x1 - Vf[m + q + h] * x3;
        x0 = V1[m] * y0 + V1[m + q] * y1;
        x1 = V1[m] * y0 - V1[m + q] * y1;
        x2 = V2[m] * y2 + V2[m + q] * y3;
        x3 = V2[m] * y2 - V2[m + q] * y3;
    }
    x[m] = x0;
    x[m + q] = x1;
    x[m + h] = x2;
    x[m + h + q] = x3;
}

template <typename T>
int launch_butterfly_mul(Handle* h, int64_t n, T* A, int64_t lda, const T* uv)
{
    if (n <= 0) return RFLU_OK;
    if (n % 4 != 0 || lda < n || n > (int64_t)65535 * 4) {
        set_error("butterfly: n = %lld must be a positive multiple of 4 (pad the system first) and lda >= n", (long long)n);
        return RFLU_ERR_ARG;
    }
    const int q = (int)(n / 4);
    ProfScope ps(h, RFLU_K_MISC, 0.0, 2.0 * sizeof(T) * (double)n * (double)n);
    hipLaunchKernelGGL(butterfly_mul_kernel<T>, dim3((unsigned)((q + 255) / 256), (unsigned)q), dim3(256), 0, h->stream, A, lda,
                       (int)n, uv);
    RFLU_HIP(hipGetLastError());
    return RFLU_OK;
}

template <typename T>
int launch_butterfly_vec(Handle* h, int64_t n, int64_t nrhs, T* X, int64_t ldx, const T* uv, int mode)
{
    if (n <= 0 || nrhs <= 0) return RFLU_OK;
    if (n % 4 != 0 || ldx < n || nrhs > 65535 || (mode != 0 && mode != 1)) {
        set_error("butterfly vector transform: bad arguments");
        return RFLU_ERR_ARG;
    }
    const int q = (int)(n / 4);
    hipLaunchKernelGGL(butterfly_vec_kernel<T>, dim3((unsigned)((q + 255) / 256), (unsigned)nrhs), dim3(256), 0, h->stream, X, ldx,
                       (int)n, uv, mode);
    RFLU_HIP(hipGetLastError());
    return RFLU_OK;
}

template int launch_butterfly_mul<double>(Handle*, int64_t, double*, int64_t, const double*);
template int launch_butterfly_mul<float>(Handle*, int64_t, float*, int64_t, const float*);
template int launch_butterfly_vec<double>(Handle*, int64_t, int64_t, double*, int64_t, const double*, int);
template int launch_butterfly_vec<float>(Handle*, int64_t, int64_t, float*, int64_t, const float*, int);

}  // namespace rflu
