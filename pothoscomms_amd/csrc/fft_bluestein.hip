// fft_bluestein.hip -- /comms/fft for the float sizes no other device plan takes (numBins whose factors fit neither one
// workgroup's LDS nor a four-step split, e.g. 2 x a prime beyond 10240): Bluestein's chirp-z form on the power-of-two plans.
//
// FFTFactory takes ANY numBins (fft/FFT.cpp:83-93) and kissfft factorises anything, falling back to an O(p^2) generic
// butterfly for a large prime p (fft/kissfft.hh:38-55,244-303); the device used to answer PCX_ERR_UNSUPPORTED there, so such
// a block failed to construct.  With n k = (n^2 + k^2 - (k - n)^2) / 2:
//     X[k] = w[k] * sum_n (x[n] w[n]) * conj(w)[k - n],      w[n] = exp(-j pi n^2 / N)
// i.e. a length-N chirp multiply, a linear convolution with the conjugate chirp -- evaluated as a circular one of size
// M = the power of two >= 2N - 1 through the existing forward / inverse plans -- and another chirp multiply.
// The chirp is tabulated on the host in double precision with n^2 reduced modulo 2N before the division (the phase stays
// exact for any N), so the float result stays well inside the 1e-5 bar; cost: two M-point transforms + three passes.
// The inverse transform (exp(+j ...)) is the forward one between two conjugations, folded into the first and last pass.
// complex_int16 stays unsupported beyond 32768 bins: kiss_fft's Q15 rounding sequence cannot be kept through a chirp.
#include "pcx_internal.hpp"

namespace pcx {

template <typename T>
struct CT {
    T x, y;
};
template <typename T>
__device__ __forceinline__ CT<T> cmul(CT<T> a, CT<T> b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }

// y[f][m] = (conj?) x[f][m] * w[m] for m < N, 0 for N <= m < M
template <typename T>
__global__ __launch_bounds__(256) void bluestein_pre_kernel(const CT<T> *__restrict__ x, CT<T> *__restrict__ y, const CT<T> *__restrict__ w, size_t N,
                                                            size_t M, size_t nframes, int inverse)
{
    const size_t total = nframes * M, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const size_t f = i / M, m = i - f * M;
        CT<T> v = {0, 0};
        if (m < N) {
            v = x[f * N + m];
            if (inverse) v.y = -v.y;
            v = cmul(v, w[m]);
        }
        y[i] = v;
    }
}
// Y[f][m] *= B[m]
template <typename T>
__global__ __launch_bounds__(256) void bluestein_mul_kernel(CT<T> *__restrict__ Y, const CT<T> *__restrict__ B, size_t M, size_t nframes)
{
    const size_t total = nframes * M, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) Y[i] = cmul(Y[i], B[i % M]);
}
// X[f][k] = (conj?) z[f][k] * w[k] / M for k < N
template <typename T>
__global__ __launch_bounds__(256) void bluestein_post_kernel(const CT<T> *__restrict__ z, CT<T> *__restrict__ X, const CT<T> *__restrict__ w, size_t N,
                                                             size_t M, size_t nframes, int inverse, T inv_m)
{
    const size_t total = nframes * N, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const size_t f = i / N, k = i - f * N;
        CT<T> v = cmul(z[f * M + k], w[k]);
        v.x *= inv_m; v.y *= inv_m;
        if (inverse) v.y = -v.y;
        X[i] = v;
    }
}

template <typename T>
static int pre_t(const void *x, void *y, const void *w, size_t N, size_t M, size_t nframes, bool inverse, hipStream_t st)
{
    hipLaunchKernelGGL(bluestein_pre_kernel<T>, dim3(stream_grid(nframes * M, 256)), dim3(256), 0, st, (const CT<T> *)x, (CT<T> *)y, (const CT<T> *)w, N, M,
                       nframes, inverse ? 1 : 0);
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}
template <typename T>
static int mul_t(void *Y, const void *B, size_t M, size_t nframes, hipStream_t st)
{
    hipLaunchKernelGGL(bluestein_mul_kernel<T>, dim3(stream_grid(nframes * M, 256)), dim3(256), 0, st, (CT<T> *)Y, (const CT<T> *)B, M, nframes);
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}
template <typename T>
static int post_t(const void *z, void *X, const void *w, size_t N, size_t M, size_t nframes, bool inverse, hipStream_t st)
{
    hipLaunchKernelGGL(bluestein_post_kernel<T>, dim3(stream_grid(nframes * N, 256)), dim3(256), 0, st, (const CT<T> *)z, (CT<T> *)X, (const CT<T> *)w, N, M,
                       nframes, inverse ? 1 : 0, (T)(1.0 / (double)M));
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}

int launch_bluestein_pre(int scalar, const void *x, void *y, const void *w, size_t N, size_t M, size_t nframes, bool inverse, hipStream_t st)
{
    return scalar == PCX_F32 ? pre_t<float>(x, y, w, N, M, nframes, inverse, st) : pre_t<double>(x, y, w, N, M, nframes, inverse, st);
}
int launch_bluestein_mul(int scalar, void *Y, const void *B, size_t M, size_t nframes, hipStream_t st)
{
    return scalar == PCX_F32 ? mul_t<float>(Y, B, M, nframes, st) : mul_t<double>(Y, B, M, nframes, st);
}
int launch_bluestein_post(int scalar, const void *z, void *X, const void *w, size_t N, size_t M, size_t nframes, bool inverse, hipStream_t st)
{
    return scalar == PCX_F32 ? post_t<float>(z, X, w, N, M, nframes, inverse, st) : post_t<double>(z, X, w, N, M, nframes, inverse, st);
}

}  // namespace pcx
