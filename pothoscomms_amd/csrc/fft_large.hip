// fft_large.hip -- power-of-two transforms too long for one workgroup's LDS (complex_float32 beyond
// 16384 bins, complex_float64 beyond 4096): the four-step decomposition N = N1 * N2 around the
// single-workgroup kernels.  With n = N2*n1 + n2 and k = k1 + N1*k2,
//     X[k1 + N1 k2] = sum_n2 W_N2^(n2 k2) * [ W_N^(n2 k1) * sum_n1 x[N2 n1 + n2] W_N1^(n1 k1) ]
// so a frame viewed as an N1 x N2 matrix is  transposed, transformed along N1 (N2 short frames),
// transposed back with the twiddle W_N^(n2 k1) applied on the way, transformed along N2, and
// transposed once more into natural order.  Same unnormalised DFT as kissfft<T>::transform
// (fft/kissfft.hh:81-161), which accepts any size; parity bar 1e-5 of max|X|.  This file holds the
// batched tiled transpose (+ twiddle); pcx_api.hip strings the five launches together.
#include "pcx_internal.hpp"

namespace pcx {

namespace {

template <typename T>
struct C2 {
    T x, y;
};

// in: [batch][rows][cols], out: [batch][cols][rows]; MODE 0 plain, 1 multiply by exp(-j 2 pi r c / N),
// 2 by exp(+j 2 pi r c / N).  32 x 32 tiles through LDS (padded), 32 x 8 lanes.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void transpose_kernel(const C2<T> *__restrict__ in, C2<T> *__restrict__ out, unsigned rows, unsigned cols,
                                                        size_t batch, double inv_n)
{
    __shared__ C2<T> tile[32][33];
    const unsigned tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const unsigned c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (size_t b = blockIdx.z; b < batch; b += gridDim.z) {
        const C2<T> *src = in + b * (size_t)rows * cols;
        C2<T> *dst = out + b * (size_t)rows * cols;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 32; i += 8) {
            const unsigned r = r0 + ty + i, c = c0 + tx;
            if (r < rows && c < cols) {
                C2<T> v = src[(size_t)r * cols + c];
                if (MODE != 0) {
                    // r*c < rows*cols = N: the exponent needs no reduction; evaluated in double
                    double s, co;
                    sincospi(2.0 * (double)((unsigned long long)r * c) * inv_n, &s, &co);
                    if (MODE == 1) s = -s;
                    const T wr = (T)co, wi = (T)s;
                    const T re = v.x * wr - v.y * wi, im = v.x * wi + v.y * wr;
                    v.x = re; v.y = im;
                }
                tile[ty + i][tx] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 32; i += 8) {
            const unsigned c = c0 + ty + i, r = r0 + tx;   // output row = input column
            if (r < rows && c < cols) dst[(size_t)c * rows + r] = tile[tx][ty + i];
        }
    }
}

template <typename T>
int launch_transpose_t(const void *in, void *out, size_t rows, size_t cols, size_t batch, int mode, hipStream_t st)
{
    const dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32), (unsigned)(batch < 4096 ? batch : 4096));
    const double inv_n = 1.0 / ((double)rows * (double)cols);
    const C2<T> *pi = static_cast<const C2<T> *>(in);
    C2<T> *po = static_cast<C2<T> *>(out);
    if (mode == 0) hipLaunchKernelGGL((transpose_kernel<T, 0>), grid, dim3(256), 0, st, pi, po, (unsigned)rows, (unsigned)cols, batch, inv_n);
    else if (mode == 1) hipLaunchKernelGGL((transpose_kernel<T, 1>), grid, dim3(256), 0, st, pi, po, (unsigned)rows, (unsigned)cols, batch, inv_n);
    else hipLaunchKernelGGL((transpose_kernel<T, 2>), grid, dim3(256), 0, st, pi, po, (unsigned)rows, (unsigned)cols, batch, inv_n);
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}

}  // namespace

// complex matrices of `scalar` (PCX_F32 / PCX_F64): out[b][c][r] = in[b][r][c] * twiddle(mode)
int launch_transpose(int scalar, const void *in, void *out, size_t rows, size_t cols, size_t batch, int mode, hipStream_t st)
{
    if (rows == 0 || cols == 0 || batch == 0) return PCX_OK;
    if (rows > 65535u * 32u || cols > 0x7fffffffu) { set_error("transpose: %zu x %zu exceeds the launch grid", rows, cols); return PCX_ERR_UNSUPPORTED; }
    if (scalar == PCX_F32) return launch_transpose_t<float>(in, out, rows, cols, batch, mode, st);
    if (scalar == PCX_F64) return launch_transpose_t<double>(in, out, rows, cols, batch, mode, st);
    set_error("transpose: unsupported scalar type %d", scalar);
    return PCX_ERR_ARG;
}

}  // namespace pcx
