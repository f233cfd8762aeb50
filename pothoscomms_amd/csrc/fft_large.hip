// fft_large.hip -- power-of-two transforms too long for one workgroup's LDS (complex_float32 beyond
// 16384 bins, complex_float64 beyond 4096): the four-step decomposition N = N1 * N2 around the
// single-workgroup kernels.  With n = N2*n1 + n2 and k = k1 + N1*k2,
//     X[k1 + N1 k2] = sum_n2 W_N2^(n2 k2) * [ W_N^(n2 k1) * sum_n1 x[N2 n1 + n2] W_N1^(n1 k1) ]
// so a frame viewed as an N1 x N2 matrix is  transposed, transformed along N1 (N2 short frames),
// transposed back with the twiddle W_N^(n2 k1) applied on the way, transformed along N2, and
// transposed once more into natural order.  Same unnormalised DFT as kissfft<T>::transform
// (fft/kissfft.hh:81-161), which accepts any size; parity bar 1e-5 of max|X|.  This file holds the
// batched tiled transpose (+ twiddle); pcx_fft_api.hip strings the five launches together.
#include "fft4096.hpp"
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

// --------------------------------------------------------------------------------- //
// complex_float32, 128- and 256-bin transforms over STRIDED frames, for the four-step plans that
// fit two or three passes (numBins <= 4 Mi): 256 lanes carry FPW = 4096/N frames, the tile of
// 4096 elements is copied between global memory and a padded LDS image in whatever order makes
// 16+ consecutive lanes touch consecutive addresses, and the radix-16 passes of fft4096.hpp run
// on the image.  Frame f, element e of a group sits at  base + e*es + f*fs  with
//   (es, fs) = (n2, 1)  the columns of an n1 x n2 matrix (pass 1: transform along n1), or
//   (es, fs) = (1, N)   contiguous frames, or
//   (es, fs) = (n1, 1)  on the way out of pass 2: element k2 of row k1 lands at k2*n1 + k1,
//                       i.e. the final transpose rides on the store.
// tw != 0: results are multiplied by W^(col * bin), W = exp(-+ j 2 pi / numBins), col = the frame's
// column index (four-step twiddle), evaluated in double from two sincospi calls per lane.
// --------------------------------------------------------------------------------- //
using namespace fft4k;

template <int LOG2N>
struct SPlan {
    static_assert(LOG2N == 7 || LOG2N == 8, "strided plans: 128 or 256 bins");
    static constexpr int N = 1 << LOG2N;
    static constexpr int LPF = N / 16;
    static constexpr int FPW = 256 / LPF;                 // 32 or 16 frames per workgroup
    static constexpr int R = LOG2N == 7 ? 8 : 1;          // 128 = 16 x 8, 256 = 16 x 16
    static constexpr int FS = N + N / 16 + 1;             // LDS frame stride: +1 keeps the strided copies conflict-free
    static constexpr int NTWF = R > 1 ? (16 / R) * (R - 1) : 0;
    static constexpr int TF_OFF = 15 * 16;                // table layout of make_tw_r16: [15][16], then [NTWF][LPF]
};

struct StridedIo {
    size_t es, fs;        // element / frame stride (in elements)
    size_t gs;            // distance between consecutive groups of FPW frames inside a batch
    size_t gpb;           // groups per batch
    size_t bs;            // distance between batches
};

template <int LOG2N, bool INV>
__global__ __launch_bounds__(256) void fft_r16_strided_kernel(const float2 *__restrict__ in, float2 *__restrict__ out, size_t ngroups,
                                                              const float2 *__restrict__ twtab, StridedIo si, StridedIo so, double tw)
{
    typedef SPlan<LOG2N> P;
    constexpr int N = P::N, LPF = P::LPF, FPW = P::FPW, R = P::R;
    __shared__ cf img[FPW * P::FS];
    const int tid = threadIdx.x;
    const int fi = tid / LPF, l = tid % LPF;
    cf *lds = img + fi * P::FS;
    const cf *tab = reinterpret_cast<const cf *>(twtab);
    LaneTw t2;
    cf tf[P::NTWF > 0 ? P::NTWF : 1];
    if (R == 1) {
#pragma unroll
        for (int p = 0; p < 3; p++) t2.a[p] = tab[p * 16 + (l & 15)];
#pragma unroll
        for (int p = 0; p < 12; p++) t2.c[p] = tab[(3 + p) * 16 + (l & 15)];
    }
#pragma unroll
    for (int p = 0; p < P::NTWF; p++) tf[p] = tab[P::TF_OFF + p * LPF + l];

    for (size_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const size_t b = g / si.gpb, gi = g % si.gpb;
        const float2 *src = in + b * si.bs + gi * si.gs;
        float2 *dst = out + b * so.bs + gi * so.gs;
        // ---- tile in: 16 elements per lane, consecutive lanes on consecutive addresses ----
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int idx = i * 256 + tid;
            const int f = si.fs == 1 ? idx % FPW : idx / N, e = si.fs == 1 ? idx / FPW : idx % N;
            const float2 t = src[(size_t)e * si.es + (size_t)f * si.fs];
            img[f * P::FS + e + (e >> 4)] = cf{t.x, INV ? -t.y : t.y};
        }
        __syncthreads();
        cf v[16];
#pragma unroll
        for (int s = 0; s < 16; s++) { const int e = l + s * LPF; v[s] = lds[e + (e >> 4)]; }
        // ---- pass Ns = 1 ----
        fft16_plain(v);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; q++) lds[17 * l + bin_of(q)] = v[q];
        __syncthreads();
        bool natural;
        if (R == 1) {
            // ---- pass Ns = 16 (256 bins) ----
#pragma unroll
            for (int s = 0; s < 16; s++) v[s] = lds[l + (l >> 4) + s * (LPF + LPF / 16)];
            fft16_tw(v, t2);
            natural = false;
        } else {
            // ---- final radix-8 pass (128 bins) ----
#pragma unroll
            for (int s = 0; s < 16; s++) v[s] = lds[padi(l + s * LPF)];
            constexpr int G = 16 / (R > 1 ? R : 16);
#pragma unroll
            for (int t = 0; t < G; t++) {
#pragma unroll
                for (int r = 1; r < R; r++) v[t + r * G] = cmul1(v[t + r * G], tf[t * (R - 1) + (r - 1)]);
                fft8(v[t], v[t + G], v[t + 2 * G], v[t + 3 * G], v[t + 4 * G], v[t + 5 * G], v[t + 6 * G], v[t + 7 * G]);
            }
            natural = true;
        }
        // ---- results (bin k*LPF + l of frame fi) back into the image, with the four-step twiddle ----
        double wr = 1.0, wi = 0.0, sr = 1.0, sim = 0.0;
        if (tw != 0.0) {
            const double col = (double)(gi * FPW + fi);
            sincospi(tw * col * (double)l, &wi, &wr);             // W^(col * l)
            sincospi(tw * col * (double)LPF, &sim, &sr);          // W^(col * LPF): one step of k
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int q = natural ? k : (4 * (k & 3) + (k >> 2));   // the register holding bin k*LPF + l (inverse of bin_of)
            cf r = v[q];
            if (INV) r.y = -r.y;
            if (tw != 0.0) {
                const float cr = (float)wr, ci = (float)wi;
                r = cf{r.x * cr - r.y * ci, r.x * ci + r.y * cr};
                const double nr = wr * sr - wi * sim, ni = wr * sim + wi * sr;
                wr = nr; wi = ni;
            }
            const int e = k * LPF + l;
            lds[e + (e >> 4)] = r;
        }
        __syncthreads();
        // ---- tile out ----
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int idx = i * 256 + tid;
            const int f = so.fs == 1 ? idx % FPW : idx / N, e = so.fs == 1 ? idx / FPW : idx % N;
            const cf t = img[f * P::FS + e + (e >> 4)];
            dst[(size_t)e * so.es + (size_t)f * so.fs] = make_float2(t.x, t.y);
        }
    }
}

template <int LOG2N>
int launch_strided_t(const void *in, void *out, size_t ngroups, bool inverse, const void *tw, const StridedIo &si, const StridedIo &so,
                     double twf, hipStream_t st)
{
    const unsigned grid = persistent_grid(ngroups, 256 * 4);
    if (inverse)
        hipLaunchKernelGGL((fft_r16_strided_kernel<LOG2N, true>), dim3(grid), dim3(256), 0, st, (const float2 *)in, (float2 *)out, ngroups,
                           (const float2 *)tw, si, so, twf);
    else
        hipLaunchKernelGGL((fft_r16_strided_kernel<LOG2N, false>), dim3(grid), dim3(256), 0, st, (const float2 *)in, (float2 *)out, ngroups,
                           (const float2 *)tw, si, so, twf);
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}

}  // namespace

// Pass 1 of the short four-step plans: `batch` matrices of n1 x n2 (n1 = 2^log2n1 in {128, 256}), transform along
// n1 for every column, multiply element (k1, n2) by W_N^(n2 k1), same layout out.  tw_table = make_tw_r16(log2n1).
int launch_fft_columns(const void *in, void *out, int log2n1, size_t n2, size_t batch, bool inverse, const void *tw_table, hipStream_t st)
{
    const size_t n1 = (size_t)1 << log2n1, fpw = 4096 / n1;
    if (n2 % fpw) { set_error("fft columns: %zu columns not a multiple of %zu", n2, fpw); return PCX_ERR_UNSUPPORTED; }
    const StridedIo io{n2, 1, fpw, n2 / fpw, n1 * n2};
    const double twf = (inverse ? 2.0 : -2.0) / ((double)n1 * (double)n2);
    if (log2n1 == 8) return launch_strided_t<8>(in, out, batch * (n2 / fpw), inverse, tw_table, io, io, twf, st);
    if (log2n1 == 7) return launch_strided_t<7>(in, out, batch * (n2 / fpw), inverse, tw_table, io, io, twf, st);
    set_error("fft columns: no plan for 2^%d", log2n1);
    return PCX_ERR_UNSUPPORTED;
}
// Pass 2 when n2 is 128 or 256 too: `batch` matrices of n1 rows x n2, transform every row and store element k2
// of row k1 at k2*n1 + k1 (natural order of the long transform).  tw_table = make_tw_r16(log2n2).
int launch_fft_rows_transposed(const void *in, void *out, size_t n1, int log2n2, size_t batch, bool inverse, const void *tw_table, hipStream_t st)
{
    const size_t n2 = (size_t)1 << log2n2, fpw = 4096 / n2;
    if (n1 % fpw) { set_error("fft rows: %zu rows not a multiple of %zu", n1, fpw); return PCX_ERR_UNSUPPORTED; }
    const StridedIo si{1, n2, fpw * n2, n1 / fpw, n1 * n2};
    const StridedIo so{n1, 1, fpw, n1 / fpw, n1 * n2};
    if (log2n2 == 8) return launch_strided_t<8>(in, out, batch * (n1 / fpw), inverse, tw_table, si, so, 0.0, st);
    if (log2n2 == 7) return launch_strided_t<7>(in, out, batch * (n1 / fpw), inverse, tw_table, si, so, 0.0, st);
    set_error("fft rows: no plan for 2^%d", log2n2);
    return PCX_ERR_UNSUPPORTED;
}

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
