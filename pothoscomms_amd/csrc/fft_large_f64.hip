// fft_large_f64.hip -- complex_float64 counterpart of fft_large.hip's strided 128- / 256-bin transforms, for the
// two- and three-pass four-step plans of power-of-two sizes beyond 8192 bins (to 2 Mi bins).
//
// Same tiling as the float kernel: 256 lanes carry FPW = 4096/N frames; the tile of 4096 elements is copied
// between global memory and a padded LDS image in whatever order makes consecutive lanes touch consecutive
// addresses; the radix-16 passes of fft_f64.hpp run on the image; frame f, element e of a group sits at
// base + e*es + f*fs (columns of an n1 x n2 matrix, contiguous frames, or the transposing store of pass 2).
// An element is 16 bytes, so the image is 70 KB (dynamic LDS, two workgroups per CU) and the Ns = 16 pass reads
// its 15 lane constants from an LDS table instead of holding 60 more VGPRs.
// tw != 0: results are multiplied by W^(col * bin), W = exp(-+ j 2 pi / numBins) (four-step twiddle), evaluated
// in double from two sincospi calls per lane and a recurrence over the lane's 16 bins.
// Same unnormalised DFT as kissfft<double>::transform (fft/kissfft.hh:81-161); parity bar 1e-13 of max|X|.
#include "fft_f64.hpp"
#include "pcx_internal.hpp"

namespace pcx {

namespace {
using namespace fft64;

template <int LOG2N>
struct SPlan {
    static_assert(LOG2N == 7 || LOG2N == 8, "strided plans: 128 or 256 bins");
    static constexpr int N = 1 << LOG2N;
    static constexpr int LPF = N / 16;
    static constexpr int FPW = 256 / LPF;
    static constexpr int R = LOG2N == 7 ? 8 : 1;
    static constexpr int FS = N + N / 16 + 1;             // +1 keeps the strided copies conflict-free
    static constexpr int NTWF = R > 1 ? (16 / R) * (R - 1) : 0;
    static constexpr int TF_OFF = 15 * 16;
    static constexpr int IMG = FPW * FS;
    static constexpr int T2 = R == 1 ? 240 : 0;
};

struct StridedIo {
    size_t es, fs;        // element / frame stride (in elements)
    size_t gs;            // distance between consecutive groups of FPW frames inside a batch
    size_t gpb;           // groups per batch
    size_t bs;            // distance between batches
};

template <int LOG2N, bool INV>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void fft_r16_strided_f64_kernel(const double2 *__restrict__ in, double2 *__restrict__ out, size_t ngroups,
                                                                  const double2 *__restrict__ twtab, StridedIo si, StridedIo so, double tw)
{
    typedef SPlan<LOG2N> P;
    constexpr int N = P::N, LPF = P::LPF, FPW = P::FPW, R = P::R;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cd *img = reinterpret_cast<cd *>(smem_raw);
    cd *t2tab = img + P::IMG;
    const int tid = threadIdx.x;
    const int fi = tid / LPF, l = tid % LPF;
    cd *lds = img + fi * P::FS;
    const cd *tab = reinterpret_cast<const cd *>(twtab);
    cd tf[P::NTWF > 0 ? P::NTWF : 1];
    if (R == 1)
        for (int i = tid; i < 240; i += 256) t2tab[i] = tab[i];
#pragma unroll
    for (int p = 0; p < P::NTWF; p++) tf[p] = tab[P::TF_OFF + p * LPF + l];

    for (size_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const size_t b = g / si.gpb, gi = g % si.gpb;
        const cd *src = reinterpret_cast<const cd *>(in) + b * si.bs + gi * si.gs;
        cd *dst = reinterpret_cast<cd *>(out) + b * so.bs + gi * so.gs;
        // the four-step twiddle of this lane's first bin and its step over k, while few registers are live
        cd w = {1.0, 0.0}, step = {1.0, 0.0};
        if (tw != 0.0) {
            const double col = (double)(gi * FPW + fi);
            double s0, c0, s1, c1;
            sincospi(tw * col * (double)l, &s0, &c0);             // W^(col * l)
            sincospi(tw * col * (double)LPF, &s1, &c1);           // W^(col * LPF): one step of k
            w = cd{c0, s0};
            step = cd{c1, s1};
        }
        __syncthreads();
#pragma unroll 8
        for (int i = 0; i < 16; i++) {
            const int idx = i * 256 + tid;
            const int f = si.fs == 1 ? idx % FPW : idx / N, e = si.fs == 1 ? idx / FPW : idx % N;
            img[f * P::FS + e + (e >> 4)] = conj_if(INV, src[(size_t)e * si.es + (size_t)f * si.fs]);
        }
        __syncthreads();
        cd v[16];
#pragma unroll
        for (int s = 0; s < 16; s++) { const int e = l + s * LPF; v[s] = lds[e + (e >> 4)]; }
        fft16_plain(v);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; q++) lds[17 * l + bin_of(q)] = v[q];
        __syncthreads();
        bool natural;
        if (R == 1) {
#pragma unroll
            for (int s = 0; s < 16; s++) v[s] = lds[l + (l >> 4) + s * (LPF + LPF / 16)];
            const cd *t2 = t2tab + (l & 15);
            fft16_tw(v, [&](int p) { return t2[p * 16]; });
            natural = false;
        } else {
#pragma unroll
            for (int s = 0; s < 16; s++) v[s] = lds[padi(l + s * LPF)];
            constexpr int G = 16 / (R > 1 ? R : 16);
#pragma unroll
            for (int t = 0; t < G; t++) {
#pragma unroll
                for (int r = 1; r < R; r++) v[t + r * G] = cmul(v[t + r * G], tf[t * (R - 1) + (r - 1)]);
                fft8(v[t], v[t + G], v[t + 2 * G], v[t + 3 * G], v[t + 4 * G], v[t + 5 * G], v[t + 6 * G], v[t + 7 * G]);
            }
            natural = true;
        }
        // results (bin k*LPF + l of frame fi) back into the image, with the four-step twiddle
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int q = natural ? k : (4 * (k & 3) + (k >> 2));   // the register holding bin k*LPF + l
            cd r = conj_if(INV, v[q]);
            if (tw != 0.0) {
                r = cmul(r, w);
                w = cmul(w, step);
            }
            const int e = k * LPF + l;
            lds[e + (e >> 4)] = r;
        }
        __syncthreads();
#pragma unroll 8
        for (int i = 0; i < 16; i++) {
            const int idx = i * 256 + tid;
            const int f = so.fs == 1 ? idx % FPW : idx / N, e = so.fs == 1 ? idx / FPW : idx % N;
            dst[(size_t)e * so.es + (size_t)f * so.fs] = img[f * P::FS + e + (e >> 4)];
        }
    }
}

template <int LOG2N>
int launch_strided_t(const void *in, void *out, size_t ngroups, bool inverse, const void *tw, const StridedIo &si, const StridedIo &so,
                     double twf, hipStream_t st)
{
    typedef SPlan<LOG2N> P;
    const size_t lds = (size_t)(P::IMG + P::T2) * sizeof(cd);
    auto k = inverse ? fft_r16_strided_f64_kernel<LOG2N, true> : fft_r16_strided_f64_kernel<LOG2N, false>;
    PCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned grid = persistent_grid(ngroups, 256 * 2);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, (const double2 *)in, (double2 *)out, ngroups, (const double2 *)tw, si, so, twf);
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}

}  // namespace

// complex_float64 forms of launch_fft_columns / launch_fft_rows_transposed (fft_large.hip); tw_table = make_tw_r16<double>
int launch_fft_columns_f64(const void *in, void *out, int log2n1, size_t n2, size_t batch, bool inverse, const void *tw_table, hipStream_t st)
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
int launch_fft_rows_transposed_f64(const void *in, void *out, size_t n1, int log2n2, size_t batch, bool inverse, const void *tw_table, hipStream_t st)
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

}  // namespace pcx
