// fft_r16_f64.hip -- complex_float64 FFT for every power-of-two numBins from 16 to 8192 on the
// register-resident radix-16 Stockham plan of fft_r16.hip, restated in double precision.
//
// Same decomposition, lane maps and LDS image as the float kernel (N = 16^A * R, LPF = N/16 lanes per
// frame, lane l holds x[l + s*LPF], image padded i + i/16); what differs is what the double type forces:
//   * an element is 16 bytes: one buffer_load_b128 / ds_read_b128 per point, a frame image of 4096
//     points is 69.6 KB, so two workgroups fit a CU (the float kernel runs four);
//   * 16 points are 64 VGPRs and one pass's 15 lane constants another 60: the Ns = 16 constants are
//     read from a 3.8 KB LDS table at their use, only the Ns = 256 set stays in registers;
//   * arithmetic is plain scalar FMA (v_fma_f64): there is no packed double pipe to write asm for.
// At ~2 flop per streamed byte the transform stays HBM-bound on the 78 TFLOP/s vector-f64 pipe.
//
// Same transform as kissfft<double> (fft/kissfft.hh:81-161): forward exp(-j..), inverse exp(+j..) taken
// as conj(FFT(conj x)), unscaled.  Twiddles are generated on the host in double precision.
#include "fft_f64.hpp"
#include <cstdlib>
#include "pcx_internal.hpp"

namespace pcx {

namespace {

using namespace fft64;

template <int LOG2N>
struct Plan {
    static constexpr int N = 1 << LOG2N;
    static constexpr int LPF = N / 16;
    static constexpr int A = LOG2N / 4;
    static constexpr int R = 1 << (LOG2N % 4);
    static constexpr int THREADS = LPF < 256 ? 256 : LPF;
    static constexpr int FPW = THREADS / LPF;
    static constexpr bool STAGED = LOG2N <= 5;      // lanes' points of a frame closer than 64 bytes: stage through LDS
    static constexpr int PASSES = A + (R > 1 ? 1 : 0);
    static constexpr int IMAGE = (PASSES > 1 || STAGED) ? (N * FPW) + (N * FPW) / 16 : 0;
    static constexpr int LDS_FRAME = N + N / 16;
    static constexpr int T2_ELEMS = A >= 2 ? 240 : 0;
    static constexpr int NTWF = R > 1 ? (16 / R) * (R - 1) : 0;
    static constexpr int T2_OFF = 0;
    static constexpr int T3_OFF = 15 * 16;
    static constexpr int TF_OFF = T3_OFF + (A >= 3 ? 15 * 256 : 0);
};

template <int LOG2N, bool INV>
__global__ __launch_bounds__(Plan<LOG2N>::THREADS) void fft_r16_f64_kernel(const double2 *__restrict__ in, double2 *__restrict__ out,
                                                                          size_t nframes, const double2 *__restrict__ twtab)
{
    typedef Plan<LOG2N> P;
    constexpr int N = P::N, LPF = P::LPF, A = P::A, R = P::R, FPW = P::FPW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cd *lds_all = reinterpret_cast<cd *>(smem_raw);
    const int tid = threadIdx.x;
    const int fi = tid / LPF, l = tid % LPF;
    cd *lds = lds_all + fi * P::LDS_FRAME;
    const cd *tab = reinterpret_cast<const cd *>(twtab);

    cd t3[A >= 3 ? 15 : 1];
    cd tf[P::NTWF > 0 ? P::NTWF : 1];
    cd *t2tab = lds_all + P::IMAGE;
    if (A >= 2)
        for (int i = tid; i < 240; i += P::THREADS) t2tab[i] = tab[P::T2_OFF + i];
    if (A >= 3) {
#pragma unroll
        for (int p = 0; p < 15; p++) t3[p] = tab[P::T3_OFF + p * 256 + (l & 255)];
    }
#pragma unroll
    for (int p = 0; p < P::NTWF; p++) tf[p] = tab[P::TF_OFF + p * LPF + l];

    const size_t ngroups = (nframes + FPW - 1) / FPW;
    const unsigned voff = (unsigned)(fi * N + l) * 16u;
    for (size_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const size_t f0 = g * FPW;
        const size_t valid = nframes - f0 < (size_t)FPW ? nframes - f0 : (size_t)FPW;
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(in + f0 * N, (unsigned)(valid * N * 16));
        const __amdgpu_buffer_rsrc_t ws = make_rsrc(out + f0 * N, (unsigned)(valid * N * 16));
        cd v[16];
        if (P::STAGED) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int e = i * 256 + tid;
                lds_all[padi(e)] = as_cd(__builtin_amdgcn_raw_buffer_load_b128(rs, e * 16, 0, kAuxStream));
            }
            __syncthreads();
#pragma unroll
            for (int s = 0; s < 16; s++) v[s] = conj_if(INV, lds_all[padi(fi * N + l + s * LPF)]);
        } else {
#pragma unroll
            for (int s = 0; s < 16; s++) v[s] = conj_if(INV, as_cd(__builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, s * LPF * 16, kAuxStream)));
        }
        bool natural = false;
        fft16_plain(v);
        if (!(A == 1 && R == 1)) {
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 16; q++) lds[17 * l + bin_of(q)] = v[q];
        }
        if (A >= 2) {
            __syncthreads();
#pragma unroll
            for (int s = 0; s < 16; s++) v[s] = lds[l + (l >> 4) + s * (LPF + LPF / 16)];
            const cd *t2 = t2tab + (l & 15);
            fft16_tw(v, [&](int p) { return t2[p * 16]; });
            if (!(A == 2 && R == 1)) {
                __syncthreads();
                const int wb = (l >> 4) * 272 + (l & 15);
#pragma unroll
                for (int q = 0; q < 16; q++) lds[wb + 17 * bin_of(q)] = v[q];
            }
        }
        if (A >= 3) {
            __syncthreads();
#pragma unroll
            for (int s = 0; s < 16; s++) v[s] = lds[l + (l >> 4) + s * (LPF + LPF / 16)];
            fft16_tw(v, [&](int p) { return t3[p]; });
            if (R > 1) {
                __syncthreads();
                const int wb = (l >> 8) * 4352 + (l & 255) + ((l & 255) >> 4);
#pragma unroll
                for (int q = 0; q < 16; q++) lds[wb + 272 * bin_of(q)] = v[q];
            }
        }
        if (R > 1) {
            __syncthreads();
#pragma unroll
            for (int s = 0; s < 16; s++) v[s] = lds[padi(l + s * LPF)];
            constexpr int G = 16 / R;
#pragma unroll
            for (int t = 0; t < G; t++) {
#pragma unroll
                for (int r = 1; r < R; r++) v[t + r * G] = cmul(v[t + r * G], tf[t * (R - 1) + (r - 1)]);
                if (R == 2) {
                    const cd a = v[t], b = v[t + G];
                    v[t] = a + b;
                    v[t + G] = a - b;
                } else if (R == 4) {
                    fft4(v[t], v[t + G], v[t + 2 * G], v[t + 3 * G]);
                } else {
                    fft8(v[t], v[t + G], v[t + 2 * G], v[t + 3 * G], v[t + 4 * G], v[t + 5 * G], v[t + 6 * G], v[t + 7 * G]);
                }
            }
            natural = true;
        }
        if (P::STAGED) {
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int k = natural ? q : bin_of(q);
                lds_all[padi(fi * N + l + k * LPF)] = conj_if(INV, v[q]);
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int e = i * 256 + tid;
                __builtin_amdgcn_raw_buffer_store_b128(as_u4(lds_all[padi(e)]), ws, e * 16, 0, kAuxStream);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int k = natural ? q : bin_of(q);
                __builtin_amdgcn_raw_buffer_store_b128(as_u4(conj_if(INV, v[q])), ws, (int)voff, k * LPF * 16, kAuxStream);
            }
        }
    }
}

template <int LOG2N>
int launch_r16(const void *in, void *out, size_t nframes, bool inverse, const void *tw, hipStream_t st)
{
    typedef Plan<LOG2N> P;
    const size_t lds = ((size_t)P::IMAGE + P::T2_ELEMS) * sizeof(cd);
    auto k = inverse ? fft_r16_f64_kernel<LOG2N, true> : fft_r16_f64_kernel<LOG2N, false>;
    if (lds > 64 * 1024) PCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const size_t ngroups = (nframes + P::FPW - 1) / P::FPW;
    unsigned per_cu = lds ? (unsigned)(160 * 1024 / lds) : 2;
    if (per_cu > 2) per_cu = 2;     // ~200 VGPRs: two waves per SIMD
    if (P::THREADS > 256) per_cu = 1;
    if (per_cu < 1) per_cu = 1;
    const long f64_rounds = PCX_ENV_INT("PCX_F64_ROUNDS", 0);   // (diagnostic library: groups per workgroup instead of the fixed factor, A/B)
    const unsigned grid = f64_rounds > 0 ? rounds_grid(ngroups, 256 * per_cu, (unsigned)f64_rounds)
                                         : persistent_grid(ngroups, 256 * per_cu, 4);   // four queued per slot: +3..5 % (tools/sweep_fft_f64.py, PCX_OVERSUB A/B)
    hipLaunchKernelGGL(k, dim3(grid), dim3(P::THREADS), lds, st, (const double2 *)in, (double2 *)out, nframes, (const double2 *)tw);
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}

}  // namespace

int launch_fft_r16_cf64(const void *in, void *out, int log2n, size_t nframes, bool inverse, const void *tw, hipStream_t st)
{
    if (nframes == 0) return PCX_OK;
    switch (log2n) {
    case 4: return launch_r16<4>(in, out, nframes, inverse, tw, st);
    case 5: return launch_r16<5>(in, out, nframes, inverse, tw, st);
    case 6: return launch_r16<6>(in, out, nframes, inverse, tw, st);
    case 7: return launch_r16<7>(in, out, nframes, inverse, tw, st);
    case 8: return launch_r16<8>(in, out, nframes, inverse, tw, st);
    case 9: return launch_r16<9>(in, out, nframes, inverse, tw, st);
    case 10: return launch_r16<10>(in, out, nframes, inverse, tw, st);
    case 11: return launch_r16<11>(in, out, nframes, inverse, tw, st);
    case 12: return launch_r16<12>(in, out, nframes, inverse, tw, st);
    case 13: return launch_r16<13>(in, out, nframes, inverse, tw, st);
    }
    set_error("fft r16 f64: log2(numBins) = %d has no plan", log2n);
    return PCX_ERR_UNSUPPORTED;
}

}  // namespace pcx
