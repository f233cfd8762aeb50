// fft_r16.hip -- complex_float32 FFT for every power-of-two numBins from 16 to 16384 on the
// register-resident radix-16 Stockham pipeline of fft4096.hpp -- including numBins = 4096 (BASELINE configs[2]): the
// dedicated persistent kernel of fft.hip (register prefetch, dynamic dealing) measured 3-10 % slower than this one with
// three frames per workgroup and is kept in the diagnostic library only (tools/ab_fft4096_family.sh).
//
// N = 16^A * R, R in {1, 2, 4, 8}: A radix-16 passes (sub-transform sizes Ns = 1, 16, 256) and,
// when R > 1, one final radix-R pass.  Every lane holds 16 points of one frame, a frame takes
// LPF = N/16 lanes, a 256-lane workgroup carries 256/LPF frames at once (N < 4096) so small
// transforms still fill the waves; N = 8192 / 16384 use 512 / 1024 lanes per frame.
// In every pass lane l reads x[l + s*LPF], s = 0..15 -- the radix-R pass does 16/R butterflies
// on (s = t + r*16/R) -- so global loads/stores are unit-stride across the lanes of a frame in
// all passes and the LDS image (padded i + i/16 per frame) is shared by all plans.
// Twiddles are lane constants (fft4096.hpp): loaded once per persistent workgroup.
//
// Same transform as kissfft<float> (fft/kissfft.hh:81-161): forward exp(-j..), inverse exp(+j..)
// taken as conj(FFT(conj x)), unscaled.  Parity bar 1e-5 of max|X|.
#include "fft4096.hpp"
#include <cstdlib>
#include "pcx_internal.hpp"

namespace pcx {

namespace {
using namespace fft4k;

template <int LOG2N>
struct Plan {
    static constexpr int N = 1 << LOG2N;
    static constexpr int LPF = N / 16;                    // lanes per frame
    static constexpr int A = LOG2N / 4;                   // radix-16 passes
    static constexpr int R = 1 << (LOG2N % 4);            // final radix (1: none)
    static constexpr int THREADS = LPF < 256 ? 256 : LPF;
    static constexpr int FPW = THREADS / LPF;             // frames per workgroup
    static constexpr int LDS_FRAME = (A + (R > 1 ? 1 : 0)) > 1 ? N + N / 16 : 0;
    // numBins <= 64: a frame is 128..512 bytes and lane l's elements l + s*LPF of it sit 8..32 bytes apart,
    // so loading them straight from global memory touches 64 lines per wave-instruction (measured 0.5 TB/s
    // at 16 bins).  The 256 lanes' frames are one contiguous 32 KiB chunk: copied in and out with
    // lane-contiguous 16-byte accesses through a padded LDS image (i + i/16) that shares the exchange space.
    // 8192 / 16384 bins: one frame takes 8 / 16 waves, so a second resident frame (8192) or any at all (16384)
    // needs <= 128 VGPRs: the Ns = 16 pass's 15 lane constants (they depend on l & 15 only) come from a
    // 1.9 KB LDS table instead of 30 registers, and the register budget is pinned with launch bounds
    static constexpr bool T2_LDS = LOG2N >= 13;
    static constexpr int T2_ELEMS = T2_LDS ? 240 : 0;
    static constexpr bool STAGED = LOG2N <= 6;
    static constexpr int STAGE = STAGED ? 4096 + 4096 / 16 : 0;
    static constexpr int NTWF = R > 1 ? (16 / R) * (R - 1) : 0;   // final-pass twiddles per lane
    // device table: [15][16] pass Ns=16, [15][256] pass Ns=256 (A >= 3), [NTWF][LPF] final pass
    static constexpr int T2_OFF = 0;
    static constexpr int T3_OFF = 15 * 16;
    static constexpr int TF_OFF = T3_OFF + (A >= 3 ? 15 * 256 : 0);
    static constexpr int TABLE = TF_OFF + NTWF * LPF;
};

template <int LOG2N, bool INV, int SAUX>
__global__ __launch_bounds__(Plan<LOG2N>::THREADS, Plan<LOG2N>::THREADS >= 512 ? 4 : 1) void fft_r16_kernel(const float2 *__restrict__ in, float2 *__restrict__ out,
                                                                       size_t nframes, const float2 *__restrict__ twtab)
{
    typedef Plan<LOG2N> P;
    constexpr int N = P::N, LPF = P::LPF, A = P::A, R = P::R, FPW = P::FPW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf *lds_all = reinterpret_cast<cf *>(smem_raw);
    const int tid = threadIdx.x;
    const int fi = tid / LPF, l = tid % LPF;
    cf *lds = lds_all + fi * P::LDS_FRAME;
    const cf *tab = reinterpret_cast<const cf *>(twtab);

    // lane constants
    LaneTw t2, t3;
    cf tf[P::NTWF > 0 ? P::NTWF : 1];
    cf *t2tab = lds_all + (size_t)P::LDS_FRAME * FPW;    // behind the frame images (T2_LDS plans)
    if (A >= 2 && P::T2_LDS) {
        for (int i = tid; i < 240; i += P::THREADS) t2tab[i] = tab[P::T2_OFF + i];
    } else if (A >= 2) {
#pragma unroll
        for (int p = 0; p < 3; p++) t2.a[p] = tab[P::T2_OFF + p * 16 + (l & 15)];
#pragma unroll
        for (int p = 0; p < 12; p++) t2.c[p] = tab[P::T2_OFF + (3 + p) * 16 + (l & 15)];
    }
    if (A >= 3) {
#pragma unroll
        for (int p = 0; p < 3; p++) t3.a[p] = tab[P::T3_OFF + p * 256 + (l & 255)];
#pragma unroll
        for (int p = 0; p < 12; p++) t3.c[p] = tab[P::T3_OFF + (3 + p) * 256 + (l & 255)];
    }
#pragma unroll
    for (int p = 0; p < P::NTWF; p++) tf[p] = tab[P::TF_OFF + p * LPF + l];

    const size_t ngroups = (nframes + FPW - 1) / FPW;
    const unsigned voff = (unsigned)(fi * N + l) * 8u;
    for (size_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const size_t f0 = g * FPW;
        const size_t valid = nframes - f0 < (size_t)FPW ? nframes - f0 : (size_t)FPW;
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(in + f0 * N, (unsigned)(valid * N * 8));
        cf v[16];
        if (P::STAGED) {
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            __syncthreads();   // the image is free (previous group's stores / passes are done)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int e = (i * 256 + tid) * 2;      // two elements per lane and access, lane-contiguous
                const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, e * 8, 0, kAuxStream);
                const int pe = e + (e >> 4);
                lds_all[pe] = cf{__uint_as_float(t.x), __uint_as_float(t.y)};
                lds_all[pe + 1] = cf{__uint_as_float(t.z), __uint_as_float(t.w)};
            }
            __syncthreads();
#pragma unroll
            for (int s = 0; s < 16; s++) {
                const int e = fi * N + l + s * LPF;
                const cf t = lds_all[e + (e >> 4)];
                v[s] = cf{t.x, INV ? -t.y : t.y};
            }
        } else {
#pragma unroll
            for (int s = 0; s < 16; s++) {
                const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)voff, s * LPF * 8, kAuxStream);
                v[s] = cf{__uint_as_float(t.x), INV ? -__uint_as_float(t.y) : __uint_as_float(t.y)};
            }
        }
        const __amdgpu_buffer_rsrc_t ws = make_rsrc(out + f0 * N, (unsigned)(valid * N * 8));
        bool natural = false;   // are the results in v[] in natural order (true) or in bin_of order?
        // ---- pass Ns = 1 ----
        fft16_plain(v);
        if (A == 1 && R == 1) {
            // N = 16: a frame per lane, done
        } else {
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 16; q++) lds[17 * l + bin_of(q)] = v[q];
        }
        // ---- pass Ns = 16 ----
        if (A >= 2) {
            __syncthreads();
#pragma unroll
            for (int s = 0; s < 16; s++) v[s] = lds[l + (l >> 4) + s * (LPF + LPF / 16)];
            if (P::T2_LDS) {
                LaneTw tl;     // first used behind at least two barriers after the table was staged
#pragma unroll
                for (int p = 0; p < 3; p++) tl.a[p] = t2tab[p * 16 + (l & 15)];
#pragma unroll
                for (int p = 0; p < 12; p++) tl.c[p] = t2tab[(3 + p) * 16 + (l & 15)];
                fft16_tw(v, tl);
            } else {
                fft16_tw(v, t2);
            }
            if (!(A == 2 && R == 1)) {
                __syncthreads();
                const int wb = (l >> 4) * 272 + (l & 15);
#pragma unroll
                for (int q = 0; q < 16; q++) lds[wb + 17 * bin_of(q)] = v[q];
            }
        }
        // ---- pass Ns = 256 ----
        if (A >= 3) {
            __syncthreads();
#pragma unroll
            for (int s = 0; s < 16; s++) v[s] = lds[l + (l >> 4) + s * (LPF + LPF / 16)];
            fft16_tw(v, t3);
            if (R > 1) {
                __syncthreads();
                const int wb = (l >> 8) * 4352 + (l & 255) + ((l & 255) >> 4);
#pragma unroll
                for (int q = 0; q < 16; q++) lds[wb + 272 * bin_of(q)] = v[q];
            }
        }
        // ---- final radix-R pass (Ns = 16^A) ----
        if (R > 1) {
            __syncthreads();
#pragma unroll
            for (int s = 0; s < 16; s++) v[s] = lds[padi(l + s * LPF)];
            constexpr int G = 16 / R;   // butterflies per lane; butterfly t uses v[t + r*G], twiddle (W_N^(l + t*LPF))^r
#pragma unroll
            for (int t = 0; t < G; t++) {
#pragma unroll
                for (int r = 1; r < R; r++) v[t + r * G] = cmul1(v[t + r * G], tf[t * (R - 1) + (r - 1)]);
                if (R == 2) {
                    const cf a = v[t], b = v[t + G];
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
        // ---- store: X at frame offset (see each pass's output map); all are l + k*LPF ----
        if (P::STAGED) {
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            __syncthreads();   // every lane is done reading the image (exchange or staging)
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int k = natural ? q : bin_of(q);
                const int e = fi * N + l + k * LPF;
                lds_all[e + (e >> 4)] = INV ? cf{v[q].x, -v[q].y} : v[q];
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int e = (i * 256 + tid) * 2;
                const int pe = e + (e >> 4);
                const cf a = lds_all[pe], b = lds_all[pe + 1];
                const u32x4 t = {__float_as_uint(a.x), __float_as_uint(a.y), __float_as_uint(b.x), __float_as_uint(b.y)};
                __builtin_amdgcn_raw_buffer_store_b128(t, ws, e * 8, 0, SAUX);   // past `valid` frames: dropped by the range check
            }
        } else {
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int k = natural ? q : bin_of(q);
            // last radix-16 pass with Ns: out index = (l / Ns) * 16 Ns + (l % Ns) + k * Ns, and LPF = Ns there
            const cf r = INV ? cf{v[q].x, -v[q].y} : v[q];
            const u32x2 t = {__float_as_uint(r.x), __float_as_uint(r.y)};
            __builtin_amdgcn_raw_buffer_store_b64(t, ws, (int)voff, k * LPF * 8, SAUX);
        }
        }
    }
}

template <int LOG2N>
int launch_r16(const void *in, void *out, size_t nframes, bool inverse, const void *tw, hipStream_t st)
{
    typedef Plan<LOG2N> P;
    size_t lds = ((size_t)P::LDS_FRAME * P::FPW + P::T2_ELEMS) * sizeof(cf);
    if ((size_t)P::STAGE * sizeof(cf) > lds) lds = (size_t)P::STAGE * sizeof(cf);
    // PCX_FFT_STORE_AUX (A/B): cache-policy bits of the output stores; default 2 = non-temporal
    const int saux = (int)PCX_ENV_INT("PCX_FFT_STORE_AUX", 2);
    auto k = saux == 2 ? (inverse ? fft_r16_kernel<LOG2N, true, 2> : fft_r16_kernel<LOG2N, false, 2>)
                       : (inverse ? fft_r16_kernel<LOG2N, true, 0> : fft_r16_kernel<LOG2N, false, 0>);
    if (lds > 64 * 1024) PCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const size_t ngroups = (nframes + P::FPW - 1) / P::FPW;
    unsigned per_cu = lds ? (unsigned)(160 * 1024 / lds) : 8;
    const unsigned by_threads = 2048 / P::THREADS;
    if (per_cu > by_threads) per_cu = by_threads;
    if (per_cu > 4) per_cu = 4;
    if (per_cu < 1) per_cu = 1;
    // workgroups of 1-3 groups each, whatever the call size (pcx_internal.hpp rounds_grid; measured per size at 64 Mi and 256 Mi
    // samples per launch, run-to-run noise of a few per cent: 1 group per workgroup to 64 bins, 2 to 512, 3 from 1024); 16384 bins (16 waves per frame, one
    // frame per CU) stays on equal shares over the resident slots: -7..-17 % with anything queued behind them.
    // PCX_OVERSUB (diagnostic library) brings the fixed-factor grid back for A/B.
    const unsigned grid = PCX_ENV_INT("PCX_OVERSUB", 0) > 0 || LOG2N >= 14 ? persistent_grid(ngroups, 256 * per_cu, 1)
                                                                          : rounds_grid(ngroups, 256 * per_cu, LOG2N <= 6 ? 1 : LOG2N <= 9 ? 2 : 3);
    hipLaunchKernelGGL(k, dim3(grid), dim3(P::THREADS), lds, st, (const float2 *)in, (float2 *)out, nframes, (const float2 *)tw);
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}

}  // namespace

// number of float2 entries of the lane-constant table for numBins = 2^log2n
size_t fft_r16_table_elems(int log2n)
{
    const int N = 1 << log2n, LPF = N / 16, A = log2n / 4, R = 1 << (log2n % 4);
    return (size_t)15 * 16 + (A >= 3 ? 15 * 256 : 0) + (R > 1 ? (16 / R) * (R - 1) * LPF : 0);
}

int launch_fft_r16_cf32(const void *in, void *out, int log2n, size_t nframes, bool inverse, const void *tw, hipStream_t st)
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
    case 14: return launch_r16<14>(in, out, nframes, inverse, tw, st);
    }
    set_error("fft r16: log2(numBins) = %d has no plan", log2n);
    return PCX_ERR_UNSUPPORTED;
}

}  // namespace pcx
