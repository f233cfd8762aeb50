// fft.hip -- /comms/fft on the device.
//
//   fft4096_kernel     complex_float32, numBins = 4096: radix-16 x 3 Stockham, one frame
//                      per workgroup, registers + one padded LDS image (fft4096.hpp).
//                      HBM-bound: 64 KiB of traffic per 245,760 flop frame.  DIAGNOSTIC A/B ONLY
//                      since round 2 (PCX_FFT4096_DEDICATED): the product runs 4096 bins on
//                      fft_r16.hip's kernel, which measured 3-10 % faster.
//   fft_pow2_kernel    complex_float32 / complex_float64, numBins = 2^k: Stockham
//                      radix-4 passes (+ one radix-2 pass when k is odd), ping-pong LDS.
//   fft_q15_kernel     complex_int16: the reference's fixed-point kiss_fft
//                      (fft/kiss_fft.c, -DFIXED_POINT=16) restated pass by pass so every
//                      rounding (sround, C_FIXDIV by the radix) matches bit for bit.
//
// Replaces kissfft<T>::transform (fft/kissfft.hh:81-161) and kiss_fft (fft/kiss_fft.c:237-302)
// called from FFT::work (fft/FFT.cpp:61-72).
#include "fft4096.hpp"
#include "pcx_sched.hpp"
#include <cstdlib>

#include "pcx_internal.hpp"

namespace pcx {

// --------------------------------------------------------------------------------- //
// 4096-point complex_float32
// --------------------------------------------------------------------------------- //
template <bool INV, int SAUX, bool DYN = false>
__global__ __launch_bounds__(256, 4) void fft4096_kernel(const float2 *__restrict__ in, float2 *__restrict__ out,
                                                      size_t nframes, const float2 *__restrict__ twtab, SchedState *__restrict__ sched, int prio)
{
    using namespace fft4k;
    __shared__ cf lds[LDS_ELEMS];
    __shared__ unsigned sched_slot;
    const int j = threadIdx.x;
    size_t f = blockIdx.x;
    // DYN: frames dealt dynamically, two per draw, the draw one frame ahead of the prefetch (pcx_sched.hpp AheadDealer):
    // a fixed share per workgroup made every launch wait for the slowest CU (0.795 -> 0.758 ms per 65,536 frames with an
    // eight-fold oversubscribed grid doing the balancing, tools/ab_sched.sh)
    AheadDealer deal;
    if (DYN) {
        if (!deal.begin(sched, &sched_slot, nframes, j)) { deal.finish(j); return; }
        f = deal.block();
    } else if (f >= nframes) return;
    // once per workgroup: pass-3 twiddles into registers, pass-2 table into LDS (its first
    // reader sits behind two barriers); the frame loop issues no table loads
    LaneTw tw3;
    load_pass3_twiddles(tw3, twtab, j);
    stage_pass2_twiddles(lds, twtab, j);
    cf nx[16];                    // register prefetch of the next frame
    load_frame<false>(nx, make_rsrc(in + f * N, N * 8), j);
    for (;;) {
        cf v[16];
        // inverse = conj(FFT(conj(x))) on the forward passes (one set of twiddles)
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] = INV ? cf{nx[r].x, -nx[r].y} : nx[r];
        size_t fn = f + gridDim.x;
        const bool more = DYN ? deal.next(&fn) : fn < nframes;
        if (DYN) deal.draw(j);         // AHEAD of the prefetch in the in-order vmcnt queue: its result can be waited for without the frame
        if (more) load_frame<false>(nx, make_rsrc(in + fn * N, N * 8), j);
        pass1(v, lds, j);
        if (DYN) deal.publish(j);      // pass 2 and pass 3 open with barriers
        if (prio == 2) __builtin_amdgcn_s_setprio(1);
        pass2(v, lds, j);
        if (prio == 1) __builtin_amdgcn_s_setprio(1);
        pass3(v, lds, j, tw3);
        const __amdgpu_buffer_rsrc_t ws = make_rsrc(out + f * N, N * 8);
#pragma unroll
        for (int q = 0; q < 16; q++)
            store_cf<SAUX>(ws, (unsigned)(j + 256 * bin_of(q)) * 8u, INV ? cf{v[q].x, -v[q].y} : v[q]);
        if (prio) __builtin_amdgcn_s_setprio(0);
        if (!more) break;
        if (DYN) (void)deal.advance();
        f = fn;
    }
    if (DYN) deal.finish(j);
}

int launch_fft4096_cf32(const void *in, void *out, size_t nframes, bool inverse, const void *tw4096, void *sched, hipStream_t st)
{
    if (nframes == 0) return PCX_OK;
    const float2 *tab = static_cast<const float2 *>(tw4096);
    // persistent workgroups: LDS (34.8 KB) admits 4 per CU; each walks frames with a grid
    // stride, keeping its twiddles in registers and the next frame in flight
    const unsigned grid = persistent_grid(nframes, (unsigned)PCX_ENV_INT("PCX_FFT_SLOTS", 1024));   // PCX_FFT_SLOTS (diag): oversubscription A/B
    // PCX_FFT_STORE_AUX (A/B): cache-policy bits of the output stores; default 2 = non-temporal
    const int saux = (int)PCX_ENV_INT("PCX_FFT_STORE_AUX", 2);
    const bool dyn = sched && nframes > 2 * 1024 && !PCX_ENV_SET("PCX_SCHED_STATIC");
    const int prio = (int)PCX_ENV_INT("PCX_FFT_PRIO", 0);   // (diag A/B) 1: last pass + stores at wave priority 1; 2: passes 2 and 3
#define PCX_FFT_LAUNCH(INV, SAUX)                                                                                                          \
    do {                                                                                                                                   \
        if (dyn) hipLaunchKernelGGL((fft4096_kernel<INV, SAUX, true>), dim3(1024), dim3(256), 0, st, (const float2 *)in, (float2 *)out, nframes, tab, (SchedState *)sched, prio); \
        else hipLaunchKernelGGL((fft4096_kernel<INV, SAUX>), dim3(grid), dim3(256), 0, st, (const float2 *)in, (float2 *)out, nframes, tab, (SchedState *)nullptr, prio);          \
    } while (0)
    if (inverse) { if (saux == 2) PCX_FFT_LAUNCH(true, 2); else PCX_FFT_LAUNCH(true, 0); }
    else { if (saux == 2) PCX_FFT_LAUNCH(false, 2); else PCX_FFT_LAUNCH(false, 0); }
#undef PCX_FFT_LAUNCH
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}

// --------------------------------------------------------------------------------- //
// generic power-of-two Stockham (radix 4, final radix 2 when log2 N is odd)
// --------------------------------------------------------------------------------- //
template <typename T>
struct C2 {
    T x, y;
};
__device__ __forceinline__ float t_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double t_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <typename T>
__device__ __forceinline__ C2<T> c_add(C2<T> a, C2<T> b) { return {a.x + b.x, a.y + b.y}; }
template <typename T>
__device__ __forceinline__ C2<T> c_sub(C2<T> a, C2<T> b) { return {a.x - b.x, a.y - b.y}; }
// a * w (forward table entry) or a * conj(w)
template <typename T, bool INV>
__device__ __forceinline__ C2<T> c_mul_tw(C2<T> a, C2<T> w)
{
    if (INV) return {t_fma(a.x, w.x, a.y * w.y), t_fma(a.y, w.x, -a.x * w.y)};
    return {t_fma(a.x, w.x, -a.y * w.y), t_fma(a.y, w.x, a.x * w.y)};
}
template <typename T, bool INV>
__device__ __forceinline__ C2<T> c_mul_unit(C2<T> a)
{
    return INV ? C2<T>{-a.y, a.x} : C2<T>{a.y, -a.x};
}

// tw[i] = exp(-j 2 pi i / N), i < N (device, precision T)
template <typename T, bool INV>
__global__ __launch_bounds__(256) void fft_pow2_kernel(const C2<T> *__restrict__ in, C2<T> *__restrict__ out, int N,
                                                       int log2n, size_t nframes, const C2<T> *__restrict__ tw)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    C2<T> *bufA = reinterpret_cast<C2<T> *>(smem_raw);
    C2<T> *bufB = bufA + N;
    const int nt = blockDim.x;
    for (size_t f = blockIdx.x; f < nframes; f += gridDim.x) {
        const C2<T> *src = in + f * (size_t)N;
        C2<T> *dst_final = out + f * (size_t)N;
        C2<T> *ping = bufA, *pong = bufB;
        int Ns = 1, s = 0;
        const int n4 = log2n / 2;  // radix-4 passes
        const bool odd = (log2n & 1) != 0;
        const int npass = n4 + (odd ? 1 : 0);
        __syncthreads();  // previous frame's readers done
        for (int p = 0; p < npass; p++, s++) {
            const bool last = (p == npass - 1);
            const bool radix2 = odd && last;
            const C2<T> *rd = (p == 0) ? src : ping;
            C2<T> *wr = last ? dst_final : pong;
            if (!radix2) {
                const int q = N >> 2;  // butterflies this pass
                const int tstep = N / (Ns * 4);
                for (int j = threadIdx.x; j < q; j += nt) {
                    const int k = j & (Ns - 1);
                    C2<T> a0 = rd[j], a1 = rd[j + q], a2 = rd[j + 2 * q], a3 = rd[j + 3 * q];
                    if (Ns > 1) {
                        a1 = c_mul_tw<T, INV>(a1, tw[k * tstep]);
                        a2 = c_mul_tw<T, INV>(a2, tw[2 * k * tstep]);
                        a3 = c_mul_tw<T, INV>(a3, tw[3 * k * tstep]);
                    }
                    const C2<T> t0 = c_add(a0, a2), t1 = c_sub(a0, a2), t2 = c_add(a1, a3), t3 = c_mul_unit<T, INV>(c_sub(a1, a3));
                    const int j0 = ((j - k) << 2) + k;
                    wr[j0] = c_add(t0, t2);
                    wr[j0 + Ns] = c_add(t1, t3);
                    wr[j0 + 2 * Ns] = c_sub(t0, t2);
                    wr[j0 + 3 * Ns] = c_sub(t1, t3);
                }
                Ns <<= 2;
            } else {
                const int q = N >> 1;
                const int tstep = N / (Ns * 2);
                for (int j = threadIdx.x; j < q; j += nt) {
                    const int k = j & (Ns - 1);
                    C2<T> a0 = rd[j], a1 = rd[j + q];
                    if (Ns > 1) a1 = c_mul_tw<T, INV>(a1, tw[k * tstep]);
                    const int j0 = ((j - k) << 1) + k;
                    wr[j0] = c_add(a0, a1);
                    wr[j0 + Ns] = c_sub(a0, a1);
                }
                Ns <<= 1;
            }
            if (!last) {
                __syncthreads();
                C2<T> *t = ping; ping = pong; pong = t;
            }
        }
    }
}

template <typename T>
static int launch_fft_pow2_t(const void *in, void *out, size_t nbins, size_t nframes, bool inverse, const void *tw, hipStream_t st)
{
    if (nframes == 0) return PCX_OK;
    int log2n = 0;
    while (((size_t)1 << log2n) < nbins) log2n++;
    if (((size_t)1 << log2n) != nbins || nbins < 2) { set_error("fft: %zu is not a power of two >= 2", nbins); return PCX_ERR_UNSUPPORTED; }
    const size_t lds = 2 * nbins * sizeof(C2<T>);
    if (lds > 160 * 1024) { set_error("fft: numBins %zu exceeds the single-workgroup LDS plan", nbins); return PCX_ERR_UNSUPPORTED; }
    unsigned threads = (unsigned)(nbins / 4);
    if (threads < 64) threads = 64;
    if (threads > 256) threads = 256;
    const unsigned grid = (unsigned)(nframes < 4096 ? nframes : 4096);
    auto k = inverse ? fft_pow2_kernel<T, true> : fft_pow2_kernel<T, false>;
    if (lds > 64 * 1024) PCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3(grid), dim3(threads), lds, st, (const C2<T> *)in, (C2<T> *)out, (int)nbins, log2n, nframes, (const C2<T> *)tw);
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}
int launch_fft_pow2_cf32(const void *in, void *out, size_t nbins, size_t nframes, bool inverse, const void *tw, hipStream_t st)
{
    return launch_fft_pow2_t<float>(in, out, nbins, nframes, inverse, tw, st);
}
int launch_fft_pow2_cf64(const void *in, void *out, size_t nbins, size_t nframes, bool inverse, const void *tw, hipStream_t st)
{
    return launch_fft_pow2_t<double>(in, out, nbins, nframes, inverse, tw, st);
}

// --------------------------------------------------------------------------------- //
// complex_int16: kiss_fft -DFIXED_POINT=16, bit-exact
// --------------------------------------------------------------------------------- //
struct K16 {
    int16_t r, i;
};
#define Q15_SROUND(x) ((int16_t)(((x) + (1 << 14)) >> 15))
__device__ __forceinline__ K16 k16_mul(K16 a, K16 b)  // C_MUL, _kiss_fft_guts.h:69-71
{
    K16 m;
    m.r = Q15_SROUND((int32_t)a.r * b.r - (int32_t)a.i * b.i);
    m.i = Q15_SROUND((int32_t)a.r * b.i + (int32_t)a.i * b.r);
    return m;
}
__device__ __forceinline__ K16 k16_fixdiv(K16 c, int mult)  // C_FIXDIV: mult = 32767/radix
{
    K16 m;
    m.r = Q15_SROUND((int32_t)c.r * mult);
    m.i = Q15_SROUND((int32_t)c.i * mult);
    return m;
}
__device__ __forceinline__ K16 k16_add(K16 a, K16 b) { return {(int16_t)(a.r + b.r), (int16_t)(a.i + b.i)}; }
__device__ __forceinline__ K16 k16_sub(K16 a, K16 b) { return {(int16_t)(a.r - b.r), (int16_t)(a.i - b.i)}; }

// numBins = 1: kf_factor yields the single factor 1 and kf_work runs kf_bfly_generic with p = 1
// (kiss_fft.c:202-235,300), whose C_FIXDIV(scratch[0], 1) multiplies by 32767/32768 with rounding --
// the one-bin fixed-point "transform" is not the identity.
__global__ __launch_bounds__(256) void fft_q15_one_kernel(const K16 *__restrict__ in, K16 *__restrict__ out, size_t n)
{
    const size_t gstride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gstride) out[i] = k16_fixdiv(in[i], 32767);
}
int launch_fft_q15_one(const void *in, void *out, size_t nframes, hipStream_t st)
{
    if (nframes == 0) return PCX_OK;
    hipLaunchKernelGGL(fft_q15_one_kernel, dim3(stream_grid(nframes, 256)), dim3(256), 0, st, (const K16 *)in, (K16 *)out, nframes);
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}

constexpr int kMaxStages = 16;
struct Q15Plan {
    int nstages;
    int radix[kMaxStages];  // top (stage 0) .. bottom, as kf_factor emits them (kiss_fft.c:309-328)
};

// kf_work's recursion (kiss_fft.c:237-302) unrolled: a digit-reversing gather (the
// m==1 leaves, :276-280) followed by the butterfly passes bottom-up.  Butterflies of
// one pass are independent, so running them in parallel leaves every rounding as is.
// Every radix is 2 or 4, so m and fstride are powers of two (shifts and masks, no division); the
// gather order is a host-made table (`perm`, numBins entries) and the Q15 twiddles sit in LDS next
// to the frame.  FPW frames share a workgroup when numBins is small.
__global__ __launch_bounds__(256) void fft_q15_kernel(const K16 *__restrict__ in, K16 *__restrict__ out, int N, int log2N,
                                                      size_t nframes, const K16 *__restrict__ tw, const unsigned short *__restrict__ perm,
                                                      Q15Plan plan, int inverse, int fpw, int stage_tw)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    K16 *twstage = reinterpret_cast<K16 *>(smem_raw);      // N twiddles (when they fit beside the frames)
    K16 *bufs = twstage + (stage_tw ? N : 0);              // fpw frames
    const K16 *twl = stage_tw ? twstage : tw;
    const int nt = blockDim.x / fpw;                       // lanes per frame
    const int fl = threadIdx.x / nt, t = threadIdx.x % nt; // frame slot, lane within the frame
    K16 *buf = bufs + (size_t)fl * N;
    if (stage_tw)
        for (int i = threadIdx.x; i < N; i += blockDim.x) twstage[i] = tw[i];
    const size_t ngroups = (nframes + fpw - 1) / fpw;
    for (size_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const size_t f = g * fpw + fl;
        const bool live = f < nframes;
        const K16 *src = in + f * (size_t)N;
        __syncthreads();
        if (live)
            for (int pos = t; pos < N; pos += nt) buf[pos] = src[perm[pos]];
        __syncthreads();
        // butterfly passes, bottom (m = 1) to top (m = N/p0)
        int lm = 0;   // log2(m)
        for (int s = plan.nstages - 1; s >= 0; s--) {
            const int p = plan.radix[s];
            const int lp = p == 4 ? 2 : 1;
            const int m = 1 << lm;
            const int lfs = log2N - lp - lm;   // log2(fstride), fstride = N / (p*m)
            const int nb = N >> lp;            // butterflies this pass
            if (live)
            for (int b = t; b < nb; b += nt) {
                const int k = b & (m - 1), gq = b >> lm;
                K16 *F = buf + (gq << (lp + lm)) + k;
                if (p == 4) {  // kf_bfly4, kiss_fft.c:44-90
                    K16 f0 = k16_fixdiv(F[0], 8191), f1 = k16_fixdiv(F[m], 8191), f2 = k16_fixdiv(F[2 * m], 8191), f3 = k16_fixdiv(F[3 * m], 8191);
                    const K16 s0 = k16_mul(f1, twl[k << lfs]);
                    const K16 s1 = k16_mul(f2, twl[(k << lfs) * 2]);
                    const K16 s2 = k16_mul(f3, twl[(k << lfs) * 3]);
                    const K16 s5 = k16_sub(f0, s1);
                    f0 = k16_add(f0, s1);
                    const K16 s3 = k16_add(s0, s2);
                    const K16 s4 = k16_sub(s0, s2);
                    F[2 * m] = k16_sub(f0, s3);
                    F[0] = k16_add(f0, s3);
                    if (inverse) {
                        F[m] = {(int16_t)(s5.r - s4.i), (int16_t)(s5.i + s4.r)};
                        F[3 * m] = {(int16_t)(s5.r + s4.i), (int16_t)(s5.i - s4.r)};
                    } else {
                        F[m] = {(int16_t)(s5.r + s4.i), (int16_t)(s5.i - s4.r)};
                        F[3 * m] = {(int16_t)(s5.r - s4.i), (int16_t)(s5.i + s4.r)};
                    }
                } else {  // p == 2: kf_bfly2, kiss_fft.c:21-42
                    const K16 f0 = k16_fixdiv(F[0], 16383), f1 = k16_fixdiv(F[m], 16383);
                    const K16 tt = k16_mul(f1, twl[k << lfs]);
                    F[m] = k16_sub(f0, tt);
                    F[0] = k16_add(f0, tt);
                }
            }
            __syncthreads();
            lm += lp;
        }
        if (live) {
            K16 *dst = out + f * (size_t)N;
            for (int i = t; i < N; i += nt) dst[i] = buf[i];
        }
    }
}

// --------------------------------------------------------------------------------- //
// complex_int16, numBins = 16 ... 16384: the same passes with TWO radix-4 stages fused per
// trip through LDS.  A lane holds 16 elements of a frame in registers; for the pair of stages with
// butterfly strides m and 4m (m = 16^p) those are positions G*16m + k + j*m, j = 0..15: stage m
// runs its four butterflies on j = 4a + {0,1,2,3}, stage 4m on j = b + {0,4,8,12} -- every operand
// of both stages is in the lane, and each butterfly is kf_bfly4 (kiss_fft.c:44-90) with its own
// FIXDIV / sround sequence, so the result is the reference's bit for bit.  When log4(numBins) is
// odd one single radix-4 pass follows (stride numBins/4, elements t + j*T).  The digit-reversing
// leaf gather goes straight from global memory into the registers of the first pair, the last
// pair's results are stored straight to global memory (lane-contiguous): 2 LDS round trips for
// 4096 bins instead of 6 passes plus a gather.  LDS image padded i + i/16.
// --------------------------------------------------------------------------------- //
__device__ __forceinline__ void k16_bfly4(K16 &f0, K16 &f1, K16 &f2, K16 &f3, K16 w1, K16 w2, K16 w3, int inverse)
{
    f0 = k16_fixdiv(f0, 8191); f1 = k16_fixdiv(f1, 8191); f2 = k16_fixdiv(f2, 8191); f3 = k16_fixdiv(f3, 8191);
    const K16 s0 = k16_mul(f1, w1);
    const K16 s1 = k16_mul(f2, w2);
    const K16 s2 = k16_mul(f3, w3);
    const K16 s5 = k16_sub(f0, s1);
    f0 = k16_add(f0, s1);
    const K16 s3 = k16_add(s0, s2);
    const K16 s4 = k16_sub(s0, s2);
    f2 = k16_sub(f0, s3);
    f0 = k16_add(f0, s3);
    if (inverse) {
        f1 = {(int16_t)(s5.r - s4.i), (int16_t)(s5.i + s4.r)};
        f3 = {(int16_t)(s5.r + s4.i), (int16_t)(s5.i - s4.r)};
    } else {
        f1 = {(int16_t)(s5.r + s4.i), (int16_t)(s5.i - s4.r)};
        f3 = {(int16_t)(s5.r - s4.i), (int16_t)(s5.i + s4.r)};
    }
}

// Stage schedule for numBins = 2^LOG2N, 16 <= numBins <= 16384: kf_factor gives radix 4 as often as
// possible and one radix-2 stage at the BOTTOM when LOG2N is odd (kiss_fft.c:309-328).  Pass 0 works
// on the 16 consecutive leaf positions of a lane: two radix-4 stages (m = 1, 4), or -- LEAD2 -- the
// radix-2 stage plus the m = 2 radix-4 stage (two groups of 8).  Every later pass pairs two radix-4
// stages with strides m, 4m, m = 2^LM(p); an odd stage left over is the single TAIL pass.
template <int LOG2N>
struct Q15R16 {
    static constexpr int N = 1 << LOG2N;
    static constexpr int T = N / 16;                      // lanes per frame
    static constexpr int THREADS = T < 256 ? 256 : T;
    static constexpr int FPW = THREADS / T;               // frames per workgroup
    static constexpr bool LEAD2 = (LOG2N % 2) == 1;
    static constexpr int R4 = LOG2N / 2;                  // radix-4 stages
    static constexpr int REST = R4 - (LEAD2 ? 1 : 2);     // radix-4 stages after pass 0
    static constexpr int NF = 1 + REST / 2;               // passes that run in registers: pass 0 + pairs
    static constexpr bool TAIL = (REST % 2) == 1;         // one more single radix-4 pass
    static constexpr int LM(int p) { return p == 0 ? 0 : (LEAD2 ? 4 * p - 1 : 4 * p); }   // log2 stride of pass p
    static constexpr int IMG = N + N / 16;                // padded frame image (elements)
    static constexpr size_t LDS = ((size_t)N + (size_t)FPW * IMG) * sizeof(K16);
};

template <int LOG2N>
__global__ __launch_bounds__(Q15R16<LOG2N>::THREADS) void fft_q15_r16_kernel(const K16 *__restrict__ in, K16 *__restrict__ out, size_t nframes,
                                                                           const K16 *__restrict__ tw, const unsigned short *__restrict__ perm,
                                                                           int inverse)
{
    typedef Q15R16<LOG2N> P;
    constexpr int N = P::N, T = P::T, FPW = P::FPW, NF = P::NF;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    K16 *twl = reinterpret_cast<K16 *>(smem_raw);
    const int fl = threadIdx.x / T, t = threadIdx.x % T;
    K16 *img = twl + N + (size_t)fl * P::IMG;
    for (int i = threadIdx.x; i < N; i += P::THREADS) twl[i] = tw[i];
    // the lane's 16 leaf positions 16t .. 16t+15 and where they come from
    unsigned short src_idx[16];
#pragma unroll
    for (int j = 0; j < 16; j++) src_idx[j] = perm[16 * t + j];
    __syncthreads();
    // positions of pass p: base(p) + (j << LM(p))
    auto base_of = [&](int lm) { return ((t >> lm) << (lm + 4)) + (t & ((1 << lm) - 1)); };
    const size_t ngroups = (nframes + FPW - 1) / FPW;
    for (size_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const size_t f = g * FPW + fl;
        const bool live = f < nframes;
        const K16 *src = in + (live ? f : 0) * (size_t)N;
        K16 v[16];
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = src[src_idx[j]];
        // ---- pass 0 on the leaf positions ----
        if (P::LEAD2) {
            // kf_bfly2 (kiss_fft.c:21-42), m = 1, twiddle tw[0]
            const K16 w0 = twl[0];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const K16 f0 = k16_fixdiv(v[2 * i], 16383), f1 = k16_fixdiv(v[2 * i + 1], 16383);
                const K16 tt = k16_mul(f1, w0);
                v[2 * i + 1] = k16_sub(f0, tt);
                v[2 * i] = k16_add(f0, tt);
            }
            // radix-4 stage m = 2: butterflies (g, k in {0,1}), fstride = N / 8
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int ti = k << (LOG2N - 3);
                const K16 w1 = twl[ti], w2 = twl[2 * ti], w3 = twl[3 * ti];
#pragma unroll
                for (int h = 0; h < 2; h++) k16_bfly4(v[8 * h + k], v[8 * h + k + 2], v[8 * h + k + 4], v[8 * h + k + 6], w1, w2, w3, inverse);
            }
        } else {
            {   // stage m = 1: twiddle index 0 for every butterfly (tw[0] = 32767: NOT an identity in Q15)
                const K16 w0 = twl[0];
#pragma unroll
                for (int a = 0; a < 4; a++) k16_bfly4(v[4 * a], v[4 * a + 1], v[4 * a + 2], v[4 * a + 3], w0, w0, w0, inverse);
            }
#pragma unroll
            for (int b = 0; b < 4; b++) {   // stage m = 4: k = b, fstride = N / 16
                const int ti = b << (LOG2N - 4);
                k16_bfly4(v[b], v[b + 4], v[b + 8], v[b + 12], twl[ti], twl[2 * ti], twl[3 * ti], inverse);
            }
        }
        // ---- pairs of radix-4 stages, strides m and 4m ----
#pragma unroll
        for (int p = 1; p < NF; p++) {
            const int lm = P::LM(p), m = 1 << lm, lmp = P::LM(p - 1);
            const int k = t & (m - 1);
            const int bp = base_of(lmp), b0 = base_of(lm);
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 16; j++) { const int pos = bp + (j << lmp); img[pos + (pos >> 4)] = v[j]; }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 16; j++) { const int pos = b0 + (j << lm); v[j] = img[pos + (pos >> 4)]; }
            {   // stage m: butterflies (g = 4G + a, k), fstride = N / (4m)
                const int ti = k << (LOG2N - lm - 2);
                const K16 w1 = twl[ti], w2 = twl[2 * ti], w3 = twl[3 * ti];
#pragma unroll
                for (int a = 0; a < 4; a++) k16_bfly4(v[4 * a], v[4 * a + 1], v[4 * a + 2], v[4 * a + 3], w1, w2, w3, inverse);
            }
#pragma unroll
            for (int b = 0; b < 4; b++) {   // stage 4m: butterflies (g = G, k + b*m), fstride = N / (16m)
                const int ti = (k + b * m) << (LOG2N - lm - 4);
                k16_bfly4(v[b], v[b + 4], v[b + 8], v[b + 12], twl[ti], twl[2 * ti], twl[3 * ti], inverse);
            }
        }
        constexpr int lml = P::LM(NF - 1);   // stride of the last register pass
        if (P::TAIL) {
            // last pass's positions -> t + j*T; butterflies k = t + b*T of the stride-N/4 stage (fstride 1)
            const int bp = base_of(lml);
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 16; j++) { const int pos = bp + (j << lml); img[pos + (pos >> 4)] = v[j]; }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 16; j++) { const int pos = t + j * T; v[j] = img[pos + (pos >> 4)]; }
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int ti = t + b * T;
                k16_bfly4(v[b], v[b + 4], v[b + 8], v[b + 12], twl[ti], twl[2 * ti], twl[3 * ti], inverse);
            }
        }
        if (live) {
            K16 *dst = out + f * (size_t)N;
            if (P::TAIL) {
#pragma unroll
                for (int j = 0; j < 16; j++) dst[t + j * T] = v[j];
            } else {
                const int b0 = base_of(lml);
#pragma unroll
                for (int j = 0; j < 16; j++) dst[b0 + (j << lml)] = v[j];
            }
        }
    }
}

template <int LOG2N>
static int launch_q15_r16(const void *in, void *out, size_t nframes, bool inverse, const void *tw, const void *perm, hipStream_t st)
{
    typedef Q15R16<LOG2N> P;
    auto k = fft_q15_r16_kernel<LOG2N>;
    if (P::LDS > 64 * 1024) PCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)P::LDS));
    const size_t ngroups = (nframes + P::FPW - 1) / P::FPW;
    unsigned per_cu = (unsigned)(160 * 1024 / P::LDS);
    const unsigned by_threads = 2048 / P::THREADS;
    if (per_cu > by_threads) per_cu = by_threads;
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    hipLaunchKernelGGL(k, dim3(persistent_grid(ngroups, 256 * per_cu, 4)), dim3(P::THREADS), P::LDS, st, (const K16 *)in, (K16 *)out, nframes,
                       (const K16 *)tw, (const unsigned short *)perm, inverse ? 1 : 0);
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}

int launch_fft_q15(const void *in, void *out, size_t nbins, size_t nframes, bool inverse, const void *tw, const void *perm,
                   const int *radix_host, int nstages, hipStream_t st)
{
    if (nframes == 0) return PCX_OK;
    if (nstages > kMaxStages) { set_error("fft(int16): too many stages"); return PCX_ERR_UNSUPPORTED; }
    Q15Plan plan;
    plan.nstages = nstages;
    for (int s = 0; s < nstages; s++) {
        if (radix_host[s] != 4 && radix_host[s] != 2) {
            set_error("fft(int16): radix-%d stage (numBins %zu) is not implemented on the device", radix_host[s], nbins);
            return PCX_ERR_UNSUPPORTED;
        }
        plan.radix[s] = radix_host[s];
    }
    int log2n = 0;
    while (((size_t)1 << log2n) < nbins) log2n++;
    // 16 <= numBins <= 16384 take the fused-pair kernel; PCX_FFT_Q15_PASSES=1 keeps the pass kernel (A/B)
    const int passes_only = (int)PCX_ENV_INT("PCX_FFT_Q15_PASSES", 0);
    if (!passes_only) {
        switch (log2n) {
        case 4: return launch_q15_r16<4>(in, out, nframes, inverse, tw, perm, st);
        case 5: return launch_q15_r16<5>(in, out, nframes, inverse, tw, perm, st);
        case 6: return launch_q15_r16<6>(in, out, nframes, inverse, tw, perm, st);
        case 7: return launch_q15_r16<7>(in, out, nframes, inverse, tw, perm, st);
        case 8: return launch_q15_r16<8>(in, out, nframes, inverse, tw, perm, st);
        case 9: return launch_q15_r16<9>(in, out, nframes, inverse, tw, perm, st);
        case 10: return launch_q15_r16<10>(in, out, nframes, inverse, tw, perm, st);
        case 11: return launch_q15_r16<11>(in, out, nframes, inverse, tw, perm, st);
        case 12: return launch_q15_r16<12>(in, out, nframes, inverse, tw, perm, st);
        case 13: return launch_q15_r16<13>(in, out, nframes, inverse, tw, perm, st);
        case 14: return launch_q15_r16<14>(in, out, nframes, inverse, tw, perm, st);
        }
    }
    // lanes per frame: one per radix-4 butterfly, at least 16; frames per workgroup fill 256 lanes
    unsigned lanes = (unsigned)(nbins / 4);
    if (lanes < 16) lanes = 16;
    if (lanes > 256) lanes = 256;
    const unsigned fpw = 256 / lanes;
    size_t lds = nbins * sizeof(K16) * (1 + fpw);            // twiddles + fpw frames
    const int stage_tw = lds <= 160 * 1024;
    if (!stage_tw) lds = nbins * sizeof(K16) * fpw;          // largest sizes: twiddles stay in L2
    if (lds > 160 * 1024) { set_error("fft(int16): numBins %zu exceeds LDS", nbins); return PCX_ERR_UNSUPPORTED; }
    if (lds > 64 * 1024) PCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fft_q15_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const size_t ngroups = (nframes + fpw - 1) / fpw;
    unsigned per_cu = (unsigned)(160 * 1024 / lds);
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    // (PCX_Q15_ROUNDS, diagnostic library: groups per workgroup instead of the fixed factor, A/B)
    const long q15_rounds = PCX_ENV_INT("PCX_Q15_ROUNDS", 0);
    const unsigned grid = q15_rounds > 0 ? rounds_grid(ngroups, 256 * per_cu, (unsigned)q15_rounds)
                                         : persistent_grid(ngroups, 256 * per_cu, 4);   // four queued per slot: +5..9 % on the Q15 kernels (tools/sweep_fft.py, PCX_OVERSUB A/B)
    hipLaunchKernelGGL(fft_q15_kernel, dim3(grid), dim3(lanes * fpw), lds, st, (const K16 *)in, (K16 *)out, (int)nbins, log2n, nframes,
                       (const K16 *)tw, (const unsigned short *)perm, plan, inverse ? 1 : 0, (int)fpw, stage_tw);
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}

}  // namespace pcx
