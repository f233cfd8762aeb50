// fft_mixed.hip -- /comms/fft for ANY numBins: the mixed-radix decimation-in-time plan of
// kissfft (factor out 4s, then 2s, then 3, 5, 7, ...: fft/kissfft.hh:38-55, fft/kiss_fft.c:309-328),
// one frame per workgroup in LDS.
//
// kf_work's recursion (kissfft.hh:87-120 / kiss_fft.c:237-302) is a digit-reversing gather
// (the m == 1 leaves) followed by the butterfly passes bottom-up; the butterflies of one pass
// are independent, so they run one per lane.  Radix 2/3/4/5 passes work in place
// (kf_bfly2/3/4/5); any other prime radix p uses the O(p^2) kf_bfly_generic, computed one
// OUTPUT per lane out of place between two LDS images (so a prime numBins still spreads over
// the workgroup).
//
//   Arith = F32 / F64 : same butterfly algebra as kissfft.hh in fp32 / fp64 with FMA-free
//                       complex products; twiddles from a double-precision host table.
//                       Parity bar: 1e-5 of max|X| (fp32), 1e-13 (fp64).
//   Arith = Q15       : fft/kiss_fft.c with -DFIXED_POINT=16 -- C_FIXDIV by the radix, sround,
//                       HALF_OF, S_MUL exactly as _kiss_fft_guts.h:44-83: BIT-EXACT.
//
// This is the coverage path (sizes the radix-16 and power-of-two kernels do not take); it is
// correctness-first and sized by LDS: numBins * sizeof(complex) * 2 <= 160 KB.
#include "pcx_internal.hpp"

namespace pcx {

namespace {

constexpr int kMaxStagesMixed = 24;
struct MixedPlan {
    int nstages;
    int radix[kMaxStagesMixed];  // top (stage 0) .. bottom, as kf_factor emits them
    int needs_pingpong;          // any generic-radix pass
};

template <typename T>
struct Cx {
    T r, i;
};

// ---------------- floating point arithmetic (kissfft.hh) ----------------
template <typename T>
struct FloatArith {
    typedef Cx<T> cpx;
    typedef T scalar;
    static __device__ __forceinline__ cpx mul(cpx a, cpx b) { return {a.r * b.r - a.i * b.i, a.r * b.i + a.i * b.r}; }
    static __device__ __forceinline__ cpx add(cpx a, cpx b) { return {a.r + b.r, a.i + b.i}; }
    static __device__ __forceinline__ cpx sub(cpx a, cpx b) { return {a.r - b.r, a.i - b.i}; }
    static __device__ __forceinline__ cpx fixdiv(cpx c, int) { return c; }          // C_FIXDIV: no-op for floats
    static __device__ __forceinline__ scalar smul(scalar a, scalar b) { return a * b; }
    static __device__ __forceinline__ scalar half(scalar a) { return a * (T)0.5; }
    static __device__ __forceinline__ scalar sadd(scalar a, scalar b) { return a + b; }
    static __device__ __forceinline__ scalar ssub(scalar a, scalar b) { return a - b; }
    static __device__ __forceinline__ scalar neg(scalar a) { return -a; }
};
// ---------------- Q15 arithmetic (kiss_fft.c, FIXED_POINT=16) ----------------
struct Q15Arith {
    typedef Cx<int16_t> cpx;
    typedef int16_t scalar;
    static __device__ __forceinline__ int16_t sround(int32_t x) { return (int16_t)((x + (1 << 14)) >> 15); }
    static __device__ __forceinline__ cpx mul(cpx a, cpx b)
    {
        return {sround((int32_t)a.r * b.r - (int32_t)a.i * b.i), sround((int32_t)a.r * b.i + (int32_t)a.i * b.r)};
    }
    static __device__ __forceinline__ cpx add(cpx a, cpx b) { return {(int16_t)(a.r + b.r), (int16_t)(a.i + b.i)}; }
    static __device__ __forceinline__ cpx sub(cpx a, cpx b) { return {(int16_t)(a.r - b.r), (int16_t)(a.i - b.i)}; }
    static __device__ __forceinline__ cpx fixdiv(cpx c, int p)
    {
        const int32_t k = 32767 / p;
        return {sround((int32_t)c.r * k), sround((int32_t)c.i * k)};
    }
    static __device__ __forceinline__ scalar smul(scalar a, scalar b) { return sround((int32_t)a * b); }
    static __device__ __forceinline__ scalar half(scalar a) { return (int16_t)(a >> 1); }
    static __device__ __forceinline__ scalar sadd(scalar a, scalar b) { return (int16_t)(a + b); }
    static __device__ __forceinline__ scalar ssub(scalar a, scalar b) { return (int16_t)(a - b); }
    static __device__ __forceinline__ scalar neg(scalar a) { return (int16_t)(-a); }
};

template <typename A>
__global__ __launch_bounds__(256) void fft_mixed_kernel(const typename A::cpx *__restrict__ in, typename A::cpx *__restrict__ out,
                                                        int N, size_t nframes, const typename A::cpx *__restrict__ tw,
                                                        MixedPlan plan, int inverse)
{
    typedef typename A::cpx cpx;
    typedef typename A::scalar scalar;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cpx *cur = reinterpret_cast<cpx *>(smem_raw);
    cpx *alt = cur + N;
    const int nt = blockDim.x;
    for (size_t f = blockIdx.x; f < nframes; f += gridDim.x) {
        const cpx *src = in + f * (size_t)N;
        __syncthreads();
        // leaves: position sum_s q_s*m_s  <-  input index sum_s q_s*fstride_s
        for (int pos = threadIdx.x; pos < N; pos += nt) {
            int rem = pos, m = N, fstride = 1, idx = 0;
            for (int s = 0; s < plan.nstages; s++) {
                const int p = plan.radix[s];
                m /= p;
                const int q = rem / m;
                rem -= q * m;
                idx += q * fstride;
                fstride *= p;
            }
            cur[pos] = src[idx];
        }
        __syncthreads();
        int m = 1;
        for (int s = plan.nstages - 1; s >= 0; s--) {
            const int p = plan.radix[s];
            const int fstride = N / (p * m);
            if (p == 2 || p == 3 || p == 4 || p == 5) {
                const int nb = N / p;
                for (int b = threadIdx.x; b < nb; b += nt) {
                    const int k = b % m, g = b / m;
                    cpx *F = cur + g * (p * m) + k;
                    if (p == 2) {  // kf_bfly2
                        const cpx f0 = A::fixdiv(F[0], 2), f1 = A::fixdiv(F[m], 2);
                        const cpx t = A::mul(f1, tw[k * fstride]);
                        F[m] = A::sub(f0, t);
                        F[0] = A::add(f0, t);
                    } else if (p == 4) {  // kf_bfly4
                        cpx f0 = A::fixdiv(F[0], 4);
                        const cpx f1 = A::fixdiv(F[m], 4), f2 = A::fixdiv(F[2 * m], 4), f3 = A::fixdiv(F[3 * m], 4);
                        const cpx s0 = A::mul(f1, tw[k * fstride]), s1 = A::mul(f2, tw[k * fstride * 2]), s2 = A::mul(f3, tw[k * fstride * 3]);
                        const cpx s5 = A::sub(f0, s1);
                        f0 = A::add(f0, s1);
                        const cpx s3 = A::add(s0, s2), s4 = A::sub(s0, s2);
                        F[2 * m] = A::sub(f0, s3);
                        F[0] = A::add(f0, s3);
                        if (inverse) {
                            F[m] = {A::ssub(s5.r, s4.i), A::sadd(s5.i, s4.r)};
                            F[3 * m] = {A::sadd(s5.r, s4.i), A::ssub(s5.i, s4.r)};
                        } else {
                            F[m] = {A::sadd(s5.r, s4.i), A::ssub(s5.i, s4.r)};
                            F[3 * m] = {A::ssub(s5.r, s4.i), A::sadd(s5.i, s4.r)};
                        }
                    } else if (p == 3) {  // kf_bfly3
                        const cpx epi3 = tw[fstride * m];
                        const cpx f0 = A::fixdiv(F[0], 3), f1 = A::fixdiv(F[m], 3), f2 = A::fixdiv(F[2 * m], 3);
                        const cpx s1 = A::mul(f1, tw[k * fstride]), s2 = A::mul(f2, tw[k * fstride * 2]);
                        const cpx s3 = A::add(s1, s2);
                        cpx s0 = A::sub(s1, s2);
                        cpx fm = {A::ssub(f0.r, A::half(s3.r)), A::ssub(f0.i, A::half(s3.i))};
                        s0 = {A::smul(s0.r, epi3.i), A::smul(s0.i, epi3.i)};
                        F[0] = A::add(f0, s3);
                        F[2 * m] = {A::sadd(fm.r, s0.i), A::ssub(fm.i, s0.r)};
                        F[m] = {A::ssub(fm.r, s0.i), A::sadd(fm.i, s0.r)};
                    } else {  // kf_bfly5
                        const cpx ya = tw[fstride * m], yb = tw[fstride * 2 * m];
                        cpx f0 = A::fixdiv(F[0], 5);
                        const cpx f1 = A::fixdiv(F[m], 5), f2 = A::fixdiv(F[2 * m], 5), f3 = A::fixdiv(F[3 * m], 5), f4 = A::fixdiv(F[4 * m], 5);
                        const cpx s0 = f0;
                        const cpx s1 = A::mul(f1, tw[k * fstride]), s2 = A::mul(f2, tw[2 * k * fstride]);
                        const cpx s3 = A::mul(f3, tw[3 * k * fstride]), s4 = A::mul(f4, tw[4 * k * fstride]);
                        const cpx s7 = A::add(s1, s4), s10 = A::sub(s1, s4), s8 = A::add(s2, s3), s9 = A::sub(s2, s3);
                        // Fout0 += s7 + s8 (kiss_fft.c:174-175 adds the sum; kissfft.hh:228-229 adds twice: same for floats up to rounding)
                        f0 = {A::sadd(f0.r, A::sadd(s7.r, s8.r)), A::sadd(f0.i, A::sadd(s7.i, s8.i))};
                        F[0] = f0;
                        const cpx s5 = {A::sadd(A::sadd(s0.r, A::smul(s7.r, ya.r)), A::smul(s8.r, yb.r)),
                                        A::sadd(A::sadd(s0.i, A::smul(s7.i, ya.r)), A::smul(s8.i, yb.r))};
                        const cpx s6 = {A::sadd(A::smul(s10.i, ya.i), A::smul(s9.i, yb.i)),
                                        A::ssub(A::neg(A::smul(s10.r, ya.i)), A::smul(s9.r, yb.i))};
                        F[m] = A::sub(s5, s6);
                        F[4 * m] = A::add(s5, s6);
                        const cpx s11 = {A::sadd(A::sadd(s0.r, A::smul(s7.r, yb.r)), A::smul(s8.r, ya.r)),
                                         A::sadd(A::sadd(s0.i, A::smul(s7.i, yb.r)), A::smul(s8.i, ya.r))};
                        const cpx s12 = {A::sadd(A::neg(A::smul(s10.i, yb.i)), A::smul(s9.i, ya.i)),
                                         A::ssub(A::smul(s10.r, yb.i), A::smul(s9.r, ya.i))};
                        F[2 * m] = A::add(s11, s12);
                        F[3 * m] = A::sub(s11, s12);
                    }
                }
                __syncthreads();
            } else {
                // kf_bfly_generic, one output element per lane, cur -> alt
                const int span = p * m;
                for (int e = threadIdx.x; e < N; e += nt) {
                    const int g = e / span, within = e - g * span;   // within = u + q1*m
                    const int u = within % m;
                    const cpx *S = cur + g * span + u;               // scratch[q] = fixdiv(S[q*m])
                    const int k = within;                            // the reference's running k = u + q1*m
                    cpx acc = A::fixdiv(S[0], p);
                    int twidx = 0;
                    for (int q = 1; q < p; q++) {
                        twidx += fstride * k;
                        if (twidx >= N) twidx -= N;
                        acc = A::add(acc, A::mul(A::fixdiv(S[q * m], p), tw[twidx]));
                    }
                    alt[e] = acc;
                }
                __syncthreads();
                cpx *t = cur; cur = alt; alt = t;
            }
            m *= p;
        }
        cpx *dst = out + f * (size_t)N;
        for (int i = threadIdx.x; i < N; i += nt) dst[i] = cur[i];
        (void)sizeof(scalar);
    }
}

template <typename A>
int launch_mixed(const void *in, void *out, size_t nbins, size_t nframes, bool inverse, const void *tw, const int *radix, int nstages, hipStream_t st)
{
    if (nframes == 0) return PCX_OK;
    if (nstages > kMaxStagesMixed) { set_error("fft: too many stages for numBins %zu", nbins); return PCX_ERR_UNSUPPORTED; }
    MixedPlan plan;
    plan.nstages = nstages;
    plan.needs_pingpong = 0;
    for (int s = 0; s < nstages; s++) {
        plan.radix[s] = radix[s];
        if (radix[s] != 2 && radix[s] != 3 && radix[s] != 4 && radix[s] != 5) plan.needs_pingpong = 1;
    }
    const size_t lds = nbins * sizeof(typename A::cpx) * 2;
    if (lds > 160 * 1024) {
        set_error("fft: numBins %zu does not fit the single-workgroup LDS plan (%zu bytes)", nbins, lds);
        return PCX_ERR_UNSUPPORTED;
    }
    auto k = fft_mixed_kernel<A>;
    if (lds > 64 * 1024) PCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    unsigned threads = (unsigned)((nbins + 3) / 4);
    threads = (threads + 63) / 64 * 64;
    if (threads < 64) threads = 64;
    if (threads > 256) threads = 256;
    const unsigned grid = (unsigned)(nframes < 4096 ? nframes : 4096);
    hipLaunchKernelGGL(k, dim3(grid), dim3(threads), lds, st, (const typename A::cpx *)in, (typename A::cpx *)out, (int)nbins, nframes,
                       (const typename A::cpx *)tw, plan, inverse ? 1 : 0);
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}

}  // namespace

int launch_fft_mixed(int scalar, const void *in, void *out, size_t nbins, size_t nframes, bool inverse, const void *tw,
                     const int *radix_host, int nstages, hipStream_t st)
{
    switch (scalar) {
    case PCX_F32: return launch_mixed<FloatArith<float>>(in, out, nbins, nframes, inverse, tw, radix_host, nstages, st);
    case PCX_F64: return launch_mixed<FloatArith<double>>(in, out, nbins, nframes, inverse, tw, radix_host, nstages, st);
    case PCX_I16: return launch_mixed<Q15Arith>(in, out, nbins, nframes, inverse, tw, radix_host, nstages, st);
    }
    set_error("fft: unsupported scalar %d", scalar);
    return PCX_ERR_ARG;
}

}  // namespace pcx
