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
#include <cstdlib>
#include "fft4096.hpp"
#include "fft_f64.hpp"
#include "pcx_internal.hpp"

namespace pcx {

namespace {

constexpr int kMaxStagesMixed = 24;
// Per-stage geometry is resolved on the host: the kernel never divides by a run-time value, and every index product
// is a 24-bit multiply (full rate; a 32-bit v_mul_lo/hi is quarter rate and the butterfly passes are VALU-bound).
// q = floor(n / d) is (int)((n + 0.5f) * fl(1/d)): the exact (n + 0.5)/d sits at least 0.5/d from an integer and the
// float product is off by at most (n + 0.5)/d * 2^-23, so the truncation is exact while n + 0.5 < 2^22 (here n <= 20480).
struct MixedPlan {
    int nstages;
    int radix[kMaxStagesMixed];        // top (stage 0) .. bottom, as kf_factor emits them
    int m[kMaxStagesMixed];            // butterfly span below stage s (product of the radices after it)
    int fstride[kMaxStagesMixed];      // N / (radix * m)
    int nb[kMaxStagesMixed];           // N / radix: butterflies per frame and pass
    float inv_m[kMaxStagesMixed], inv_nb[kMaxStagesMixed], inv_span[kMaxStagesMixed];   // 1/m, 1/(N/radix), 1/(radix*m)
    float inv_n;                       // 1/N
    int fpw;                           // frames per workgroup
    int needs_pingpong;                // any generic-radix pass
};
__device__ __forceinline__ int fdiv(int n, float inv) { return (int)(((float)n + 0.5f) * inv); }
__device__ __forceinline__ int mul24(int a, int b) { return __mul24(a, b); }

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

// TWLDS: the twiddle table (N entries, gathered per lane at k * fstride * j) sits in LDS behind the frame images,
// copied once per persistent workgroup; otherwise it is read through L1/L2
// kf_bfly2 / kf_bfly3 / kf_bfly4 / kf_bfly5 (kiss_fft.c:21-185, kissfft.hh:163-262) on the p elements F[0], F[m] .. F[(p-1) m] with the
// twiddles W^(k q) = tw[kf q], kf = k * fstride: every C_FIXDIV / sround of the Q15 build in the reference's own order (Q15Arith), plain
// arithmetic for floats.  Shared by the LDS kernel below and the stage-per-launch plan for frames beyond one workgroup's LDS.
template <typename A>
__device__ __forceinline__ void kf_bfly(int p, typename A::cpx *F, int m, const typename A::cpx *tw, int kf, int fstride, int inverse)
{
    typedef typename A::cpx cpx;
    if (p == 2) {  // kf_bfly2
        const cpx f0 = A::fixdiv(F[0], 2), f1 = A::fixdiv(F[m], 2);
        const cpx t = A::mul(f1, tw[kf]);
        F[m] = A::sub(f0, t);
        F[0] = A::add(f0, t);
    } else if (p == 4) {  // kf_bfly4
        cpx f0 = A::fixdiv(F[0], 4);
        const cpx f1 = A::fixdiv(F[m], 4), f2 = A::fixdiv(F[2 * m], 4), f3 = A::fixdiv(F[3 * m], 4);
        const cpx s0 = A::mul(f1, tw[kf]), s1 = A::mul(f2, tw[kf * 2]), s2 = A::mul(f3, tw[kf * 3]);
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
        const cpx s1 = A::mul(f1, tw[kf]), s2 = A::mul(f2, tw[kf * 2]);
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
        const cpx s1 = A::mul(f1, tw[kf]), s2 = A::mul(f2, tw[kf * 2]);
        const cpx s3 = A::mul(f3, tw[kf * 3]), s4 = A::mul(f4, tw[kf * 4]);
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

template <typename A, bool TWLDS>
__global__ __launch_bounds__(1024) void fft_mixed_kernel(const typename A::cpx *__restrict__ in, typename A::cpx *__restrict__ out,
                                                         int N, size_t nframes, const typename A::cpx *__restrict__ tw_global,
                                                         const uint16_t *__restrict__ iperm, MixedPlan plan, int inverse)
{
    typedef typename A::cpx cpx;
    typedef typename A::scalar scalar;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int nt = blockDim.x, FPW = plan.fpw;
    const size_t ngroups = (nframes + FPW - 1) / FPW;
    cpx *tw_lds = reinterpret_cast<cpx *>(smem_raw) + (size_t)FPW * N * (plan.needs_pingpong ? 2 : 1);
    if (TWLDS)
        for (int i = threadIdx.x; i < N; i += nt) tw_lds[i] = tw_global[i];
    const cpx *tw = TWLDS ? tw_lds : tw_global;
    for (size_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        cpx *cur = reinterpret_cast<cpx *>(smem_raw);
        cpx *alt = cur + FPW * N;
        const size_t f0 = grp * FPW;
        const int nvalid = nframes - f0 < (size_t)FPW ? (int)(nframes - f0) : FPW;   // frames of this group
        const cpx *src = in + f0 * (size_t)N;
        __syncthreads();
        // leaves of kf_work: input index sum_s q_s*fstride_s goes to position sum_s q_s*m_s (host table, inverted so
        // that the global reads are contiguous and the scatter lands in LDS)
        for (int i = threadIdx.x; i < nvalid * N; i += nt) {
            const int fi = fdiv(i, plan.inv_n), e = i - mul24(fi, N);
            cur[mul24(fi, N) + iperm[e]] = src[i];
        }
        __syncthreads();
        for (int s = plan.nstages - 1; s >= 0; s--) {
            const int p = plan.radix[s], m = plan.m[s], fstride = plan.fstride[s];
            if (p <= 1) { __syncthreads(); continue; }     // never in a real plan: the timing-only skeleton (PCX_FFT_MIXED_DIAG=2)
            if (p == 2 || p == 3 || p == 4 || p == 5) {
                const int nb = plan.nb[s];
                for (int bb = threadIdx.x; bb < nvalid * nb; bb += nt) {
                    const int fi = fdiv(bb, plan.inv_nb[s]), b = bb - mul24(fi, nb);
                    const int g = fdiv(b, plan.inv_m[s]), k = b - mul24(g, m);
                    cpx *F = cur + mul24(fi, N) + mul24(g, mul24(p, m)) + k;
                    const int kf = mul24(k, fstride);   // twiddle index of W^k; its multiples are below radix * N < 2^24
                    kf_bfly<A>(p, F, m, tw, kf, fstride, inverse);
                }
                __syncthreads();
            } else {
                // kf_bfly_generic, one output element per lane, cur -> alt
                const int span = p * m;
                for (int ee = threadIdx.x; ee < nvalid * N; ee += nt) {
                    const int fi = fdiv(ee, plan.inv_n), e = ee - mul24(fi, N);
                    const int g = fdiv(e, plan.inv_span[s]), within = e - mul24(g, span);   // within = u + q1*m
                    const int u = within - mul24(fdiv(within, plan.inv_m[s]), m);
                    const cpx *S = cur + mul24(fi, N) + mul24(g, span) + u;      // scratch[q] = fixdiv(S[q*m])
                    const int k = within;                            // the reference's running k = u + q1*m
                    cpx acc = A::fixdiv(S[0], p);
                    const int step = mul24(fstride, k);              // < N
                    int twidx = 0;
                    for (int q = 1; q < p; q++) {
                        twidx += step;
                        if (twidx >= N) twidx -= N;
                        S += m;
                        acc = A::add(acc, A::mul(A::fixdiv(S[0], p), tw[twidx]));
                    }
                    alt[ee] = acc;
                }
                __syncthreads();
                cpx *t = cur; cur = alt; alt = t;
            }
        }
        cpx *dst = out + f0 * (size_t)N;
        for (int i = threadIdx.x; i < nvalid * N; i += nt) dst[i] = cur[i];
        (void)sizeof(scalar);
    }
}

// --------------------------------------------------------------------------------- //
// complex_float32, 5-smooth sizes (2^a 3^b 5^c: 1536, 1200, 600, 1000, 1920 ...): the same in-place
// decimation-in-time passes over the LDS image, but with the radices a float transform is free to choose --
// 16s first (fft16_plain of fft4096.hpp in registers), then 8 / 4 / 2, then 5s and 3s -- where kissfft's order
// (4s, 2s, 3s, 5s: kissfft.hh:38-55) is only binding for the bit-exact Q15 path.  1536 bins take 3 passes
// (16, 16, 2*3 ...) instead of 6, and one lane does 16 points' worth of arithmetic per index computation.
// Forward twiddles only: the inverse is conj . FFT . conj, folded into the leaf scatter and the final store.
// Same DFT as kissfft<float>::transform (kissfft.hh:81-161), parity bar 1e-5 of max|X|.
// --------------------------------------------------------------------------------- //
// element-type glue for the 5-smooth plan: E = fft4k::cf (float pair) or fft64::cd (double pair)
template <typename E> struct ElemOf;
template <> struct ElemOf<fft4k::cf> { typedef float S; typedef float2 G; };
template <> struct ElemOf<fft64::cd> { typedef double S; typedef double2 G; };
template <typename E> __device__ __forceinline__ E splat(double c) { typedef typename ElemOf<E>::S S; return E{(S)c, (S)c}; }
__device__ __forceinline__ fft4k::cf cm(fft4k::cf a, fft4k::cf w) { return fft4k::cmul1(a, w); }
__device__ __forceinline__ fft64::cd cm(fft64::cd a, fft64::cd w) { return fft64::cmul(a, w); }
using fft4k::fft16_plain; using fft64::fft16_plain;
using fft4k::fft8; using fft64::fft8;
using fft4k::fft4; using fft64::fft4;

// forward 3- and 5-point DFTs in place, natural order
template <typename E>
__device__ __forceinline__ void dft3(E &a, E &b, E &c)
{
    // X1 = a - (b+c)/2 - i (sqrt3/2)(b-c), X2 = a - (b+c)/2 + i (sqrt3/2)(b-c)
    const E sum = b + c, d = (b - c) * splat<E>(0.86602540378443864676);
    const E h = a - sum * splat<E>(0.5);
    a = a + sum;
    b = E{h.x + d.y, h.y - d.x};
    c = E{h.x - d.y, h.y + d.x};
}
template <typename E>
__device__ __forceinline__ void dft5(E &a, E &x1, E &x2, E &x3, E &x4)
{
    // with c1 = cos(2pi/5), c2 = cos(4pi/5), s1 = sin(2pi/5), s2 = sin(4pi/5):
    // X1,4 = a + c1 t1 + c2 t2 -+ i (s1 t3 + s2 t4),  X2,3 = a + c2 t1 + c1 t2 -+ i (s2 t3 - s1 t4)
    constexpr double C1 = 0.30901699437494742410, C2 = -0.80901699437494742410;
    constexpr double S1 = 0.95105651629515357212, S2 = 0.58778525229247312917;
    const E t1 = x1 + x4, t2 = x2 + x3, t3 = x1 - x4, t4 = x2 - x3;
    const E u1 = a + t1 * splat<E>(C1) + t2 * splat<E>(C2), u2 = a + t1 * splat<E>(C2) + t2 * splat<E>(C1);
    const E w1 = t3 * splat<E>(S1) + t4 * splat<E>(S2), w2 = t3 * splat<E>(S2) - t4 * splat<E>(S1);
    a = a + t1 + t2;
    x1 = E{u1.x + w1.y, u1.y - w1.x};         // u1 - i w1
    x4 = E{u1.x - w1.y, u1.y + w1.x};         // u1 + i w1
    x2 = E{u2.x + w2.y, u2.y - w2.x};
    x3 = E{u2.x - w2.y, u2.y + w2.x};
}

// W_9^e (forward sign), the inner twiddles of the 3 x 3 pass: indexed with compile-time constants only, so every use
// folds to immediates.  (A 5 x 5 pass was built the same way and dropped: its 50 data registers halve the occupancy of
// the whole kernel -- 3000 bins fell from 165 to 113 Gsamples/s -- for one pass saved on sizes with 5^2.)
struct W2 { double c, s; };
__device__ constexpr W2 kW9[9] = {{1.0, -0.0}, {0.766044443118978, -0.6427876096865393}, {0.17364817766693041, -0.984807753012208}, {-0.4999999999999998, -0.8660254037844387}, {-0.9396926207859083, -0.3420201433256689}, {-0.9396926207859084, 0.34202014332566866}, {-0.5000000000000004, 0.8660254037844384}, {0.17364817766692997, 0.9848077530122081}, {0.7660444431189778, 0.6427876096865396}};
// v[n] = x[n] in, v[k] = X[k] out: DFT of 9 points as 3 DFT_3's, the inner twiddles W_9^(n2 k1), 3 DFT_3's
template <typename E>
__device__ __forceinline__ void dft9(E (&v)[9])
{
    typedef typename ElemOf<E>::S S;
    E y[3][3];                                    // y[n2][k1]
#pragma unroll
    for (int n2 = 0; n2 < 3; n2++) {
        E a = v[n2], b = v[3 + n2], c = v[6 + n2];
        dft3(a, b, c);
        y[n2][0] = a; y[n2][1] = b; y[n2][2] = c;
#pragma unroll
        for (int k1 = 1; k1 < 3; k1++) {
            if (n2 == 0) continue;
            const S wc = (S)kW9[(n2 * k1) % 9].c, ws = (S)kW9[(n2 * k1) % 9].s;
            const E t = y[n2][k1];
            y[n2][k1] = E{t.x * wc - t.y * ws, t.x * ws + t.y * wc};
        }
    }
#pragma unroll
    for (int k1 = 0; k1 < 3; k1++) {
        E a = y[0][k1], b = y[1][k1], c = y[2][k1];
        dft3(a, b, c);
        v[k1] = a; v[k1 + 3] = b; v[k1 + 6] = c;
    }
}

template <typename cf, bool TWLDS, bool INV, bool PAD>
__global__ __launch_bounds__(1024) void fft_smooth_kernel(const typename ElemOf<cf>::G *__restrict__ in, typename ElemOf<cf>::G *__restrict__ out, int N,
                                                          size_t nframes, const typename ElemOf<cf>::G *__restrict__ tw_global,
                                                          const uint16_t *__restrict__ iperm, MixedPlan plan)
{
    using fft4k::bin_of;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int nt = blockDim.x, FPW = plan.fpw;
    const size_t ngroups = (nframes + FPW - 1) / FPW;
    cf *cur = reinterpret_cast<cf *>(smem_raw);
    const int img = FPW * N;
    cf *tw_lds = cur + (PAD ? img + (img >> 4) + 1 : img);
    auto P = [](int i) { return PAD ? i + (i >> 4) : i; };   // padded image: spans that are multiples of 32 elements no longer alias
    const cf *twg = reinterpret_cast<const cf *>(tw_global);
    if (TWLDS)
        for (int i = threadIdx.x; i < N; i += nt) tw_lds[i] = twg[i];
    const cf *tw = TWLDS ? tw_lds : twg;
    for (size_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const size_t f0 = grp * FPW;
        const int nvalid = nframes - f0 < (size_t)FPW ? (int)(nframes - f0) : FPW;
        const cf *src = reinterpret_cast<const cf *>(in) + f0 * (size_t)N;
        __syncthreads();
        for (int i = threadIdx.x; i < nvalid * N; i += nt) {
            const int fi = fdiv(i, plan.inv_n), e = i - mul24(fi, N);
            const cf t = src[i];
            cur[P(mul24(fi, N) + iperm[e])] = cf{t.x, INV ? -t.y : t.y};
        }
        __syncthreads();
        for (int s = plan.nstages - 1; s >= 0; s--) {
            const int p = plan.radix[s], m = plan.m[s], fstride = plan.fstride[s], nb = plan.nb[s];
            for (int bb = threadIdx.x; bb < nvalid * nb; bb += nt) {
                const int fi = fdiv(bb, plan.inv_nb[s]), b = bb - mul24(fi, nb);
                const int g = fdiv(b, plan.inv_m[s]), k = b - mul24(g, m);
                const int fb = mul24(fi, N) + mul24(g, mul24(p, m)) + k;   // element j of the butterfly: cur[P(fb + j*m)]
                const int kf = mul24(k, fstride);       // W^k; j * kf < N for j < p
                if (p == 16) {
                    cf v[16];
#pragma unroll
                    for (int j = 0; j < 16; j++) v[j] = cur[P(fb + j * m)];
#pragma unroll
                    for (int j = 1; j < 16; j++) v[j] = cm(v[j], tw[j * kf]);
                    fft16_plain(v);
#pragma unroll
                    for (int q = 0; q < 16; q++) cur[P(fb + bin_of(q) * m)] = v[q];
                } else if (p == 8) {
                    cf v[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) v[j] = cur[P(fb + j * m)];
#pragma unroll
                    for (int j = 1; j < 8; j++) v[j] = cm(v[j], tw[j * kf]);
                    fft8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
#pragma unroll
                    for (int j = 0; j < 8; j++) cur[P(fb + j * m)] = v[j];
                } else if (p == 4) {
                    cf a = cur[P(fb)], b1 = cm(cur[P(fb + m)], tw[kf]), c = cm(cur[P(fb + 2 * m)], tw[2 * kf]), d = cm(cur[P(fb + 3 * m)], tw[3 * kf]);
                    fft4(a, b1, c, d);
                    cur[P(fb)] = a; cur[P(fb + m)] = b1; cur[P(fb + 2 * m)] = c; cur[P(fb + 3 * m)] = d;
                } else if (p == 2) {
                    const cf a = cur[P(fb)], t = cm(cur[P(fb + m)], tw[kf]);
                    cur[P(fb)] = a + t;
                    cur[P(fb + m)] = a - t;
                } else if (p == 3) {
                    cf a = cur[P(fb)], b1 = cm(cur[P(fb + m)], tw[kf]), c = cm(cur[P(fb + 2 * m)], tw[2 * kf]);
                    dft3(a, b1, c);
                    cur[P(fb)] = a; cur[P(fb + m)] = b1; cur[P(fb + 2 * m)] = c;
                } else if (p == 5) {
                    cf a = cur[P(fb)], x1 = cm(cur[P(fb + m)], tw[kf]), x2 = cm(cur[P(fb + 2 * m)], tw[2 * kf]),
                       x3 = cm(cur[P(fb + 3 * m)], tw[3 * kf]), x4 = cm(cur[P(fb + 4 * m)], tw[4 * kf]);
                    dft5(a, x1, x2, x3, x4);
                    cur[P(fb)] = a; cur[P(fb + m)] = x1; cur[P(fb + 2 * m)] = x2; cur[P(fb + 3 * m)] = x3; cur[P(fb + 4 * m)] = x4;
                } else if (p == 6) {
                    // 6 = 2 x 3, coprime: prime-factor map, no inner twiddles.  n = (3 n1 + 2 n2) mod 6, k = (3 k1 + 4 k2) mod 6
                    cf v[6];
#pragma unroll
                    for (int j = 0; j < 6; j++) v[j] = cur[P(fb + j * m)];
#pragma unroll
                    for (int j = 1; j < 6; j++) v[j] = cm(v[j], tw[j * kf]);
                    cf y[2][3];                                   // y[k1][n2]
#pragma unroll
                    for (int n2 = 0; n2 < 3; n2++) {
                        const cf e = v[(2 * n2) % 6], o = v[(3 + 2 * n2) % 6];
                        y[0][n2] = e + o;
                        y[1][n2] = e - o;
                    }
#pragma unroll
                    for (int k1 = 0; k1 < 2; k1++) {
                        dft3(y[k1][0], y[k1][1], y[k1][2]);
#pragma unroll
                        for (int k2 = 0; k2 < 3; k2++) cur[P(fb + ((3 * k1 + 4 * k2) % 6) * m)] = y[k1][k2];
                    }
                } else if (p == 9) {
                    cf v[9];
#pragma unroll
                    for (int j = 0; j < 9; j++) v[j] = cur[P(fb + j * m)];
#pragma unroll
                    for (int j = 1; j < 9; j++) v[j] = cm(v[j], tw[j * kf]);
                    dft9(v);
#pragma unroll
                    for (int j = 0; j < 9; j++) cur[P(fb + j * m)] = v[j];
                } else {   // p == 15 = 3 x 5: n = (5 n1 + 3 n2) mod 15, k = (10 k1 + 6 k2) mod 15
                    cf v[15];
#pragma unroll
                    for (int j = 0; j < 15; j++) v[j] = cur[P(fb + j * m)];
#pragma unroll
                    for (int j = 1; j < 15; j++) v[j] = cm(v[j], tw[j * kf]);
                    cf y[3][5];                                   // y[k1][n2]
#pragma unroll
                    for (int n2 = 0; n2 < 5; n2++) {
                        cf a = v[(3 * n2) % 15], b1 = v[(5 + 3 * n2) % 15], c = v[(10 + 3 * n2) % 15];
                        dft3(a, b1, c);
                        y[0][n2] = a; y[1][n2] = b1; y[2][n2] = c;
                    }
#pragma unroll
                    for (int k1 = 0; k1 < 3; k1++) {
                        dft5(y[k1][0], y[k1][1], y[k1][2], y[k1][3], y[k1][4]);
#pragma unroll
                        for (int k2 = 0; k2 < 5; k2++) cur[P(fb + ((10 * k1 + 6 * k2) % 15) * m)] = y[k1][k2];
                    }
                }
            }
            __syncthreads();
        }
        cf *dst = reinterpret_cast<cf *>(out) + f0 * (size_t)N;
        for (int i = threadIdx.x; i < nvalid * N; i += nt) {
            const cf t = cur[P(i)];
            dst[i] = cf{t.x, INV ? -t.y : t.y};
        }
    }
}

template <typename A>
int launch_mixed(const void *in, void *out, size_t nbins, size_t nframes, bool inverse, const void *tw, const void *iperm, const int *radix,
                 int nstages, hipStream_t st)
{
    if (nframes == 0) return PCX_OK;
    if (nstages > kMaxStagesMixed) { set_error("fft: too many stages for numBins %zu", nbins); return PCX_ERR_UNSUPPORTED; }
    auto magic = [](size_t d) { return 1.0f / (float)d; };
    MixedPlan plan;
    plan.nstages = nstages;
    plan.needs_pingpong = 0;
    plan.inv_n = magic(nbins);
    size_t m = 1;
    for (int s = nstages - 1; s >= 0; s--) {
        const size_t p = (size_t)radix[s];
        plan.radix[s] = radix[s];
        plan.m[s] = (int)m;
        plan.fstride[s] = (int)(nbins / (p * m));
        plan.nb[s] = (int)(nbins / p);
        plan.inv_m[s] = magic(m);
        plan.inv_nb[s] = magic(nbins / p);
        plan.inv_span[s] = magic(p * m);
        if (p != 2 && p != 3 && p != 4 && p != 5) plan.needs_pingpong = 1;
        m *= p;
    }
    const size_t images = plan.needs_pingpong ? 2 : 1;
    if (nbins * sizeof(typename A::cpx) * images > 160 * 1024) {
        set_error("fft: numBins %zu does not fit the single-workgroup LDS plan (%zu bytes)", nbins, nbins * sizeof(typename A::cpx) * images);
        return PCX_ERR_UNSUPPORTED;
    }
    // short frames share a workgroup: about 4096 elements per group, so that every pass has >= 1024 butterflies
    // PCX_FFT_MIXED_ELEMS (A/B): elements per workgroup the frame count is sized for
    const size_t group_elems = (size_t)PCX_ENV_INT("PCX_FFT_MIXED_ELEMS", 4096);
    size_t fpw = nbins >= group_elems ? 1 : group_elems / nbins;
    if (fpw > nframes) fpw = nframes;
    plan.fpw = (int)fpw;
    size_t lds = fpw * nbins * sizeof(typename A::cpx) * images;
    // PCX_FFT_MIXED_TWLDS=0 (A/B) keeps the twiddles in global memory
    const int tw_in_lds = (int)PCX_ENV_INT("PCX_FFT_MIXED_TWLDS", 1);
    const bool twlds = tw_in_lds && lds + nbins * sizeof(typename A::cpx) <= 80 * 1024;
    if (twlds) lds += nbins * sizeof(typename A::cpx);
    auto k = twlds ? fft_mixed_kernel<A, true> : fft_mixed_kernel<A, false>;
    if (lds > 64 * 1024) PCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    unsigned threads = (unsigned)((fpw * nbins + 3) / 4);
    threads = (threads + 63) / 64 * 64;
    if (threads < 64) threads = 64;
    if (threads > 1024) threads = 1024;
    // PCX_FFT_MIXED_DIAG=1 (timing only, wrong outputs): no butterfly passes, the load/scatter/store skeleton alone
    const int diag = (int)PCX_ENV_INT("PCX_FFT_MIXED_DIAG", 0);
    if (diag == 1) plan.nstages = 0;
    if (diag == 2) for (int q = 0; q < nstages; q++) plan.radix[q] = 1;   // stage loop, plan loads and barriers without butterflies
    const size_t ngroups = (nframes + fpw - 1) / fpw;
    unsigned per_cu = (unsigned)(160 * 1024 / lds);
    const unsigned by_threads = 2048 / threads;
    if (per_cu > by_threads) per_cu = by_threads;
    if (per_cu < 1) per_cu = 1;
    const long mixed_rounds = PCX_ENV_INT("PCX_MIXED_ROUNDS", 0);   // (diagnostic library: groups per workgroup instead of equal shares, A/B)
    const unsigned grid = mixed_rounds > 0 ? rounds_grid(ngroups, 256 * per_cu, (unsigned)mixed_rounds) : persistent_grid(ngroups, 256 * per_cu);
    hipLaunchKernelGGL(k, dim3(grid), dim3(threads), lds, st, (const typename A::cpx *)in, (typename A::cpx *)out, (int)nbins, nframes,
                       (const typename A::cpx *)tw, (const uint16_t *)iperm, plan, inverse ? 1 : 0);
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}

template <typename E>
int launch_smooth(const void *in, void *out, size_t nbins, size_t nframes, bool inverse, const void *tw, const void *iperm, const int *radix,
                  int nstages, hipStream_t st)
{
    typedef typename ElemOf<E>::G G;
    constexpr size_t EB = sizeof(E);
    if (nframes == 0) return PCX_OK;
    if (nstages > kMaxStagesMixed) { set_error("fft: too many stages for numBins %zu", nbins); return PCX_ERR_UNSUPPORTED; }
    MixedPlan plan;
    plan.nstages = nstages;
    plan.needs_pingpong = 0;
    plan.inv_n = 1.0f / (float)nbins;
    size_t m = 1;
    for (int s = nstages - 1; s >= 0; s--) {
        const size_t p = (size_t)radix[s];
        plan.radix[s] = radix[s];
        plan.m[s] = (int)m;
        plan.fstride[s] = (int)(nbins / (p * m));
        plan.nb[s] = (int)(nbins / p);
        plan.inv_m[s] = 1.0f / (float)m;
        plan.inv_nb[s] = 1.0f / (float)(nbins / p);
        plan.inv_span[s] = 1.0f / (float)(p * m);
        m *= p;
    }
    const size_t group = 32768 / EB;     // 4096 float / 2048 double elements per group
    size_t fpw = nbins >= group ? 1 : group / nbins;
    if (fpw > nframes) fpw = nframes;
    plan.fpw = (int)fpw;
    // PCX_FFT_SMOOTH_PAD=1 (A/B): image padded i + i/16.  Measured: 1536 bins (span 96 = 3 * 32 elements, the worst
    // aliasing case) unchanged at 140 Gsamples/s, every other size 5-15 % slower from the extra index arithmetic -- the
    // passes wait on VALU issue, not on LDS banks -- so the plain image is the default
    const int pad = (int)PCX_ENV_INT("PCX_FFT_SMOOTH_PAD", 0);
    const size_t img = fpw * nbins;
    size_t lds = (pad ? img + img / 16 + 1 : img) * EB;
    const bool twlds = lds + nbins * EB <= 96 * 1024;
    if (twlds) lds += nbins * EB;
    if (lds > 160 * 1024) { set_error("fft: numBins %zu does not fit the single-workgroup LDS plan", nbins); return PCX_ERR_UNSUPPORTED; }
    auto k = pad ? (twlds ? (inverse ? fft_smooth_kernel<E, true, true, true> : fft_smooth_kernel<E, true, false, true>)
                          : (inverse ? fft_smooth_kernel<E, false, true, true> : fft_smooth_kernel<E, false, false, true>))
                 : (twlds ? (inverse ? fft_smooth_kernel<E, true, true, false> : fft_smooth_kernel<E, true, false, false>)
                          : (inverse ? fft_smooth_kernel<E, false, true, false> : fft_smooth_kernel<E, false, false, false>));
    if (lds > 64 * 1024) PCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // lanes per group = elements / div.  Measured (tools/sweep_fft_mixed.py with PCX_FFT_SMOOTH_DIV = 4 ... 16): more, smaller
    // workgroups per CU beat one butterfly per lane in every pass -- 6 for short frames, 8 to 4095 bins, 4 beyond
    int rmin = nbins < 256 ? 6 : nbins < 4096 ? 8 : 4;
    const int div_forced = (int)PCX_ENV_INT("PCX_FFT_SMOOTH_DIV", 0);
    if (div_forced > 0) rmin = div_forced;
    unsigned threads = (unsigned)((fpw * nbins + rmin - 1) / rmin);
    threads = (threads + 63) / 64 * 64;
    if (threads < 64) threads = 64;
    if (threads > 1024) threads = 1024;
    const size_t ngroups = (nframes + fpw - 1) / fpw;
    unsigned per_cu = (unsigned)(160 * 1024 / lds);
    const unsigned by_threads = 2048 / threads;
    if (per_cu > by_threads) per_cu = by_threads;
    if (per_cu < 1) per_cu = 1;
    const long mixed_rounds = PCX_ENV_INT("PCX_MIXED_ROUNDS", 0);   // (diagnostic library: groups per workgroup instead of equal shares, A/B)
    const unsigned grid = mixed_rounds > 0 ? rounds_grid(ngroups, 256 * per_cu, (unsigned)mixed_rounds) : persistent_grid(ngroups, 256 * per_cu);
    hipLaunchKernelGGL(k, dim3(grid), dim3(threads), lds, st, (const G *)in, (G *)out, (int)nbins, nframes, (const G *)tw,
                       (const uint16_t *)iperm, plan);
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}

}  // namespace

// --------------------------------------------------------------------------------- //
// complex_int16 frames beyond one workgroup's LDS (power-of-two sizes above 32,768 bins, other sizes above ~20,000): kf_work's
// recursion (kiss_fft.c:237-302) unrolled over GLOBAL memory, one launch per stage.  The reference takes any numBins
// (FFT.cpp:83-93, kiss_fft.c:339-368); a four-step split would round in another order, but the recursion itself is a leaf gather
// followed by the butterfly passes bottom-up, the butterflies of one pass touch disjoint elements, and every one of them is
// kf_bfly (above) with its C_FIXDIV per stage -- so running a pass as one launch over all frames leaves every rounding where the
// reference has it: bit-exact, whatever the size.  Generic radices (primes above 5) go cur -> alt as in the LDS kernel.
// 2 + 2 bytes per element read and written per pass: a correctness path for sizes no DSP block normally asks of a Q15 transform.
// --------------------------------------------------------------------------------- //
struct GlobalPlan {
    int nstages;
    int radix[kMaxStagesMixed], m[kMaxStagesMixed], fstride[kMaxStagesMixed];
};
template <typename A>
__global__ __launch_bounds__(256) void fft_global_gather_kernel(const typename A::cpx *__restrict__ in, typename A::cpx *__restrict__ out, unsigned N,
                                                                size_t total, GlobalPlan plan)
{
    // position sum_s q_s m_s  <-  input index sum_s q_s fstride_s  (the m == 1 leaves, kiss_fft.c:276-280)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t f = i / N;
        unsigned rem = (unsigned)(i - f * N), idx = 0;
        for (int s = 0; s < plan.nstages; s++) {
            const unsigned q = rem / (unsigned)plan.m[s];
            rem -= q * (unsigned)plan.m[s];
            idx += q * (unsigned)plan.fstride[s];
        }
        out[i] = in[f * N + idx];
    }
}
template <typename A>
__global__ __launch_bounds__(256) void fft_global_stage_kernel(typename A::cpx *__restrict__ buf, unsigned N, size_t total_bfly, int p, int m, int fstride,
                                                               const typename A::cpx *__restrict__ tw, int inverse)
{
    const unsigned nb = N / (unsigned)p;
    for (size_t bb = (size_t)blockIdx.x * blockDim.x + threadIdx.x; bb < total_bfly; bb += (size_t)gridDim.x * blockDim.x) {
        const size_t f = bb / nb;
        const unsigned b = (unsigned)(bb - f * nb);
        const unsigned g = b / (unsigned)m, k = b - g * (unsigned)m;
        kf_bfly<A>(p, buf + f * N + (size_t)g * (unsigned)(p * m) + k, m, tw, (int)(k * (unsigned)fstride), fstride, inverse);
    }
}
template <typename A>
__global__ __launch_bounds__(256) void fft_global_generic_kernel(const typename A::cpx *__restrict__ cur, typename A::cpx *__restrict__ alt, unsigned N,
                                                                 size_t total, int p, int m, int fstride, const typename A::cpx *__restrict__ tw)
{
    typedef typename A::cpx cpx;
    const unsigned span = (unsigned)(p * m);
    for (size_t ee = (size_t)blockIdx.x * blockDim.x + threadIdx.x; ee < total; ee += (size_t)gridDim.x * blockDim.x) {
        const size_t f = ee / N;
        const unsigned e = (unsigned)(ee - f * N);
        const unsigned g = e / span, within = e - g * span, u = within % (unsigned)m;
        const cpx *S = cur + f * N + (size_t)g * span + u;
        cpx acc = A::fixdiv(S[0], p);
        const unsigned step = (unsigned)(((unsigned long long)fstride * within) % N);    // kf_bfly_generic: twidx += fstride*k, reduced modulo Norig
        unsigned twidx = 0;
        for (int q = 1; q < p; q++) {
            twidx += step;
            if (twidx >= N) twidx -= N;
            S += m;
            acc = A::add(acc, A::mul(A::fixdiv(S[0], p), tw[twidx]));
        }
        alt[ee] = acc;
    }
}

int launch_fft_q15_global(const void *in, void *out, void *ws, size_t nbins, size_t nframes, bool inverse, const void *tw, const int *radix,
                          int nstages, hipStream_t st)
{
    typedef Q15Arith A;
    if (nframes == 0) return PCX_OK;
    if (nstages > kMaxStagesMixed || nbins >= ((size_t)1 << 27)) { set_error("fft (Q15, global plan): %zu bins", nbins); return PCX_ERR_UNSUPPORTED; }
    GlobalPlan plan;
    plan.nstages = nstages;
    size_t m = nbins, fs = 1;
    for (int s = 0; s < nstages; s++) {
        plan.radix[s] = radix[s];
        m /= (size_t)radix[s];
        plan.m[s] = (int)m;
        plan.fstride[s] = (int)fs;
        fs *= (size_t)radix[s];
    }
    const size_t total = nframes * nbins;
    const unsigned N = (unsigned)nbins;
    A::cpx *cur = (A::cpx *)out, *alt = (A::cpx *)ws;
    if (in == out) {            // a transform in place: the gather needs another destination
        if (!ws) { set_error("fft (Q15, global plan): in-place needs the workspace"); return PCX_ERR_STATE; }
        cur = (A::cpx *)ws; alt = (A::cpx *)out;
    }
    hipLaunchKernelGGL(fft_global_gather_kernel<A>, dim3(stream_grid(total, 256)), dim3(256), 0, st, (const A::cpx *)in, cur, N, total, plan);
    PCX_LAUNCH_CHECK();
    for (int s = nstages - 1; s >= 0; s--) {
        const int p = plan.radix[s];
        if (p <= 5) {
            const size_t nbfly = total / (size_t)p;
            hipLaunchKernelGGL(fft_global_stage_kernel<A>, dim3(stream_grid(nbfly, 256)), dim3(256), 0, st, cur, N, nbfly, p, plan.m[s], plan.fstride[s],
                               (const A::cpx *)tw, inverse ? 1 : 0);
        } else {
            if (!alt) { set_error("fft (Q15, global plan): radix %d needs the workspace", p); return PCX_ERR_STATE; }
            hipLaunchKernelGGL(fft_global_generic_kernel<A>, dim3(stream_grid(total, 256)), dim3(256), 0, st, (const A::cpx *)cur, alt, N, total, p,
                               plan.m[s], plan.fstride[s], (const A::cpx *)tw);
            A::cpx *t = cur; cur = alt; alt = t;
        }
        PCX_LAUNCH_CHECK();
    }
    if (cur != (A::cpx *)out) PCX_HIP(hipMemcpyAsync(out, cur, total * sizeof(A::cpx), hipMemcpyDeviceToDevice, st));
    return PCX_OK;
}

int launch_fft_mixed(int scalar, const void *in, void *out, size_t nbins, size_t nframes, bool inverse, const void *tw, const void *iperm,
                     const int *radix_host, int nstages, hipStream_t st)
{
    switch (scalar) {
    case PCX_F32: return launch_mixed<FloatArith<float>>(in, out, nbins, nframes, inverse, tw, iperm, radix_host, nstages, st);
    case PCX_F64: return launch_mixed<FloatArith<double>>(in, out, nbins, nframes, inverse, tw, iperm, radix_host, nstages, st);
    case PCX_I16: return launch_mixed<Q15Arith>(in, out, nbins, nframes, inverse, tw, iperm, radix_host, nstages, st);
    }
    set_error("fft: unsupported scalar %d", scalar);
    return PCX_ERR_ARG;
}

}  // namespace pcx

namespace pcx {
// complex_float32 / complex_float64, numBins = 2^a 3^b 5^c: radices (16 / 8 / 4 / 2 / 6 / 15 / 9 / 5 / 3) chosen by pcx_fft_api.hip,
// forward twiddle table of the element type
int launch_fft_smooth(int scalar, const void *in, void *out, size_t nbins, size_t nframes, bool inverse, const void *tw, const void *iperm,
                      const int *radix_host, int nstages, hipStream_t st)
{
    return scalar == PCX_F64 ? launch_smooth<fft64::cd>(in, out, nbins, nframes, inverse, tw, iperm, radix_host, nstages, st)
                             : launch_smooth<fft4k::cf>(in, out, nbins, nframes, inverse, tw, iperm, radix_host, nstages, st);
}
}  // namespace pcx
