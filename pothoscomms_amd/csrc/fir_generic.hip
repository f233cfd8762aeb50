// fir_generic.hip -- the /comms/fir_filter loop for EVERY type combination the
// reference factory accepts (FIRFilter.cpp:369-384) and any interpolation L /
// decimation M: one output per lane, polyphase row j = i % L, input n = i / L with
// i = (o+1)*M - 1 (the flat iteration index on which FIRFilter.cpp:291-292 emits).
//
// This is the coverage + bit-exactness kernel, not the fast path:
//   * integer types: exact ring arithmetic modulo 2^bits(Q) (std::complex<intN>
//     wraps), then fromQ (>> bits/2) -- bit-identical to the reference by construction;
//   * float types, EXACT=true: products and sums in the reference's order with no FMA
//     contraction (this TU is built with -ffp-contract=off) -- bit-identical to
//     FIRFilter.cpp:295-300 on baseline x86-64;
//   * float types, EXACT=false: same order, fused multiply-add.
// The LDS-tiled direct kernel (fir_direct.hip) and the frequency-domain kernel
// (fir_ols.hip) are the performance paths for complex_float32, M=L=1.
#include "pcx_cplx.hpp"
#include "pcx_internal.hpp"

#include <cstdlib>
#include <type_traits>

namespace pcx {

__device__ __forceinline__ float t_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double t_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

template <typename Q>
struct QComp {
    using type = typename std::conditional<(sizeof(Q) < 4), uint32_t, typename std::make_unsigned<Q>::type>::type;
};

// One output of a complex-taps float filter once more, in the reference's order, with every product going through the
// reference's complex multiply INCLUDING its slow path (pcx_cplx.hpp): what the EXACT kernels run for an output whose two
// parts both came out NaN -- the only outputs in which a product can have taken that path (a NaN + i NaN product makes the
// running sum NaN + i NaN for good).  xp -> x[n] (the newest sample of the window), taps in the stored row order.
// (inlined into a cold branch and returning by value: a call, or accumulators passed by reference, put the hot loop's
// registers into scratch -- measured -38 % at 15 taps)
template <typename S>
struct CPair { S re, im; };
template <typename S>
__device__ __forceinline__ CPair<S> fir_output_annex_g(const S *xp, const S *tp, size_t len)
{
    S ar = 0, ai = 0;
    for (size_t k = 0; k < len; k++) {
        const S a = tp[2 * k], b = tp[2 * k + 1], c = xp[-(ptrdiff_t)k * 2], d = xp[-(ptrdiff_t)k * 2 + 1];
        const S ac = a * c, bd = b * d, ad = a * d, bc = b * c;
        S pr = ac - bd, pi = ad + bc;
        if (both_nan(pr, pi)) cmul_annex_g(a, b, c, d, pr, pi);
        ar = ar + pr; ai = ai + pi;
    }
    return CPair<S>{ar, ai};
}

// S = element scalar, TT = stored tap scalar (float/double or Q int), CPLX/CTAPS flags
template <typename S, typename TT, bool CPLX, bool CTAPS, bool EXACT>
__global__ __launch_bounds__(256) void fir_generic_kernel(const S *__restrict__ in, S *__restrict__ out, size_t n_out,
                                                          size_t L, size_t M, size_t K,
                                                          const uint32_t *__restrict__ rowLen, const TT *__restrict__ rowTaps, QShift qs)
{
    constexpr bool FLT = std::is_floating_point<S>::value;
    constexpr int EW = CPLX ? 2 : 1, TW = CTAPS ? 2 : 1;
    const size_t gstride = (size_t)gridDim.x * blockDim.x;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < n_out; o += gstride) {
        const size_t i = (o + 1) * M - 1;
        const size_t n = i / L, j = i - n * L;
        const uint32_t len = rowLen[j];
        const TT *tp = rowTaps + j * K * TW;
        const S *xp = in + (K - 1 + n) * EW;  // x[n]; x[n-k] = xp - k*EW
        if constexpr (FLT) {
            S ar = 0, ai = 0;
            for (uint32_t k = 0; k < len; k++) {
                const S c = xp[-(ptrdiff_t)k * EW];
                const S d = CPLX ? xp[-(ptrdiff_t)k * EW + 1] : S(0);
                const S a = tp[k * TW];
                const S b = CTAPS ? tp[k * TW + 1] : S(0);
                if constexpr (!CPLX) {
                    if constexpr (EXACT) { const S p = a * c; ar = ar + p; }
                    else ar = t_fma(a, c, ar);
                } else if constexpr (!CTAPS) {
                    if constexpr (EXACT) { const S pr = c * a, pi = d * a; ar = ar + pr; ai = ai + pi; }
                    else { ar = t_fma(c, a, ar); ai = t_fma(d, a, ai); }
                } else {
                    if constexpr (EXACT) {
                        const S ac = a * c, bd = b * d, ad = a * d, bc = b * c;
                        const S pr = ac - bd, pi = ad + bc;
                        ar = ar + pr; ai = ai + pi;
                    } else {
                        ar = t_fma(a, c, ar); ar = t_fma(-b, d, ar);
                        ai = t_fma(a, d, ai); ai = t_fma(b, c, ai);
                    }
                }
            }
            if constexpr (EXACT && CPLX && CTAPS) {
                if (__builtin_expect(both_nan(ar, ai), 0)) {
                    const CPair<S> fx = fir_output_annex_g<S>(xp, reinterpret_cast<const S *>(tp), len);
                    ar = fx.re; ai = fx.im;
                }
            }
            out[o * EW] = ar;
            if constexpr (CPLX) out[o * EW + 1] = ai;
        } else {
            using C = typename QComp<TT>::type;  // TT is the Q type for integers
            C ar = 0, ai = 0;
            for (uint32_t k = 0; k < len; k++) {
                const C c = (C)(TT)xp[-(ptrdiff_t)k * EW];
                const C a = (C)tp[k * TW];
                if constexpr (!CPLX) {
                    ar += a * c;
                } else {
                    const C d = (C)(TT)xp[-(ptrdiff_t)k * EW + 1];
                    if constexpr (!CTAPS) { ar += a * c; ai += a * d; }
                    else {
                        const C b = (C)tp[k * TW + 1];
                        ar += a * c - b * d; ai += a * d + b * c;
                    }
                }
            }
            out[o * EW] = (S)from_q_bits<TT>((TT)ar, qs);        // fromQ<OutType>(y_n), FIRFilter.cpp:300, under the reading in force
            if constexpr (CPLX) out[o * EW + 1] = (S)from_q_bits<TT>((TT)ai, qs);
        }
    }
}

// --------------------------------------------------------------------------------- //
// M = L = 1 (no resampling), every type: R consecutive outputs per lane from a register sliding
// window.  Output o needs x[o + K-1 - k] for tap k; a lane that owns outputs o0 .. o0+R-1 keeps
// w[r] = x[o0 + r + K-1 - k] in registers, and stepping k by one shifts the window down by one
// sample: one new element per lane and tap instead of one per lane, tap AND output.  The k loop is
// unrolled by R so the shift is a static renaming; the R elements a group needs are loaded together
// ahead of the arithmetic.  Taps are wave-uniform (scalar loads).  Accumulation order per output is
// k = 0, 1, 2, ... exactly as FIRFilter.cpp:294-300, so the EXACT float mode stays bit-identical.
// MAD24 (int16 / int8 elements): both factors fit 24 bits (16-bit samples; Q taps checked on the
// host), so the products are v_mul_i32_i24 / v_mad_i32_i24 -- full rate where v_mul_lo_u32 runs at
// a quarter -- and still exact modulo 2^32.
// --------------------------------------------------------------------------------- //
template <typename S, typename TT, bool FLT> struct SlideAcc { typedef S type; };
template <typename S, typename TT> struct SlideAcc<S, TT, false> { typedef typename QComp<TT>::type type; };

// STAGE: the tile's 256*R + K-1 input elements are first copied into LDS with lane-contiguous loads (a lane's
// own window walks global memory with a stride of R elements between lanes otherwise: poorly coalesced) and
// the window reads come from there; the image is padded by one element per R so that the lane stride
// R+1 elements is conflict-free for 4-, 8- and 16-byte elements.
template <typename S, typename TT, bool CPLX, bool CTAPS, bool EXACT, bool MAD24, int R, bool STAGE>
__global__ __launch_bounds__(256) void fir_slide_kernel(const S *__restrict__ in, S *__restrict__ out, size_t n_out, size_t K,
                                                        const TT *__restrict__ taps, QShift qs)
{
    constexpr bool FLT = std::is_floating_point<S>::value;
    constexpr int EW = CPLX ? 2 : 1, TW = CTAPS ? 2 : 1;
    using C = typename SlideAcc<S, TT, FLT>::type;   // accumulator: S for floats, the unsigned compute type of Q otherwise
    struct alignas(sizeof(S) * EW) Elem { S v[EW]; };
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Elem *img = reinterpret_cast<Elem *>(smem_raw);
    const size_t gstride = (size_t)gridDim.x * blockDim.x * R;
    const size_t last = n_out - 1;          // input index o + K-1-k <= last + K-1 always exists
    for (size_t o0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * R; o0 - (size_t)threadIdx.x * R < n_out; o0 += gstride) {
        const size_t tile0 = o0 - (size_t)threadIdx.x * R;      // first output / input element of the workgroup's tile
        if (STAGE) {
            const size_t n_stage = (size_t)blockDim.x * R + K - 1;
            __syncthreads();
            for (size_t i = threadIdx.x; i < n_stage; i += blockDim.x) {
                const size_t g = tile0 + i < last + K ? tile0 + i : last + K - 1;   // clamp past the call's last sample (unused)
                img[i + i / R] = reinterpret_cast<const Elem *>(in)[g];
            }
            __syncthreads();
        }
        C ar[R], ai[R], wr[R], wi[R];
        auto fetch = [&](size_t idx, C &re, C &im) {
            if (STAGE) {
                const size_t i = idx - tile0;
                const Elem e = img[i + i / R];
                if constexpr (FLT) { re = e.v[0]; im = CPLX ? e.v[EW - 1] : S(0); }
                else { re = (C)(TT)e.v[0]; im = CPLX ? (C)(TT)e.v[EW - 1] : C(0); }
                return;
            }
            // idx may run past the last sample the call owns for the lanes of the ragged tail: clamp (unused)
            const size_t i = idx < last + K ? idx : last + K - 1;
            if constexpr (FLT) { re = in[i * EW]; im = CPLX ? in[i * EW + 1] : S(0); }
            else { re = (C)(TT)in[i * EW]; im = CPLX ? (C)(TT)in[i * EW + 1] : C(0); }
        };
#pragma unroll
        for (int r = 0; r < R; r++) { ar[r] = 0; ai[r] = 0; fetch(o0 + r + K - 1, wr[r], wi[r]); }
        for (size_t kb = 0; kb < K; kb += R) {
            // the R elements that enter the window during this group: x[o0 + K-2-k], k = kb .. kb+R-1
            C nr[R], ni[R];
#pragma unroll
            for (int u = 0; u < R; u++) {
                const size_t k = kb + u;
                if (k + 1 < K) fetch(o0 + K - 2 - k, nr[u], ni[u]);
                else { nr[u] = 0; ni[u] = 0; }
            }
#pragma unroll
            for (int u = 0; u < R; u++) {
                const size_t k = kb + u;
                if (k >= K) break;
                const C a = FLT ? (C)taps[k * TW] : (C)taps[k * TW];
                const C b = CTAPS ? (C)taps[k * TW + 1] : C(0);
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int ph = (r - u + R) % R;     // logical w[r] lives in physical slot (r - u) mod R
                    const C c = wr[ph], d = wi[ph];
                    if constexpr (FLT) {
                        if constexpr (!CPLX) {
                            if constexpr (EXACT) { const S p = a * c; ar[r] = ar[r] + p; }
                            else ar[r] = t_fma(a, c, ar[r]);
                        } else if constexpr (!CTAPS) {
                            if constexpr (EXACT) { const S pr = c * a, pi = d * a; ar[r] = ar[r] + pr; ai[r] = ai[r] + pi; }
                            else { ar[r] = t_fma(c, a, ar[r]); ai[r] = t_fma(d, a, ai[r]); }
                        } else {
                            if constexpr (EXACT) {
                                const S ac = a * c, bd = b * d, ad = a * d, bc = b * c;
                                const S pr = ac - bd, pi = ad + bc;
                                ar[r] = ar[r] + pr; ai[r] = ai[r] + pi;
                            } else {
                                ar[r] = t_fma(a, c, ar[r]); ar[r] = t_fma(-b, d, ar[r]);
                                ai[r] = t_fma(a, d, ai[r]); ai[r] = t_fma(b, c, ai[r]);
                            }
                        }
                    } else if constexpr (MAD24) {
                        const int ia = (int)a, ib = (int)b, ic = (int)c, id = (int)d;
                        if constexpr (!CPLX) ar[r] += (C)__mul24(ia, ic);
                        else if constexpr (!CTAPS) { ar[r] += (C)__mul24(ia, ic); ai[r] += (C)__mul24(ia, id); }
                        else {
                            ar[r] += (C)__mul24(ia, ic) - (C)__mul24(ib, id);
                            ai[r] += (C)__mul24(ia, id) + (C)__mul24(ib, ic);
                        }
                    } else {
                        if constexpr (!CPLX) ar[r] += a * c;
                        else if constexpr (!CTAPS) { ar[r] += a * c; ai[r] += a * d; }
                        else { ar[r] += a * c - b * d; ai[r] += a * d + b * c; }
                    }
                }
                // slide: the oldest slot (logical w[R-1]) takes the element that becomes w[0]
                const int slot = (R - 1 - u + R) % R;
                wr[slot] = nr[u]; wi[slot] = ni[u];
            }
        }
        // the lane's R outputs are R*EW*sizeof(S) contiguous bytes: 16-byte stores when the run is whole and aligned
        unsigned badmask = 0;
        S res[R * EW];
#pragma unroll
        for (int r = 0; r < R; r++) {
            if constexpr (FLT) {
                if constexpr (EXACT && CPLX && CTAPS) {
                    if (both_nan(ar[r], ai[r])) badmask |= 1u << r;     // see fir_output_annex_g: looked at again behind the stores
                }
                res[r * EW] = ar[r];
                if constexpr (CPLX) res[r * EW + 1] = ai[r];
            } else {
                res[r * EW] = (S)from_q_bits<TT>((TT)ar[r], qs);
                if constexpr (CPLX) res[r * EW + 1] = (S)from_q_bits<TT>((TT)ai[r], qs);
            }
        }
        constexpr int BYTES = R * EW * (int)sizeof(S);
        S *op = out + o0 * EW;
        if (BYTES % 16 == 0 && o0 + R <= n_out && (reinterpret_cast<uintptr_t>(op) & 15) == 0) {
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            u32x4 pk[BYTES / 16 > 0 ? BYTES / 16 : 1];
            __builtin_memcpy(pk, res, BYTES);
#pragma unroll
            for (int i = 0; i < BYTES / 16; i++) reinterpret_cast<u32x4 *>(op)[i] = pk[i];
        } else {
#pragma unroll
            for (int r = 0; r < R; r++) {
                if (o0 + r >= n_out) break;
#pragma unroll
                for (int c = 0; c < EW; c++) op[r * EW + c] = res[r * EW + c];
            }
        }
        if constexpr (FLT && EXACT && CPLX && CTAPS) {
            // outputs that came out NaN + i NaN, once more through the reference's complex multiply with its slow path
            // (fir_output_annex_g), one at a time and straight to memory over what was just stored: nothing of the hot
            // loop is live here, so the cold path costs the kernel no registers
            if (__builtin_expect(badmask != 0, 0)) {
#pragma unroll 1
                for (int r = 0; r < R; r++) {
                    if (!((badmask >> r) & 1u) || o0 + r >= n_out) continue;
                    const CPair<S> fx = fir_output_annex_g<S>(reinterpret_cast<const S *>(in) + (o0 + r + K - 1) * 2, reinterpret_cast<const S *>(taps), K);
                    out[(o0 + r) * 2] = fx.re;
                    out[(o0 + r) * 2 + 1] = fx.im;
                }
            }
        }
    }
}

template <typename S, typename TT, bool EXACT, bool MAD24>
static int launch_fir_slide_t(int is_complex, int complex_taps, const FirGeom &g, const void *in, void *out, size_t n_out, QShift qs, hipStream_t st)
{
    if (n_out == 0) return PCX_OK;
    constexpr int R = sizeof(S) >= 8 ? 4 : 8;
    size_t gsz = (n_out + 256 * R - 1) / (256 * R);
    if (gsz > (1u << 20)) gsz = 1u << 20;
    const unsigned grid = (unsigned)gsz;
    const S *pin = (const S *)in;
    S *pout = (S *)out;
    const TT *tp = (const TT *)g.rowTaps;
    // staged tile image: (256 R + K - 1) elements, padded by one per R; PCX_FIR_STAGE=0 reads global memory directly (A/B)
    const int stage_on = (int)PCX_ENV_INT("PCX_FIR_STAGE", 1);
    const size_t n_stage = (size_t)256 * R + g.K - 1;
    const size_t lds = (n_stage + n_stage / R + 1) * sizeof(S) * (is_complex ? 2 : 1);
    const bool stage = stage_on && lds <= 64 * 1024;
#define PCX_FIR_LAUNCH(CP, CT)                                                                                                          \
    do {                                                                                                                                \
        if (stage) hipLaunchKernelGGL((fir_slide_kernel<S, TT, CP, CT, EXACT, MAD24, R, true>), dim3(grid), dim3(256), lds, st, pin, pout, n_out, g.K, tp, qs); \
        else hipLaunchKernelGGL((fir_slide_kernel<S, TT, CP, CT, EXACT, MAD24, R, false>), dim3(grid), dim3(256), 0, st, pin, pout, n_out, g.K, tp, qs); \
    } while (0)
    if (!is_complex) PCX_FIR_LAUNCH(false, false);
    else if (!complex_taps) PCX_FIR_LAUNCH(true, false);
    else PCX_FIR_LAUNCH(true, true);
#undef PCX_FIR_LAUNCH
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}

// --------------------------------------------------------------------------------- //
// complex_int16 (and complex_int8) stream, COMPLEX taps whose Q16.16 (Q8.8) image fits 16 bits (|tap| < 0.5: the usual case for
// a unity-gain filter): one sample is one dword (re, im) and a complex multiply-accumulate is two packed
// dot products,  ar += (c, d).(a, -b),  ai += (c, d).(b, a)  (v_dot2_i32_i16, wrapping like the reference's
// complex<int32> accumulator, FIRFilter.cpp:296-300 with QType = int32).  Same LDS-staged sliding window
// as fir_slide_kernel, half its multiply instructions and half its window registers.
// tapsP[2k] = pack(a, -b), tapsP[2k+1] = pack(b, a) (low half first), built on the host.
// --------------------------------------------------------------------------------- //
// IN8: complex_int8 stream (QType = int16, Q8.8 taps always fit 16 bits): samples are widened to 16-bit pairs
// while the tile is staged, the int32 dot products carry the reference's int16 accumulator in their low half.
typedef short s16x2 __attribute__((ext_vector_type(2)));
template <bool IN8>
__global__ __launch_bounds__(256) void fir_ci16_dot2_kernel(const void *__restrict__ in_v, void *__restrict__ out_v, size_t n_out, size_t K,
                                                            const uint32_t *__restrict__ tapsP, QShift qs)
{
    const uint32_t *in = static_cast<const uint32_t *>(in_v);
    uint32_t *out = static_cast<uint32_t *>(out_v);
    constexpr int R = 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint32_t *img = reinterpret_cast<uint32_t *>(smem_raw);
    const size_t gstride = (size_t)gridDim.x * blockDim.x * R;
    const size_t last = n_out - 1;
    auto as_v = [](uint32_t u) { s16x2 v; __builtin_memcpy(&v, &u, 4); return v; };
    for (size_t o0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * R; o0 - (size_t)threadIdx.x * R < n_out; o0 += gstride) {
        const size_t tile0 = o0 - (size_t)threadIdx.x * R;
        const size_t n_stage = (size_t)blockDim.x * R + K - 1;
        __syncthreads();
        for (size_t i = threadIdx.x; i < n_stage; i += blockDim.x) {
            const size_t g = tile0 + i < last + K ? tile0 + i : last + K - 1;
            if (IN8) {
                const uint16_t v = static_cast<const uint16_t *>(in_v)[g];
                const int re = (int8_t)(v & 0xff), im = (int8_t)(v >> 8);
                img[i + i / R] = (uint32_t)(uint16_t)(int16_t)re | ((uint32_t)(uint16_t)(int16_t)im << 16);
            } else {
                img[i + i / R] = in[g];
            }
        }
        __syncthreads();
        int ar[R], ai[R];
        uint32_t w[R];
        const size_t l0 = (size_t)threadIdx.x * R;     // tile-local index of this lane's first output
#pragma unroll
        for (int r = 0; r < R; r++) { ar[r] = 0; ai[r] = 0; const size_t i = l0 + r + K - 1; w[r] = img[i + i / R]; }
        for (size_t kb = 0; kb < K; kb += R) {
            uint32_t nw[R];
#pragma unroll
            for (int u = 0; u < R; u++) {
                const size_t k = kb + u;
                if (k + 1 < K) { const size_t i = l0 + K - 2 - k; nw[u] = img[i + i / R]; }
                else nw[u] = 0;
            }
#pragma unroll
            for (int u = 0; u < R; u++) {
                const size_t k = kb + u;
                if (k >= K) break;
                const s16x2 t1 = as_v(tapsP[2 * k]), t2 = as_v(tapsP[2 * k + 1]);
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const s16x2 x = as_v(w[(r - u + R) % R]);
                    ar[r] = __builtin_amdgcn_sdot2(x, t1, ar[r], false);
                    ai[r] = __builtin_amdgcn_sdot2(x, t2, ai[r], false);
                }
                w[(R - 1 - u + R) % R] = nw[u];
            }
        }
        if (IN8) {
            // QType int16: the accumulator is the low half of the int32 sum; fromQ (>> 8 by default), truncated to int8
            uint16_t res[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const uint32_t re = (uint8_t)(int8_t)from_q_bits<int16_t>((int16_t)ar[r], qs), im = (uint8_t)(int8_t)from_q_bits<int16_t>((int16_t)ai[r], qs);
                res[r] = (uint16_t)(re | (im << 8));
            }
            uint16_t *op = static_cast<uint16_t *>(out_v) + o0;
            if (o0 + R <= n_out && (reinterpret_cast<uintptr_t>(op) & 15) == 0) {
                typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                u32x4 pk;
                __builtin_memcpy(&pk, res, 16);
                *reinterpret_cast<u32x4 *>(op) = pk;
            } else {
#pragma unroll
                for (int r = 0; r < R; r++) { if (o0 + r >= n_out) break; op[r] = res[r]; }
            }
        } else {
            // fromQ of the wrapped int32 (arithmetic >> 16 by default), truncated to int16 (FIRFilter.cpp:300)
            uint32_t res[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const uint32_t re = (uint32_t)(uint16_t)(int16_t)from_q_bits<int32_t>(ar[r], qs), im = (uint32_t)(uint16_t)(int16_t)from_q_bits<int32_t>(ai[r], qs);
                res[r] = re | (im << 16);
            }
            uint32_t *op = out + o0;
            if (o0 + R <= n_out && (reinterpret_cast<uintptr_t>(op) & 15) == 0) {
                typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                reinterpret_cast<u32x4 *>(op)[0] = u32x4{res[0], res[1], res[2], res[3]};
                reinterpret_cast<u32x4 *>(op)[1] = u32x4{res[4], res[5], res[6], res[7]};
            } else {
#pragma unroll
                for (int r = 0; r < R; r++) { if (o0 + r >= n_out) break; op[r] = res[r]; }
            }
        }
    }
}
int launch_fir_ci16_dot2(const void *in, void *out, size_t n_out, size_t K, const void *tapsP, bool in8, QShift qs, hipStream_t st)
{
    if (n_out == 0) return PCX_OK;
    constexpr int R = 8;
    const size_t n_stage = (size_t)256 * R + K - 1;
    const size_t lds = (n_stage + n_stage / R + 1) * sizeof(uint32_t);
    if (lds > 64 * 1024) { set_error("fir (int16 dot2): %zu taps exceed the LDS tile", K); return PCX_ERR_UNSUPPORTED; }
    size_t gsz = (n_out + 256 * R - 1) / (256 * R);
    if (gsz > (1u << 20)) gsz = 1u << 20;
    if (in8) hipLaunchKernelGGL(fir_ci16_dot2_kernel<true>, dim3((unsigned)gsz), dim3(256), lds, st, in, out, n_out, K, (const uint32_t *)tapsP, qs);
    else hipLaunchKernelGGL(fir_ci16_dot2_kernel<false>, dim3((unsigned)gsz), dim3(256), lds, st, in, out, n_out, K, (const uint32_t *)tapsP, qs);
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}

// M = L = 1 entry: `taps24` = every Q tap fits 24 signed bits (int16 / int8 element types)
int launch_fir_slide(int scalar, int is_complex, int complex_taps, bool exact, bool taps24, const FirGeom &g, const void *in,
                     void *out, size_t n_out, QShift qs, hipStream_t st)
{
    switch (scalar) {
    case PCX_F32:
        return exact ? launch_fir_slide_t<float, float, true, false>(is_complex, complex_taps, g, in, out, n_out, qs, st)
                     : launch_fir_slide_t<float, float, false, false>(is_complex, complex_taps, g, in, out, n_out, qs, st);
    case PCX_F64:
        return exact ? launch_fir_slide_t<double, double, true, false>(is_complex, complex_taps, g, in, out, n_out, qs, st)
                     : launch_fir_slide_t<double, double, false, false>(is_complex, complex_taps, g, in, out, n_out, qs, st);
    case PCX_I64: return launch_fir_slide_t<int64_t, int64_t, true, false>(is_complex, complex_taps, g, in, out, n_out, qs, st);
    case PCX_I32: return launch_fir_slide_t<int32_t, int64_t, true, false>(is_complex, complex_taps, g, in, out, n_out, qs, st);
    case PCX_I16:
        return taps24 ? launch_fir_slide_t<int16_t, int32_t, true, true>(is_complex, complex_taps, g, in, out, n_out, qs, st)
                      : launch_fir_slide_t<int16_t, int32_t, true, false>(is_complex, complex_taps, g, in, out, n_out, qs, st);
    case PCX_I8: return launch_fir_slide_t<int8_t, int16_t, true, true>(is_complex, complex_taps, g, in, out, n_out, qs, st);
    }
    set_error("fir: unsupported scalar type %d", scalar);
    return PCX_ERR_ARG;
}

template <typename S, typename TT, bool EXACT>
static int launch_fir_generic_t(int is_complex, int complex_taps, const FirGeom &g, const void *in, void *out, size_t n_out, QShift qs, hipStream_t st)
{
    if (n_out == 0) return PCX_OK;
    size_t gsz = (n_out + 255) / 256;
    if (gsz > (1u << 20)) gsz = 1u << 20;
    const unsigned grid = (unsigned)gsz;
    const S *pin = (const S *)in;
    S *pout = (S *)out;
    const TT *tp = (const TT *)g.rowTaps;
#define PCX_FIR_LAUNCH(CP, CT)                                                                                           \
    hipLaunchKernelGGL((fir_generic_kernel<S, TT, CP, CT, EXACT>), dim3(grid), dim3(256), 0, st, pin, pout, n_out, g.L, \
                       g.M, g.K, g.rowLen, tp, qs)
    if (!is_complex) PCX_FIR_LAUNCH(false, false);
    else if (!complex_taps) PCX_FIR_LAUNCH(true, false);
    else PCX_FIR_LAUNCH(true, true);
#undef PCX_FIR_LAUNCH
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}

int launch_fir_generic(int scalar, int is_complex, int complex_taps, bool exact, const FirGeom &g, const void *in,
                       void *out, size_t n_out, QShift qs, hipStream_t st)
{
    switch (scalar) {
    case PCX_F32:
        return exact ? launch_fir_generic_t<float, float, true>(is_complex, complex_taps, g, in, out, n_out, qs, st)
                     : launch_fir_generic_t<float, float, false>(is_complex, complex_taps, g, in, out, n_out, qs, st);
    case PCX_F64:
        return exact ? launch_fir_generic_t<double, double, true>(is_complex, complex_taps, g, in, out, n_out, qs, st)
                     : launch_fir_generic_t<double, double, false>(is_complex, complex_taps, g, in, out, n_out, qs, st);
    case PCX_I64: return launch_fir_generic_t<int64_t, int64_t, true>(is_complex, complex_taps, g, in, out, n_out, qs, st);
    case PCX_I32: return launch_fir_generic_t<int32_t, int64_t, true>(is_complex, complex_taps, g, in, out, n_out, qs, st);
    case PCX_I16: return launch_fir_generic_t<int16_t, int32_t, true>(is_complex, complex_taps, g, in, out, n_out, qs, st);
    case PCX_I8: return launch_fir_generic_t<int8_t, int16_t, true>(is_complex, complex_taps, g, in, out, n_out, qs, st);
    }
    set_error("fir: unsupported scalar type %d", scalar);
    return PCX_ERR_ARG;
}

}  // namespace pcx
