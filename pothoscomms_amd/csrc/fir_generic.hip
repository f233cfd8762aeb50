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
#include "pcx_internal.hpp"

#include <type_traits>

namespace pcx {

__device__ __forceinline__ float t_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double t_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

template <typename Q>
struct QComp {
    using type = typename std::conditional<(sizeof(Q) < 4), uint32_t, typename std::make_unsigned<Q>::type>::type;
};

// S = element scalar, TT = stored tap scalar (float/double or Q int), CPLX/CTAPS flags
template <typename S, typename TT, bool CPLX, bool CTAPS, bool EXACT>
__global__ __launch_bounds__(256) void fir_generic_kernel(const S *__restrict__ in, S *__restrict__ out, size_t n_out,
                                                          size_t L, size_t M, size_t K,
                                                          const uint32_t *__restrict__ rowLen, const TT *__restrict__ rowTaps)
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
            out[o * EW] = (S)(((TT)ar) >> (4 * sizeof(TT)));
            if constexpr (CPLX) out[o * EW + 1] = (S)(((TT)ai) >> (4 * sizeof(TT)));
        }
    }
}

template <typename S, typename TT, bool EXACT>
static int launch_fir_generic_t(int is_complex, int complex_taps, const FirGeom &g, const void *in, void *out, size_t n_out, hipStream_t st)
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
                       g.M, g.K, g.rowLen, tp)
    if (!is_complex) PCX_FIR_LAUNCH(false, false);
    else if (!complex_taps) PCX_FIR_LAUNCH(true, false);
    else PCX_FIR_LAUNCH(true, true);
#undef PCX_FIR_LAUNCH
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}

int launch_fir_generic(int scalar, int is_complex, int complex_taps, bool exact, const FirGeom &g, const void *in,
                       void *out, size_t n_out, hipStream_t st)
{
    switch (scalar) {
    case PCX_F32:
        return exact ? launch_fir_generic_t<float, float, true>(is_complex, complex_taps, g, in, out, n_out, st)
                     : launch_fir_generic_t<float, float, false>(is_complex, complex_taps, g, in, out, n_out, st);
    case PCX_F64:
        return exact ? launch_fir_generic_t<double, double, true>(is_complex, complex_taps, g, in, out, n_out, st)
                     : launch_fir_generic_t<double, double, false>(is_complex, complex_taps, g, in, out, n_out, st);
    case PCX_I64: return launch_fir_generic_t<int64_t, int64_t, true>(is_complex, complex_taps, g, in, out, n_out, st);
    case PCX_I32: return launch_fir_generic_t<int32_t, int64_t, true>(is_complex, complex_taps, g, in, out, n_out, st);
    case PCX_I16: return launch_fir_generic_t<int16_t, int32_t, true>(is_complex, complex_taps, g, in, out, n_out, st);
    case PCX_I8: return launch_fir_generic_t<int8_t, int16_t, true>(is_complex, complex_taps, g, in, out, n_out, st);
    }
    set_error("fir: unsupported scalar type %d", scalar);
    return PCX_ERR_ARG;
}

}  // namespace pcx
