// fft_f64.hpp -- double-precision radix-16 butterfly primitives shared by the complex_float64 FFT
// (fft_r16_f64.hip) and overlap-save FIR (fir_ols_f64.hip).  Same factorisation as fft4096.hpp (16 = 4 x 4,
// the lane's external twiddle merged into 15 per-lane factors); plain v_fma_f64 arithmetic.
#pragma once
#include <hip/hip_runtime.h>

namespace pcx {
namespace fft64 {

typedef double cd __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define DI __device__ __forceinline__

DI cd cmul(cd a, cd w) { return cd{__builtin_fma(a.x, w.x, -(a.y * w.y)), __builtin_fma(a.x, w.y, a.y * w.x)}; }
// a * exp(-i theta), (c, s) = (cos theta, sin theta)
DI cd cmul_cs(cd a, double c, double s) { return cd{__builtin_fma(a.x, c, a.y * s), __builtin_fma(a.y, c, -(a.x * s))}; }
DI cd mul_mi(cd a) { return cd{a.y, -a.x}; }   // a * (-i)
DI cd conj_if(bool c, cd a) { return c ? cd{a.x, -a.y} : a; }

DI void fft4(cd &a0, cd &a1, cd &a2, cd &a3)
{
    const cd t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, d = mul_mi(a1 - a3);
    a0 = t0 + t2;
    a2 = t0 - t2;
    a1 = t1 + d;
    a3 = t1 - d;
}
// forward DFT4 of (a0, a1, -i*a2, a3)
DI void fft4_mi2(cd &a0, cd &a1, cd &a2, cd &a3)
{
    const cd r = mul_mi(a2);
    const cd t0 = a0 + r, t1 = a0 - r, t2 = a1 + a3, d = mul_mi(a1 - a3);
    a0 = t0 + t2;
    a2 = t0 - t2;
    a1 = t1 + d;
    a3 = t1 - d;
}
DI void fft16_inner(cd (&v)[16])
{
#pragma unroll
    for (int n2 = 0; n2 < 4; n2++) fft4(v[n2], v[4 + n2], v[8 + n2], v[12 + n2]);
}
DI void fft16_outer(cd (&v)[16])
{
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++) fft4(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
}
// 16-point forward DFT, x[n] at v[n]; X[k] lands at v[4*(k&3) + (k>>2)]
DI void fft16_plain(cd (&v)[16])
{
    constexpr double C1 = 0.92387953251128673848313610506;   // cos(pi/8)
    constexpr double S1 = 0.38268343236508977172845998403;   // sin(pi/8)
    constexpr double R2 = 0.70710678118654752440084436210;   // cos(pi/4)
    fft16_inner(v);
    v[4 * 1 + 1] = cmul_cs(v[4 * 1 + 1], C1, S1);
    v[4 * 1 + 2] = cmul_cs(v[4 * 1 + 2], R2, R2);
    v[4 * 1 + 3] = cmul_cs(v[4 * 1 + 3], S1, C1);
    v[4 * 2 + 1] = cmul_cs(v[4 * 2 + 1], R2, R2);
    v[4 * 2 + 3] = cmul_cs(v[4 * 2 + 3], -R2, R2);
    v[4 * 3 + 1] = cmul_cs(v[4 * 3 + 1], S1, C1);
    v[4 * 3 + 2] = cmul_cs(v[4 * 3 + 2], -R2, R2);
    v[4 * 3 + 3] = cmul_cs(v[4 * 3 + 3], -C1, -S1);
    fft4(v[0], v[1], v[2], v[3]);
    fft4(v[4], v[5], v[6], v[7]);
    fft4_mi2(v[8], v[9], v[10], v[11]);
    fft4(v[12], v[13], v[14], v[15]);
}
// 16-point forward DFT of x[n] * w^n; the lane's 15 factors (layout of make_tw_r16: 3 x (w^4)^n1, then
// c[(n2-1)*4 + k1] = w^n2 W16^(n2 k1)) are fetched through `tw(p)`, p = 0..14, at their use
template <typename TW>
DI void fft16_tw(cd (&v)[16], TW tw)
{
#pragma unroll
    for (int n1 = 1; n1 < 4; n1++) {
        const cd w = tw(n1 - 1);
#pragma unroll
        for (int n2 = 0; n2 < 4; n2++) v[4 * n1 + n2] = cmul(v[4 * n1 + n2], w);
    }
    fft16_inner(v);
#pragma unroll
    for (int n2 = 1; n2 < 4; n2++)
#pragma unroll
        for (int k1 = 0; k1 < 4; k1++) v[4 * k1 + n2] = cmul(v[4 * k1 + n2], tw(3 + (n2 - 1) * 4 + k1));
    fft16_outer(v);
}
DI constexpr int bin_of(int q) { return (q >> 2) + 4 * (q & 3); }
DI int padi(int i) { return i + (i >> 4); }

DI void fft8(cd &a0, cd &a1, cd &a2, cd &a3, cd &a4, cd &a5, cd &a6, cd &a7)
{
    constexpr double R2 = 0.70710678118654752440084436210;
    cd e0 = a0, e1 = a2, e2 = a4, e3 = a6, o0 = a1, o1 = a3, o2 = a5, o3 = a7;
    fft4(e0, e1, e2, e3);
    fft4(o0, o1, o2, o3);
    const cd w1 = cmul_cs(o1, R2, R2);
    const cd w2 = mul_mi(o2);
    const cd w3 = cmul_cs(o3, -R2, R2);
    a0 = e0 + o0; a4 = e0 - o0;
    a1 = e1 + w1; a5 = e1 - w1;
    a2 = e2 + w2; a6 = e2 - w2;
    a3 = e3 + w3; a7 = e3 - w3;
}

DI __amdgpu_buffer_rsrc_t make_rsrc(const void *p, unsigned bytes)
{
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo), 0,
                                             (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
constexpr int kAuxStream = 2;   // non-temporal: the stream is touched once
DI cd as_cd(u32x4 t) { return __builtin_bit_cast(cd, t); }
DI u32x4 as_u4(cd a) { return __builtin_bit_cast(u32x4, a); }

#undef DI
}  // namespace fft64
}  // namespace pcx
