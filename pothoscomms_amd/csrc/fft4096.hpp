// fft4096.hpp -- device building blocks of the 4096-point complex_float32 Stockham
// transform: three radix-16 passes (4096 = 16^3), one frame per 256-lane workgroup,
// 16 points per lane held in registers, two LDS exchanges per transform.
//
// Replaces the recursive radix-4 kissfft<float>::kf_work (fft/kissfft.hh:87-161) for
// numBins = 4096 (six radix-4 passes there).  Same transform definition: forward
// exp(-j2pi nk/N), inverse exp(+j..), no 1/N scaling (kissfft.hh:81-84, TestFFT.cpp:79-80).
//
// Layout per pass (Stockham autosort, decimation in time), lane j = 0..255, r = 0..15:
//   pass 1 (Ns=1):   v[r] = x[j + 256 r];                 FFT16; lds[16 j + k]              = V[k]
//   pass 2 (Ns=16):  v[r] = lds[j + 256 r] * W256^(kk r); FFT16; lds[(j>>4)*256 + kk + 16k] = V[k]   (kk = j & 15)
//   pass 3 (Ns=256): v[r] = lds[j + 256 r] * W4096^(j r); FFT16; X[j + 256 k]               = V[k]
// Global loads/stores are stride-256 across r and unit-stride across lanes: every
// wave-instruction moves one contiguous 512-byte row.  The LDS image is padded by one
// element per 16 (pad(i) = i + i/16) so pass-1/2 scatter writes (16-lane groups) and the
// stride-1 gathers are bank-conflict free.
//
// Twiddles come from tables laid out [r][lane] (coalesced, L2-resident), generated on
// the host in double precision and rounded once -- more accurate than kissfft's
// float-evaluated table (kissfft.hh:21-26); results agree to ~3e-7 of max|X|.
#pragma once
#include <hip/hip_runtime.h>

namespace pcx {
namespace fft4k {

constexpr int N = 4096;
constexpr int T = 256;                 // lanes per frame
constexpr int LDS_ELEMS = N + N / 16;  // padded float2 count (34,816 bytes)

__device__ __forceinline__ int pad(int i) { return i + (i >> 4); }

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// a * (c + i*s) for the forward table entry w=(c,s); the inverse uses conj(w)
template <bool INV>
__device__ __forceinline__ float2 cmul_tw(float2 a, float2 w)
{
    if (INV) return make_float2(__builtin_fmaf(a.x, w.x, a.y * w.y), __builtin_fmaf(a.y, w.x, -a.x * w.y));
    return make_float2(__builtin_fmaf(a.x, w.x, -a.y * w.y), __builtin_fmaf(a.y, w.x, a.x * w.y));
}
__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    return make_float2(__builtin_fmaf(a.x, b.x, -a.y * b.y), __builtin_fmaf(a.x, b.y, a.y * b.x));
}
// multiply by the radix-4 unit twiddle: -i (forward) / +i (inverse)
template <bool INV>
__device__ __forceinline__ float2 mul_unit(float2 a)
{
    return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}
// a * exp(-/+ i*theta) with (c, s) = (cos theta, sin theta) compile-time constants
template <bool INV>
__device__ __forceinline__ float2 cmul_cs(float2 a, float c, float s)
{
    if (INV) return make_float2(__builtin_fmaf(a.x, c, -a.y * s), __builtin_fmaf(a.y, c, a.x * s));
    return make_float2(__builtin_fmaf(a.x, c, a.y * s), __builtin_fmaf(a.y, c, -a.x * s));
}

template <bool INV>
__device__ __forceinline__ void fft4(float2 &a0, float2 &a1, float2 &a2, float2 &a3)
{
    const float2 t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = mul_unit<INV>(csub(a1, a3));
    a0 = cadd(t0, t2);
    a1 = cadd(t1, t3);
    a2 = csub(t0, t2);
    a3 = csub(t1, t3);
}

// 16-point DFT in registers.  Input x[n] at v[n]; output X[k] at v[4*(k & 3) + (k >> 2)].
template <bool INV>
__device__ __forceinline__ void fft16(float2 (&v)[16])
{
    constexpr float C1 = 0.92387953251128673848f;  // cos(pi/8)
    constexpr float S1 = 0.38268343236508978178f;  // sin(pi/8)
    constexpr float R2 = 0.70710678118654752440f;  // cos(pi/4)
    // inner DFT4 over n1 for each n2: x[4 n1 + n2] -> y[n2][k1] stored at v[4 k1 + n2]
#pragma unroll
    for (int n2 = 0; n2 < 4; n2++) fft4<INV>(v[n2], v[4 + n2], v[8 + n2], v[12 + n2]);
    // twiddle y[n2][k1] *= W16^(n2 k1)
    v[4 * 1 + 1] = cmul_cs<INV>(v[4 * 1 + 1], C1, S1);    // e = 1
    v[4 * 1 + 2] = cmul_cs<INV>(v[4 * 1 + 2], R2, R2);    // e = 2
    v[4 * 1 + 3] = cmul_cs<INV>(v[4 * 1 + 3], S1, C1);    // e = 3
    v[4 * 2 + 1] = cmul_cs<INV>(v[4 * 2 + 1], R2, R2);    // e = 2
    v[4 * 2 + 2] = mul_unit<INV>(v[4 * 2 + 2]);           // e = 4
    v[4 * 2 + 3] = cmul_cs<INV>(v[4 * 2 + 3], -R2, R2);   // e = 6
    v[4 * 3 + 1] = cmul_cs<INV>(v[4 * 3 + 1], S1, C1);    // e = 3
    v[4 * 3 + 2] = cmul_cs<INV>(v[4 * 3 + 2], -R2, R2);   // e = 6
    v[4 * 3 + 3] = cmul_cs<INV>(v[4 * 3 + 3], -C1, -S1);  // e = 9
    // outer DFT4 over n2 for each k1: -> X[k1 + 4 k2] at v[4 k1 + k2]
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++) fft4<INV>(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
}
// register index q holds output bin k = bin_of(q)
__device__ __forceinline__ constexpr int bin_of(int q) { return (q >> 2) + 4 * (q & 3); }

// twiddle tables (device global memory, forward sign):
//   tw2[r*16  + kk] = exp(-j 2pi kk r / 256),  r < 16, kk < 16
//   tw3[r*256 + j ] = exp(-j 2pi j  r / 4096), r < 16, j  < 256
struct Tables {
    const float2 *tw2;
    const float2 *tw3;
};

// pass 1: v[r] = x[j + 256 r] on entry; leaves the pass-1 result in LDS
template <bool INV>
__device__ __forceinline__ void pass1(float2 (&v)[16], float2 *lds, int j)
{
    fft16<INV>(v);
    __syncthreads();  // previous readers of this LDS image are done
#pragma unroll
    for (int q = 0; q < 16; q++) lds[17 * j + bin_of(q)] = v[q];  // pad(16 j + k) = 17 j + k
}
template <bool INV>
__device__ __forceinline__ void pass2(float2 (&v)[16], float2 *lds, int j, const Tables &tb)
{
    const int kk = j & 15;
    float2 w[16];
#pragma unroll
    for (int r = 1; r < 16; r++) w[r] = tb.tw2[r * 16 + kk];
    __syncthreads();
    const int rb = j + (j >> 4);  // pad(j + 256 r) = rb + 272 r
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = lds[rb + 272 * r];
#pragma unroll
    for (int r = 1; r < 16; r++) v[r] = cmul_tw<INV>(v[r], w[r]);
    fft16<INV>(v);
    __syncthreads();
    const int wb = (j >> 4) * 272 + kk;  // pad((j>>4)*256 + kk + 16 k) = wb + 17 k
#pragma unroll
    for (int q = 0; q < 16; q++) lds[wb + 17 * bin_of(q)] = v[q];
}
// pass 3: on exit v[q] = X[j + 256 * bin_of(q)]
template <bool INV>
__device__ __forceinline__ void pass3(float2 (&v)[16], const float2 *lds, int j, const Tables &tb)
{
    float2 w[16];
#pragma unroll
    for (int r = 1; r < 16; r++) w[r] = tb.tw3[r * 256 + j];
    __syncthreads();
    const int rb = j + (j >> 4);
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = lds[rb + 272 * r];
#pragma unroll
    for (int r = 1; r < 16; r++) v[r] = cmul_tw<INV>(v[r], w[r]);
    fft16<INV>(v);
}

}  // namespace fft4k
}  // namespace pcx
