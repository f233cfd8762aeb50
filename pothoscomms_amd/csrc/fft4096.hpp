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
// wave-instruction moves one contiguous 512-byte row (measured: this pattern streams at the
// same 5.5 TB/s as a float4 copy, tools/ubench.hip).  The LDS image is padded by one
// element per 16 (pad(i) = i + i/16) so pass-1/2 scatter writes (16-lane groups) and the
// stride-1 gathers are bank-conflict free.
//
// Twiddles.  A lane's pass-2/pass-3 twiddles are powers w^r (r = 0..15) of ONE base that
// depends only on the lane (w = W256^(j&15), resp. W4096^j) -- not on the frame.  Writing
// r = 4 n1 + n2, w^r = (w^4)^n1 * w^n2: the factor (w^4)^n1 is applied to the inputs, the
// factor w^n2 after the inner DFT4 (it is common to the four inputs of that DFT4).  So a
// lane needs only w, w^2, w^3, w^4, w^8, w^12 per pass: 12 values, loaded ONCE per kernel
// from a [12][256] table (host-generated in double precision, rounded once -- more accurate
// than kissfft's float-evaluated table, kissfft.hh:21-26) and kept in registers while the
// persistent workgroup walks its frames.  No table traffic inside the frame loop means the
// only vector-memory operations in flight are the stream itself, so a register prefetch of
// the next frame is never stuck behind a table load in the in-order vmcnt queue.
#pragma once
#include <hip/hip_runtime.h>

namespace pcx {
namespace fft4k {

constexpr int N = 4096;
constexpr int T = 256;                 // lanes per frame
constexpr int LDS_ELEMS = N + N / 16;  // padded float2 count (34,816 bytes)
constexpr int TW_TABLE_ELEMS = 12 * 256;

// complex values are native 2-vectors so every complex add/sub is ONE v_pk_add_f32
typedef float cf __attribute__((ext_vector_type(2)));

// a * w.  Two packed instructions on the (re, im) register PAIRS as they stand: the
// operand swizzles (a.yx, w.yy, w.xx) and the sign ride on VOP3P op_sel / neg modifiers.
// Written as asm because hipcc otherwise materialises the splat / negated twiddle vectors
// in extra registers and hoists them out of the frame loop (12 lane-constant twiddles
// became ~60 live VGPRs and spilled).
__device__ __forceinline__ cf cmul(cf a, cf w)
{
    cf t, r;
    // t = (-a.y * w.y, a.x * w.y)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1] neg_lo:[0,1]" : "=v"(t) : "v"(a), "v"(w));
    // r = (a.x * w.x + t.x, a.y * w.x + t.y)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
// conj(a * h) = (a.x h.x - a.y h.y, -(a.y h.x + a.x h.y))
__device__ __forceinline__ cf cmul_conj(cf a, cf h)
{
    cf t, r;
    // t = (-a.y * h.y, -a.x * h.y)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(a), "v"(h));
    // r = (a.x * h.x + t.x, -a.y * h.x + t.y)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(h), "v"(t));
    return r;
}
// multiply by -i: (x, y) -> (y, -x)
__device__ __forceinline__ cf mul_mi(cf a) { return cf{a.y, -a.x}; }
// a * exp(-i*theta), (c, s) = (cos theta, sin theta) compile-time constants
__device__ __forceinline__ cf cmul_cs(cf a, float c, float s)
{
    return cf{__builtin_fmaf(a.x, c, a.y * s), __builtin_fmaf(a.y, c, -a.x * s)};
}

__device__ __forceinline__ void fft4(cf &a0, cf &a1, cf &a2, cf &a3)
{
    const cf t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, t3 = mul_mi(a1 - a3);
    a0 = t0 + t2;
    a1 = t1 + t3;
    a2 = t0 - t2;
    a3 = t1 - t3;
}

// per-lane twiddle powers of one pass: w[0..2] = w, w^2, w^3;  w[3..5] = w^4, w^8, w^12
struct LaneTw {
    cf w[6];
};

// Forward 16-point DFT in registers of x[n] * w^n (TW) or x[n] (no external twiddle).
// Input x[n] at v[n]; output X[k] at v[4*(k & 3) + (k >> 2)].  The inverse transform is
// taken as conj(FFT(conj(x))) by the callers, so only the forward butterfly exists.
template <bool TW>
__device__ __forceinline__ void fft16(cf (&v)[16], const LaneTw &tw)
{
    constexpr float C1 = 0.92387953251128673848f;  // cos(pi/8)
    constexpr float S1 = 0.38268343236508978178f;  // sin(pi/8)
    constexpr float R2 = 0.70710678118654752440f;  // cos(pi/4)
    if (TW) {  // (w^4)^n1 on input n = 4 n1 + n2
#pragma unroll
        for (int n1 = 1; n1 < 4; n1++)
#pragma unroll
            for (int n2 = 0; n2 < 4; n2++) v[4 * n1 + n2] = cmul(v[4 * n1 + n2], tw.w[2 + n1]);
    }
    // inner DFT4 over n1 for each n2: x[4 n1 + n2] -> y[n2][k1] stored at v[4 k1 + n2]
#pragma unroll
    for (int n2 = 0; n2 < 4; n2++) fft4(v[n2], v[4 + n2], v[8 + n2], v[12 + n2]);
    if (TW) {  // w^n2, common to the inner DFT4 of column n2
#pragma unroll
        for (int n2 = 1; n2 < 4; n2++)
#pragma unroll
            for (int k1 = 0; k1 < 4; k1++) v[4 * k1 + n2] = cmul(v[4 * k1 + n2], tw.w[n2 - 1]);
    }
    // y[n2][k1] *= W16^(n2 k1)
    v[4 * 1 + 1] = cmul_cs(v[4 * 1 + 1], C1, S1);    // e = 1
    v[4 * 1 + 2] = cmul_cs(v[4 * 1 + 2], R2, R2);    // e = 2
    v[4 * 1 + 3] = cmul_cs(v[4 * 1 + 3], S1, C1);    // e = 3
    v[4 * 2 + 1] = cmul_cs(v[4 * 2 + 1], R2, R2);    // e = 2
    v[4 * 2 + 2] = mul_mi(v[4 * 2 + 2]);             // e = 4
    v[4 * 2 + 3] = cmul_cs(v[4 * 2 + 3], -R2, R2);   // e = 6
    v[4 * 3 + 1] = cmul_cs(v[4 * 3 + 1], S1, C1);    // e = 3
    v[4 * 3 + 2] = cmul_cs(v[4 * 3 + 2], -R2, R2);   // e = 6
    v[4 * 3 + 3] = cmul_cs(v[4 * 3 + 3], -C1, -S1);  // e = 9
    // outer DFT4 over n2 for each k1: -> X[k1 + 4 k2] at v[4 k1 + k2]
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++) fft4(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
}
// register index q holds output bin k = bin_of(q)
__device__ __forceinline__ constexpr int bin_of(int q) { return (q >> 2) + 4 * (q & 3); }

// table layout (device global memory, forward sign), p = 0..5 <-> powers {1,2,3,4,8,12}:
//   tab[(p    ) * 256 + j] = exp(-j 2 pi (j & 15) * pow[p] / 256)     pass 2
//   tab[(6 + p) * 256 + j] = exp(-j 2 pi  j       * pow[p] / 4096)    pass 3
struct Twiddles {
    LaneTw p2, p3;
};
__device__ __forceinline__ void load_twiddles(Twiddles &t, const float2 *__restrict__ tab, int j)
{
    const cf *tb = reinterpret_cast<const cf *>(tab);
#pragma unroll
    for (int p = 0; p < 6; p++) {
        t.p2.w[p] = tb[p * 256 + j];
        t.p3.w[p] = tb[(6 + p) * 256 + j];
    }
}

// ---- stream access through buffer descriptors (SRSRC) ----
// One 32-bit lane offset (j*8) serves all 16 row accesses of a frame: the row offset
// 2048*r rides in the scalar soffset operand and the 64-bit base lives in the descriptor,
// so the frame's loads and stores cost one address VGPR instead of eight 64-bit pairs.
// The hardware range check (voffset + imm >= num_records -> load 0 / drop the store) is
// what handles the ragged last block: no per-element predicates in the kernel.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, unsigned bytes)
{
    // the base is wave-uniform by construction; readfirstlane makes that provable to the
    // compiler (otherwise every buffer op is wrapped in a waterfall loop)
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo), 0,
                                             (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
// v[r] = frame[j + 256 r].  CHECKED: the whole offset goes through voffset so rows beyond
// num_records read as zero (the range check does not see soffset).
template <bool CHECKED>
__device__ __forceinline__ void load_frame(cf (&v)[16], __amdgpu_buffer_rsrc_t rs, int j)
{
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const u32x2 t = CHECKED ? __builtin_amdgcn_raw_buffer_load_b64(rs, (j + 256 * r) * 8, 0, 0)
                                : __builtin_amdgcn_raw_buffer_load_b64(rs, j * 8, 2048 * r, 0);
        v[r] = cf{__uint_as_float(t.x), __uint_as_float(t.y)};
    }
}
__device__ __forceinline__ void store_cf(__amdgpu_buffer_rsrc_t rs, unsigned voff, cf a)
{
    u32x2 t;
    t.x = __float_as_uint(a.x);
    t.y = __float_as_uint(a.y);
    __builtin_amdgcn_raw_buffer_store_b64(t, rs, (int)voff, 0, 0);
}

// pass 1: v[r] = x[j + 256 r] on entry; leaves the pass-1 result in LDS
__device__ __forceinline__ void pass1(cf (&v)[16], cf *lds, int j, const Twiddles &t)
{
    fft16<false>(v, t.p2);
    __syncthreads();  // previous readers of this LDS image are done
#pragma unroll
    for (int q = 0; q < 16; q++) lds[17 * j + bin_of(q)] = v[q];  // pad(16 j + k) = 17 j + k
}
__device__ __forceinline__ void pass2(cf (&v)[16], cf *lds, int j, const Twiddles &t)
{
    __syncthreads();
    const int rb = j + (j >> 4);  // pad(j + 256 r) = rb + 272 r
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = lds[rb + 272 * r];
    fft16<true>(v, t.p2);
    __syncthreads();
    const int wb = (j >> 4) * 272 + (j & 15);  // pad((j>>4)*256 + kk + 16 k) = wb + 17 k
#pragma unroll
    for (int q = 0; q < 16; q++) lds[wb + 17 * bin_of(q)] = v[q];
}
// pass 3: on exit v[q] = X[j + 256 * bin_of(q)]
__device__ __forceinline__ void pass3(cf (&v)[16], const cf *lds, int j, const Twiddles &t)
{
    __syncthreads();
    const int rb = j + (j >> 4);
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = lds[rb + 272 * r];
    fft16<true>(v, t.p3);
}

}  // namespace fft4k
}  // namespace pcx
