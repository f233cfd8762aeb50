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
// element per 16 (pad(i) = i + i/16): the pass-1/2 scatter writes (ds_write_b64: 16-lane groups
// over 32 banks) are conflict-free; the stride-1 gathers are NOT quite -- ds_read_b64 serves
// 32-lane groups over 64 banks, the pad puts 33 elements under 32 lanes, lane 31 lands on lane
// 0's banks and each group takes two LDS cycles instead of one (SQ_LDS_BANK_CONFLICT = 20 % of
// SQ_LDS_IDX_ACTIVE).  A layout that is conflict-free for both, s(e) = (e & ~15) | ((e ^ (e >> 4)) & 15)
// in exactly 32 KiB, was built and measured (tools/ols_lab3.hip SWZ, profiles/r03/ols_lab3.md):
// its scatter addresses are no longer affine in the register index -- one v_xor per store,
// 64 VALU instructions per block on a kernel that is short of VALU issue slots, not of LDS
// cycles -- and it ran 0.7-1.2 % SLOWER.  The padded image stays.
//
// Twiddles.  The radix-16 butterfly is two layers of DFT4 (n = 4 n1 + n2, k = k1 + 4 k2).
// A lane's external twiddle w^n (w = W256^(j&15) in pass 2, W4096^j in pass 3) factors as
// (w^4)^n1 * w^n2: the first factor goes on the inputs of the inner DFT4 (3 distinct
// values), the second merges with the butterfly's own W16^(n2 k1) into ONE per-lane factor
//   c[n2][k1] = w^n2 * W16^(n2 k1)           (12 values)
// applied between the layers -- 24 complex multiplies per pass, the count of a plain
// radix-4 FFT, and no separate constant-twiddle step.  All 15 values depend only on the
// lane, never on the frame, so the persistent workgroup keeps the pass-3 set in registers
// (30 VGPRs) and the pass-2 set (a function of j & 15 only: 15 x 16 entries) in 1.9 KB of
// LDS.  Nothing but the sample stream is fetched from global memory inside the frame loop,
// so a register prefetch of the next frame never queues behind a table load in the
// in-order vmcnt counter.  Tables are generated on the host in double precision and
// rounded once (more accurate than kissfft's float-evaluated table, kissfft.hh:21-26).
//
// Arithmetic is packed (v_pk_*_f32 on (re, im) register pairs).  The complex multiply and
// the +-i rotations are written as asm with op_sel / neg modifiers: hipcc otherwise
// materialises swizzled and negated copies of the lane-constant twiddles in extra
// registers (it spilled) and spends a v_xor + v_mov per rotation.
#pragma once
#include <hip/hip_runtime.h>

namespace pcx {
namespace fft4k {

constexpr int N = 4096;
constexpr int T = 256;                    // lanes per frame
constexpr int LDS_DATA = N + N / 16;      // padded cf count of the frame image (34,816 B)
constexpr int LDS_TW2 = 15 * 16;          // pass-2 twiddle table (1,920 B)
constexpr int LDS_ELEMS = LDS_DATA + LDS_TW2;
constexpr int TW_TABLE_ELEMS = 15 * 16 + 15 * 256;   // device table: pass-2 block, then pass-3 block

typedef float cf __attribute__((ext_vector_type(2)));   // complex: one v_pk_add_f32 per add

// (a, b) <- (a * wa, b * wb): 4 packed instructions, two independent chains interleaved
__device__ __forceinline__ void cmul2(cf &a, cf &b, cf wa, cf wb)
{
    cf ra, rb;
    asm("v_pk_mul_f32 %0, %2, %4 op_sel:[1,1] op_sel_hi:[0,1] neg_lo:[0,1]\n\t"   // (-a.y w.y, a.x w.y)
        "v_pk_mul_f32 %1, %3, %5 op_sel:[1,1] op_sel_hi:[0,1] neg_lo:[0,1]\n\t"
        "v_pk_fma_f32 %0, %2, %4, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"        // (a.x w.x, a.y w.x) + t
        "v_pk_fma_f32 %1, %3, %5, %1 op_sel:[0,0,0] op_sel_hi:[1,0,1]"
        : "=&v"(ra), "=&v"(rb)
        : "v"(a), "v"(b), "v"(wa), "v"(wb));
    a = ra;
    b = rb;
}
// (a, b) <- (conj(a * ha), conj(b * hb))
__device__ __forceinline__ void cmul2_conj(cf &a, cf &b, cf ha, cf hb)
{
    cf ra, rb;
    asm("v_pk_mul_f32 %0, %2, %4 op_sel:[1,1] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"   // (-a.y h.y, -a.x h.y)
        "v_pk_mul_f32 %1, %3, %5 op_sel:[1,1] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %2, %4, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]\n\t"      // (a.x h.x, -a.y h.x) + t
        "v_pk_fma_f32 %1, %3, %5, %1 op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]"
        : "=&v"(ra), "=&v"(rb)
        : "v"(a), "v"(b), "v"(ha), "v"(hb));
    a = ra;
    b = rb;
}
// a + (-i) d = (a.x + d.y, a.y - d.x)   and   a - (-i) d = (a.x - d.y, a.y + d.x)
__device__ __forceinline__ void addsub_mi(cf &p, cf &m, cf a, cf d)
{
    asm("v_pk_add_f32 %0, %2, %3 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %1, %2, %3 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]"
        : "=&v"(p), "=&v"(m)
        : "v"(a), "v"(d));
}
// a * exp(-i*theta), (c, s) = (cos theta, sin theta) compile-time constants
__device__ __forceinline__ cf cmul_cs(cf a, float c, float s)
{
    return cf{__builtin_fmaf(a.x, c, a.y * s), __builtin_fmaf(a.y, c, -a.x * s)};
}

// forward DFT4 in place
__device__ __forceinline__ void fft4(cf &a0, cf &a1, cf &a2, cf &a3)
{
    const cf t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, d = a1 - a3;
    a0 = t0 + t2;
    a2 = t0 - t2;
    addsub_mi(a1, a3, t1, d);
}
// forward DFT4 of (a0, a1, -i*a2, a3): the W16^4 = -i twiddle of the untwiddled butterfly
__device__ __forceinline__ void fft4_mi2(cf &a0, cf &a1, cf &a2, cf &a3)
{
    cf t0, t1;
    addsub_mi(t0, t1, a0, a2);
    const cf t2 = a1 + a3, d = a1 - a3;
    a0 = t0 + t2;
    a2 = t0 - t2;
    addsub_mi(a1, a3, t1, d);
}

// forward DFT4 with CONJUGATED outputs: the negations ride on the last four additions (neg_lo / neg_hi modifiers), so a kernel
// that wants conj(FFT(..)) -- the inverse transform computed on the forward passes -- pays nothing for the final conjugation
__device__ __forceinline__ void fft4_conj(cf &a0, cf &a1, cf &a2, cf &a3)
{
    const cf t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, d = a1 - a3;
    asm("v_pk_add_f32 %0, %4, %5 neg_hi:[1,1]\n\t"                                        // ( t0.x + t2.x, -t0.y - t2.y)
        "v_pk_add_f32 %1, %4, %5 neg_lo:[0,1] neg_hi:[1,0]\n\t"                           // ( t0.x - t2.x, -t0.y + t2.y)
        "v_pk_add_f32 %2, %6, %7 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[1,0]\n\t"           // ( t1.x + d.y,  -t1.y + d.x )
        "v_pk_add_f32 %3, %6, %7 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[1,1]"     // ( t1.x - d.y,  -t1.y - d.x )
        : "=&v"(a0), "=&v"(a2), "=&v"(a1), "=&v"(a3)
        : "v"(t0), "v"(t2), "v"(t1), "v"(d));
}

// the 15 lane-constant factors of one twiddled pass
struct LaneTw {
    cf a[3];    // (w^4)^n1, n1 = 1..3
    cf c[12];   // c[(n2-1)*4 + k1] = w^n2 * W16^(n2 k1), n2 = 1..3, k1 = 0..3
};

// inner layer: DFT4 over n1 for each n2: x[4 n1 + n2] -> y[n2][k1] at v[4 k1 + n2]
__device__ __forceinline__ void fft16_inner(cf (&v)[16])
{
#pragma unroll
    for (int n2 = 0; n2 < 4; n2++) fft4(v[n2], v[4 + n2], v[8 + n2], v[12 + n2]);
}
// outer layer: DFT4 over n2 for each k1 -> X[k1 + 4 k2] at v[4 k1 + k2]
__device__ __forceinline__ void fft16_outer(cf (&v)[16])
{
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++) fft4(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
}
// 16-point forward DFT of x[n] (no external twiddle).  Output X[k] at v[4*(k&3) + (k>>2)].
__device__ __forceinline__ void fft16_plain(cf (&v)[16])
{
    constexpr float C1 = 0.92387953251128673848f;  // cos(pi/8)
    constexpr float S1 = 0.38268343236508978178f;  // sin(pi/8)
    constexpr float R2 = 0.70710678118654752440f;  // cos(pi/4)
    fft16_inner(v);
    // y[n2][k1] *= W16^(n2 k1); e = n2 k1 = 4 (k1 = 2, n2 = 2) rides inside fft4_mi2
    v[4 * 1 + 1] = cmul_cs(v[4 * 1 + 1], C1, S1);    // e = 1
    v[4 * 1 + 2] = cmul_cs(v[4 * 1 + 2], R2, R2);    // e = 2
    v[4 * 1 + 3] = cmul_cs(v[4 * 1 + 3], S1, C1);    // e = 3
    v[4 * 2 + 1] = cmul_cs(v[4 * 2 + 1], R2, R2);    // e = 2
    v[4 * 2 + 3] = cmul_cs(v[4 * 2 + 3], -R2, R2);   // e = 6
    v[4 * 3 + 1] = cmul_cs(v[4 * 3 + 1], S1, C1);    // e = 3
    v[4 * 3 + 2] = cmul_cs(v[4 * 3 + 2], -R2, R2);   // e = 6
    v[4 * 3 + 3] = cmul_cs(v[4 * 3 + 3], -C1, -S1);  // e = 9
    fft4(v[0], v[1], v[2], v[3]);
    fft4(v[4], v[5], v[6], v[7]);
    fft4_mi2(v[8], v[9], v[10], v[11]);
    fft4(v[12], v[13], v[14], v[15]);
}
// 16-point forward DFT of x[n] * w^n with the lane's factors in `tw`
__device__ __forceinline__ void fft16_tw(cf (&v)[16], const LaneTw &tw)
{
#pragma unroll
    for (int n1 = 1; n1 < 4; n1++) {
        cmul2(v[4 * n1 + 0], v[4 * n1 + 1], tw.a[n1 - 1], tw.a[n1 - 1]);
        cmul2(v[4 * n1 + 2], v[4 * n1 + 3], tw.a[n1 - 1], tw.a[n1 - 1]);
    }
    fft16_inner(v);
#pragma unroll
    for (int n2 = 1; n2 < 4; n2++) {
        cmul2(v[4 * 0 + n2], v[4 * 1 + n2], tw.c[(n2 - 1) * 4 + 0], tw.c[(n2 - 1) * 4 + 1]);
        cmul2(v[4 * 2 + n2], v[4 * 3 + n2], tw.c[(n2 - 1) * 4 + 2], tw.c[(n2 - 1) * 4 + 3]);
    }
    fft16_outer(v);
}
// the same with conjugated outputs (fft4_conj in the outer layer)
__device__ __forceinline__ void fft16_tw_conj(cf (&v)[16], const LaneTw &tw)
{
#pragma unroll
    for (int n1 = 1; n1 < 4; n1++) {
        cmul2(v[4 * n1 + 0], v[4 * n1 + 1], tw.a[n1 - 1], tw.a[n1 - 1]);
        cmul2(v[4 * n1 + 2], v[4 * n1 + 3], tw.a[n1 - 1], tw.a[n1 - 1]);
    }
    fft16_inner(v);
#pragma unroll
    for (int n2 = 1; n2 < 4; n2++) {
        cmul2(v[4 * 0 + n2], v[4 * 1 + n2], tw.c[(n2 - 1) * 4 + 0], tw.c[(n2 - 1) * 4 + 1]);
        cmul2(v[4 * 2 + n2], v[4 * 3 + n2], tw.c[(n2 - 1) * 4 + 2], tw.c[(n2 - 1) * 4 + 3]);
    }
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++) fft4_conj(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
}
// register index q holds output bin k = bin_of(q)
__device__ __forceinline__ constexpr int bin_of(int q) { return (q >> 2) + 4 * (q & 3); }

// device table (forward sign), p = 0..14: p < 3 -> (w^4)^(p+1); p = 3 + (n2-1)*4 + k1 -> c[n2][k1]
//   tab[p * 16 + kk]              w = exp(-j 2 pi kk / 256)     pass 2   (240 entries)
//   tab[240 + p * 256 + j]        w = exp(-j 2 pi j / 4096)     pass 3   (3840 entries)
__device__ __forceinline__ void load_pass3_twiddles(LaneTw &t, const float2 *__restrict__ tab, int j)
{
    const cf *tb = reinterpret_cast<const cf *>(tab) + LDS_TW2;
#pragma unroll
    for (int p = 0; p < 3; p++) t.a[p] = tb[p * 256 + j];
#pragma unroll
    for (int p = 0; p < 12; p++) t.c[p] = tb[(3 + p) * 256 + j];
}
// copy the pass-2 table into this workgroup's LDS (call once, before the first pass 2)
__device__ __forceinline__ void stage_pass2_twiddles(cf *lds, const float2 *__restrict__ tab, int j)
{
    if (j < LDS_TW2) lds[LDS_DATA + j] = reinterpret_cast<const cf *>(tab)[j];
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
// The stream is touched exactly once: loads carry the non-temporal hint (aux bit 1).  Measured
// on MI355X (tools/ubench.hip): read-only stream 6.2 -> 7.0 TB/s, copy 5.4 -> 5.8 TB/s with nt.
constexpr int kAuxStream = 2;
template <bool CHECKED, int AUX = kAuxStream>
__device__ __forceinline__ void load_frame(cf (&v)[16], __amdgpu_buffer_rsrc_t rs, int j)
{
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const u32x2 t = CHECKED ? __builtin_amdgcn_raw_buffer_load_b64(rs, (j + 256 * r) * 8, 0, AUX)
                                : __builtin_amdgcn_raw_buffer_load_b64(rs, j * 8, 2048 * r, AUX);
        v[r] = cf{__uint_as_float(t.x), __uint_as_float(t.y)};
    }
}
template <int AUX = 0>
__device__ __forceinline__ void store_cf(__amdgpu_buffer_rsrc_t rs, unsigned voff, cf a)
{
    u32x2 t;
    t.x = __float_as_uint(a.x);
    t.y = __float_as_uint(a.y);
    __builtin_amdgcn_raw_buffer_store_b64(t, rs, (int)voff, 0, AUX);
}

// the workgroup's tables through a descriptor as well: one lane offset and scalar row offsets, no 64-bit address pair per load
__device__ __forceinline__ cf load_cf(__amdgpu_buffer_rsrc_t rs, int voff, int soff)
{
    const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
    return cf{__uint_as_float(t.x), __uint_as_float(t.y)};
}
__device__ __forceinline__ void load_pass3_twiddles(LaneTw &t, __amdgpu_buffer_rsrc_t tab, int j)
{
#pragma unroll
    for (int p = 0; p < 3; p++) t.a[p] = load_cf(tab, j * 8, (LDS_TW2 + p * 256) * 8);
#pragma unroll
    for (int p = 0; p < 12; p++) t.c[p] = load_cf(tab, j * 8, (LDS_TW2 + (3 + p) * 256) * 8);
}

// atan2 for the fused demodulator: min/max ratio through v_rcp_f32 and a degree-6 minimax
// polynomial in t^2 (max error 2.5e-7 rad on [0,1], fitted offline) -- ~22 VALU instructions
// against ~46 for the library atan2f, with an error two orders below the 1e-5*pi parity bar.
// Quadrants from the SIGN BITS, so (+-0, +-0) gives 0 / +-pi exactly as atan2f does (the first
// FreqDemod output after activate() is arg of a signed zero, FreqDemod.cpp:44-47,63-64).
__device__ __forceinline__ float fast_atan2f(float y, float x)
{
    const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    const float mx = __builtin_fmaxf(ax, ay), mn = __builtin_fminf(ax, ay);
    const float t = mx > 0.f ? mn * __builtin_amdgcn_rcpf(mx) : 0.f;
    const float s = t * t;
    float p = 0.006811532657593489f;
    p = __builtin_fmaf(p, s, -0.03360334783792496f);
    p = __builtin_fmaf(p, s, 0.0796225368976593f);
    p = __builtin_fmaf(p, s, -0.1323327124118805f);
    p = __builtin_fmaf(p, s, 0.19807793200016022f);
    p = __builtin_fmaf(p, s, -0.3331736624240875f);
    p = __builtin_fmaf(p, s, 0.9999961256980896f);
    float r = t * p;
    r = ay > ax ? 1.57079632679489661923f - r : r;
    r = (__float_as_uint(x) >> 31) ? 3.14159265358979323846f - r : r;
    return __builtin_copysignf(r, y);
}

// two samples at a time: the ratio, the polynomial (Horner in t^2), the final product and the two quadrant
// reflections run on packed (v_pk_mul/fma/add_f32) pairs -- 11 fewer VALU issues per pair than two calls of
// the scalar form, same operations and roundings per component (min/max/rcp/selects have no packed form)
__device__ __forceinline__ cf fast_atan2f_x2(cf y, cf x)
{
    const cf ax = {__builtin_fabsf(x.x), __builtin_fabsf(x.y)}, ay = {__builtin_fabsf(y.x), __builtin_fabsf(y.y)};
    const cf mx = {__builtin_fmaxf(ax.x, ay.x), __builtin_fmaxf(ax.y, ay.y)};
    const cf mn = {__builtin_fminf(ax.x, ay.x), __builtin_fminf(ax.y, ay.y)};
    const cf rc = {mx.x > 0.f ? __builtin_amdgcn_rcpf(mx.x) : 0.f, mx.y > 0.f ? __builtin_amdgcn_rcpf(mx.y) : 0.f};
    const cf t = mn * rc;
    const cf s = t * t;
    auto k2 = [](float c) { return cf{c, c}; };
    cf p = k2(0.006811532657593489f);
    p = __builtin_elementwise_fma(p, s, k2(-0.03360334783792496f));
    p = __builtin_elementwise_fma(p, s, k2(0.0796225368976593f));
    p = __builtin_elementwise_fma(p, s, k2(-0.1323327124118805f));
    p = __builtin_elementwise_fma(p, s, k2(0.19807793200016022f));
    p = __builtin_elementwise_fma(p, s, k2(-0.3331736624240875f));
    p = __builtin_elementwise_fma(p, s, k2(0.9999961256980896f));
    cf r = t * p;
    const cf ro = k2(1.57079632679489661923f) - r;
    r = cf{ay.x > ax.x ? ro.x : r.x, ay.y > ax.y ? ro.y : r.y};
    const cf rn = k2(3.14159265358979323846f) - r;
    r = cf{(__float_as_uint(x.x) >> 31) ? rn.x : r.x, (__float_as_uint(x.y) >> 31) ? rn.y : r.y};
    return cf{__builtin_copysignf(r.x, y.x), __builtin_copysignf(r.y, y.y)};
}

// ---- helpers of the radix-16 family plans (fft_r16.hip) ----
__device__ __forceinline__ int padi(int i) { return i + (i >> 4); }
__device__ __forceinline__ cf cmul1(cf a, cf w)
{
    cf t, r;
    asm("v_pk_mul_f32 %0, %2, %3 op_sel:[1,1] op_sel_hi:[0,1] neg_lo:[0,1]\n\t"
        "v_pk_fma_f32 %1, %2, %3, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]"
        : "=&v"(t), "=&v"(r)
        : "v"(a), "v"(w));
    return r;
}
// forward DFT8, natural order in and out
__device__ __forceinline__ void fft8(cf &a0, cf &a1, cf &a2, cf &a3, cf &a4, cf &a5, cf &a6, cf &a7)
{
    constexpr float R2 = 0.70710678118654752440f;
    cf e0 = a0, e1 = a2, e2 = a4, e3 = a6, o0 = a1, o1 = a3, o2 = a5, o3 = a7;
    fft4(e0, e1, e2, e3);
    fft4(o0, o1, o2, o3);
    const cf w1 = cmul_cs(o1, R2, R2);        // W8^1
    const cf w3 = cmul_cs(o3, -R2, R2);       // W8^3
    cf p2, m2;
    addsub_mi(p2, m2, e2, o2);                // e2 +- (-i) o2   (W8^2 = -i)
    a0 = e0 + o0; a4 = e0 - o0;
    a1 = e1 + w1; a5 = e1 - w1;
    a2 = p2;      a6 = m2;
    a3 = e3 + w3; a7 = e3 - w3;
}

// The passes' barriers order LDS accesses only (the exchange image, the pass-2 table, a kernel's own LDS words): a release /
// acquire pair on the LOCAL address space around s_barrier = `s_waitcnt lgkmcnt(0); s_barrier`.  __syncthreads() also drains
// vmcnt -- every outstanding global load, store and atomic of the wave -- which made the first barrier of a block wait for the
// block dealer's atomic (pcx_sched.hpp) and would make any barrier wait for stores still in flight.
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// PART (timing-only builds of the energy split, wrong outputs): 0 = the real pass, 1 = butterflies only
// (no LDS exchange, no barriers), 2 = exchange only (no butterfly arithmetic)
// pass 1: v[r] = x[j + 256 r] on entry; leaves the pass-1 result in LDS.  pass1 = pass1_math + pass1_exchange: a kernel that
// deals its blocks dynamically issues the draw between the two (every load of the block has been consumed by then).
template <int PART = 0>
__device__ __forceinline__ void pass1_math(cf (&v)[16])
{
    if (PART != 2) fft16_plain(v);
}
template <int PART = 0>
__device__ __forceinline__ void pass1_exchange(cf (&v)[16], cf *lds, int j)
{
    if (PART == 1) return;
    lds_barrier();  // previous readers of this LDS image are done
#pragma unroll
    for (int q = 0; q < 16; q++) lds[17 * j + bin_of(q)] = v[q];  // pad(16 j + k) = 17 j + k
}
template <int PART = 0>
__device__ __forceinline__ void pass1(cf (&v)[16], cf *lds, int j)
{
    pass1_math<PART>(v);
    pass1_exchange<PART>(v, lds, j);
}
template <int PART = 0>
__device__ __forceinline__ void pass2(cf (&v)[16], cf *lds, int j)
{
    if (PART != 1) {
        lds_barrier();
        const int rb = j + (j >> 4);  // pad(j + 256 r) = rb + 272 r
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] = lds[rb + 272 * r];
    }
    LaneTw tw;                    // this lane's pass-2 factors: 16 distinct rows, broadcast reads
    const cf *t2 = lds + LDS_DATA + (j & 15);
#pragma unroll
    for (int p = 0; p < 3; p++) tw.a[p] = t2[p * 16];
#pragma unroll
    for (int p = 0; p < 12; p++) tw.c[p] = t2[(3 + p) * 16];
    if (PART != 2) fft16_tw(v, tw);
    else v[0] = v[0] + tw.a[0] + tw.c[0];   // keep the table reads alive
    if (PART == 1) return;
    lds_barrier();
    const int wb = (j >> 4) * 272 + (j & 15);  // pad((j>>4)*256 + kk + 16 k) = wb + 17 k
#pragma unroll
    for (int q = 0; q < 16; q++) lds[wb + 17 * bin_of(q)] = v[q];
}
// pass 3: on exit v[q] = X[j + 256 * bin_of(q)]; CONJ: conj(X[..]) (the closing conjugation of an inverse transform run on the
// forward passes, for free)
template <int PART = 0, bool CONJ = false>
__device__ __forceinline__ void pass3(cf (&v)[16], const cf *lds, int j, const LaneTw &tw3)
{
    if (PART != 1) {
        lds_barrier();
        const int rb = j + (j >> 4);
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] = lds[rb + 272 * r];
    }
    if (PART != 2) {
        if (CONJ) fft16_tw_conj(v, tw3);
        else fft16_tw(v, tw3);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------- //
// The transform PAIR of an overlap-save block with its second exchange kept inside sixteen lanes.
//
// A convolution does not care in which order the spectrum sits in the registers, only that H sits in the same order.  So the
// forward transform runs decimation in FREQUENCY (butterfly, then twiddle: `fft16_post`) and the inverse is its transpose,
// decimation in time (the passes above), and the two meet on a spectrum that is digit-reversed ACROSS LANES:
//   lane j = 16 ka + kb, register q:  X[ka + 16 kb + 256 bin_of(q)]                     (H is loaded once, in that order)
// Index algebra, n = j + 256 r on the way in, k = ka + 16 kb + 256 kc:
//   A   lane j:            DFT16 over r, times W4096^(j ka)            -> a[ka][j]             exchange 1 (whole workgroup)
//   B   lane (ka, j1):     DFT16 over j2 (j = j1 + 16 j2), W256^(j1 kb) -> b[ka][kb][j1]        exchange 2 (the 16 lanes of ka)
//   C   lane (ka, kb):     DFT16 over j1                               -> X[ka + 16 kb + 256 kc]
//   C^T, exchange 2^T, B^T (= pass 2 above: twiddle, DFT16), exchange 1^T, A^T (= pass 3 above) bring conj(IFFT) back in
//   natural order: lane j holds y[j + 256 bin_of(q)], as the stores want it.
// Image: sixteen rows of 272 elements (the same 34,816 B).  Row ka is written by everyone in exchange 1 and from then on touched
// only by the sixteen lanes of ka -- one quarter of one wave -- until exchange 1^T has been read: exchange 2 and 2^T need no
// s_barrier at all (LDS executes one wave's instructions in order), exchange 1^T needs none in front of its writes.  Three
// barriers per block instead of eight.  Inside a row exchange 1 uses element 16 j2 + j1, exchange 2 uses 17 kb + j1: every
// access below is conflict-free for ds_*_b64 (32 lanes over 64 banks) -- rows are 2,176 B = 8.5 bank rows apart, so the two
// rows under one half-wave fall on opposite halves of the banks, and a lane stride of 17 elements walks all 64 banks.
// The LDS instruction count drops as well: 16 + 8 + 8 + 8 per transform instead of 8 + 16 + 8 + 16.
// ---------------------------------------------------------------------------------------------------------------------------- //
// 16-point forward DFT of x[n], outputs X[k] * w^k: the transpose of fft16_tw, same fifteen factors (c'[n2][k1] = c[k1][n2])
__device__ __forceinline__ void fft16_post(cf (&v)[16], const LaneTw &tw)
{
    fft16_inner(v);                                  // y[n2][k1] at v[4 k1 + n2]
#pragma unroll
    for (int k1 = 1; k1 < 4; k1++) {                 // times w^k1 * W16^(n2 k1)
        cmul2(v[4 * k1 + 0], v[4 * k1 + 1], tw.c[(k1 - 1) * 4 + 0], tw.c[(k1 - 1) * 4 + 1]);
        cmul2(v[4 * k1 + 2], v[4 * k1 + 3], tw.c[(k1 - 1) * 4 + 2], tw.c[(k1 - 1) * 4 + 3]);
    }
    fft16_outer(v);                                  // X[k1 + 4 k2] * w^k1 at v[4 k1 + k2]
#pragma unroll
    for (int k2 = 1; k2 < 4; k2++) {                 // times (w^4)^k2
        cmul2(v[4 * 0 + k2], v[4 * 1 + k2], tw.a[k2 - 1], tw.a[k2 - 1]);
        cmul2(v[4 * 2 + k2], v[4 * 3 + k2], tw.a[k2 - 1], tw.a[k2 - 1]);
    }
}
// one wave's LDS writes before its LDS reads: ordering for the compiler only, the hardware keeps a wave's LDS traffic in order
__device__ __forceinline__ void wave_lds_order()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}
// the lane that holds spectrum bin (j' + 256 kc) of the natural order is lane j with j' = spec_lane(j)
__device__ __forceinline__ int spec_lane(int j) { return (j >> 4) + 16 * (j & 15); }
// H[k] = spec[spec_lane(j) + 256 k], once per workgroup: fetched in natural order (whole 512-byte rows per wave) and turned
// across the lanes through the block image -- as a gather straight from memory every lane would touch sixteen lines of its own
// (4 Mi requests per launch of 1024 workgroups: measured 0.6 % of the headline launch)
// in two halves, so that a kernel can put the requests FIRST in its prologue and the turn behind everything else it requests:
// the memory counter retires in order, and whatever is waited for drags every older request with it
struct SpectrumLoad {
    cf row[16];
    cf tw2;
};
__device__ __forceinline__ void spectrum_request(SpectrumLoad &t, const float2 *__restrict__ spec, const float2 *__restrict__ twtab, int j)
{
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(spec, N * 8);
#pragma unroll
    for (int k = 0; k < 16; k++) t.row[k] = load_cf(rs, j * 8, 2048 * k);
    t.tw2 = load_cf(make_rsrc(twtab, LDS_TW2 * 8), j * 8, 0);        // lanes >= 240: out of range, reads 0, not stored
}
// also stages the pass-2 twiddle table (stage_pass2_twiddles)
__device__ __forceinline__ void spectrum_turn(cf (&H)[16], const SpectrumLoad &t, cf *lds, int j)
{
#pragma unroll
    for (int k = 0; k < 16; k++) lds[j + 272 * k] = t.row[k];
    if (j < LDS_TW2) lds[LDS_DATA + j] = t.tw2;
    lds_barrier();
#pragma unroll
    for (int k = 0; k < 16; k++) H[k] = lds[spec_lane(j) + 272 * k];
    // (the next writer of the image is exchange 1, behind its own barrier)
}
__device__ __forceinline__ void load_spectrum_lanes(cf (&H)[16], const float2 *__restrict__ spec, const float2 *__restrict__ twtab, cf *lds, int j)
{
    SpectrumLoad t;
    spectrum_request(t, spec, twtab, j);
    spectrum_turn(H, t, lds, j);
}
__device__ __forceinline__ void load_pass2_twiddles(LaneTw &tw, const cf *lds, int j)
{
    const cf *t2 = lds + LDS_DATA + (j & 15);
#pragma unroll
    for (int p = 0; p < 3; p++) tw.a[p] = t2[p * 16];
#pragma unroll
    for (int p = 0; p < 12; p++) tw.c[p] = t2[(3 + p) * 16];
}
// forward, part 1: v[r] = x[j + 256 r] -> butterflies of A (a dealt kernel issues its draw behind this)
__device__ __forceinline__ void dif_a_math(cf (&v)[16], const LaneTw &tw3) { fft16_post(v, tw3); }
// forward, the rest: on exit v[q] = X[spec_lane(j) + 256 bin_of(q)].  `before_last` runs in front of the last sixteen-point
// transform, the registers of the pass-2 factors free again (fir_ols_part.hip issues its first table loads there)
struct NoHook { __device__ __forceinline__ void operator()() const {} };
template <typename HOOK = NoHook>
__device__ __forceinline__ void dif_rest(cf (&v)[16], cf *lds, int j, HOOK before_last = HOOK())
{
    const int row = 272 * (j >> 4), l = j & 15;
    lds_barrier();                                   // the readers of the previous exchange 1^T are done
#pragma unroll
    for (int q = 0; q < 16; q++) lds[j + 272 * bin_of(q)] = v[q];
    lds_barrier();
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = lds[row + l + 16 * r];
    LaneTw tw2;
    load_pass2_twiddles(tw2, lds, j);
    fft16_post(v, tw2);
    wave_lds_order();
#pragma unroll
    for (int q = 0; q < 16; q++) lds[row + l + 17 * bin_of(q)] = v[q];
    wave_lds_order();
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = lds[row + 17 * l + r];
    before_last();
    fft16_plain(v);
}
// inverse on conjugated input: u[r] = conj(Y)[spec_lane(j) + 256 r] on entry, u[q] = IFFT(Y)[j + 256 bin_of(q)] on exit
// (CONJ = false: conj(IFFT(Y)), for a kernel that wants the conjugate anyway)
template <bool CONJ = true>
__device__ __forceinline__ void dit_back(cf (&u)[16], cf *lds, int j, const LaneTw &tw3)
{
    const int row = 272 * (j >> 4), l = j & 15;
    fft16_plain(u);
    wave_lds_order();
#pragma unroll
    for (int q = 0; q < 16; q++) lds[row + 17 * l + bin_of(q)] = u[q];
    wave_lds_order();
#pragma unroll
    for (int r = 0; r < 16; r++) u[r] = lds[row + l + 17 * r];
    LaneTw tw2;
    load_pass2_twiddles(tw2, lds, j);
    fft16_tw(u, tw2);
    wave_lds_order();
#pragma unroll
    for (int q = 0; q < 16; q++) lds[row + l + 16 * bin_of(q)] = u[q];
    lds_barrier();
#pragma unroll
    for (int r = 0; r < 16; r++) u[r] = lds[j + 272 * r];
    if (CONJ) fft16_tw_conj(u, tw3);
    else fft16_tw(u, tw3);
}

}  // namespace fft4k
}  // namespace pcx
