// fir_ols_f64.hip -- overlap-save /comms/fir_filter for complex_float64 streams (M = L = 1), and on the same pipeline the
// complex_int16 / complex_int8 streams (bit-exact) and the real float64 / int16 / int8 ones: the y = IFFT(FFT(block) .* H)
// evaluation of the FIRFilter.cpp:294-300 convolution in double precision.
//
// Block geometry is the float kernels': Kov >= K-1 (a multiple of 16) outputs dropped per N-sample block, block b's window
// starts pad = Kov-(K-1) samples before sample b*S, S = N - Kov.
//
// K <= 2049 (N = 4096): the IN-PLACE transform pair of fft_f64.hpp (ip4096): 4 barriers per block, a conflict-free image, H
// (64 VGPRs) and thirteen powers of the lane's pass-1 factor (52 VGPRs) in registers across the block loop, blocks DEALT to 512
// persistent workgroups (pcx_sched.hpp), wave priority rising through a block, the next block's samples fetched ahead for the
// integer streams (16 VGPRs; a complex_float64 block would need 64), no scratch in any instantiation.
// 2049 < K <= 4097 (N = 8192): the Stockham radix-16 passes of xform<13> as before (lane l holds x[l + s*LPF]; a forward
// transform leaves X[l + k*LPF] in the lane, which is the next transform's first-pass layout).
//
// Rounding: everything is double; the result differs from the reference's direct sum by a few 1e-16 of the output scale
// (parity bar 1e-13).  Against the sliding-window kernel (K multiply-adds per output on the 78 TFLOP/s f64 pipe) this is
// the faster form from a few tens of taps up (tools/sweep_fir_f64.py).  Its roof is the FP64 vector pipe, not HBM: about
// 1,330 instructions per 64-lane wave and 3,840-sample block against 4 (int16) to 32 (float64) bytes per sample -- and that pipe
// issues one instruction per ~3.9 clocks and SIMD at the two waves per SIMD the 64 KB image allows (tools/f64_lab.hip): the
// complex_int16 kernel runs at 0.9 of that rate (250 Gsamples/s at 255 taps).
#include "fft_f64.hpp"
#include <cstdio>
#include <cstdlib>

#include "pcx_internal.hpp"
#include "pcx_sched.hpp"

namespace pcx {

namespace {
using namespace fft64;

template <int LOG2N>
struct OlsPlan {
    static constexpr int N = 1 << LOG2N;
    static constexpr int LPF = N / 16;
    static constexpr int A = LOG2N / 4;
    static constexpr int R = 1 << (LOG2N % 4);
    static constexpr int NTWF = R > 1 ? (16 / R) * (R - 1) : 0;
    static constexpr int LDS_IMG = N + N / 16;
    static constexpr int LDS_T2 = 240;
    static constexpr int T3_OFF = 15 * 16;
    static constexpr int TF_OFF = T3_OFF + (A >= 3 ? 15 * 256 : 0);
    static constexpr bool NATURAL = R > 1;
};

// forward DFT_N of the block held as v[s] = x[l + s*LPF]; on exit v[q] = X[l + LPF*(NATURAL ? q : bin_of(q))]
// TFS: stride of the final-pass factors behind `tf` (1: the lane's register copy; LPF: the device table itself, re-read at each use)
template <int LOG2N, typename T3, int TFS = 1>
__device__ __forceinline__ void xform(cd (&v)[16], cd *lds, int l, T3 t3, const cd *tf)
{
    typedef OlsPlan<LOG2N> P;
    constexpr int LPF = P::LPF, A = P::A, R = P::R;
    fft16_plain(v);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; q++) lds[17 * l + bin_of(q)] = v[q];
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 16; s++) v[s] = lds[l + (l >> 4) + s * (LPF + LPF / 16)];
    {
        const cd *t2 = lds + P::LDS_IMG + (l & 15);
        fft16_tw(v, [&](int p) { return t2[p * 16]; });
    }
    __syncthreads();
    {
        const int wb = (l >> 4) * 272 + (l & 15);
#pragma unroll
        for (int q = 0; q < 16; q++) lds[wb + 17 * bin_of(q)] = v[q];
    }
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
    if (R > 1) {
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 16; s++) v[s] = lds[padi(l + s * LPF)];
        constexpr int G = 16 / R;
#pragma unroll
        for (int t = 0; t < G; t++) {
#pragma unroll
            for (int r = 1; r < R; r++) v[t + r * G] = cmul(v[t + r * G], tf[(t * (R - 1) + (r - 1)) * TFS]);
            if (R == 2) {
                const cd a = v[t], b = v[t + G];
                v[t] = a + b;
                v[t + G] = a - b;
            } else if (R == 4) {
                fft4(v[t], v[t + G], v[t + 2 * G], v[t + 3 * G]);
            } else {
                fft8(v[t], v[t + G], v[t + 2 * G], v[t + 3 * G], v[t + 4 * G], v[t + 5 * G], v[t + 6 * G], v[t + 7 * G]);
            }
        }
    }
}

// Stream element types this pipeline serves.  IO = 0: complex_float64 as is.  IO = 1 / 2: complex_int16 / complex_int8 --
// the reference's integer FIR is the EXACT integer convolution with the Q-format taps reduced modulo 2^qbits, then
// fromQ (>> qbits/2) and truncation to the element width (FIRFilter.cpp:295-300, Pothos::Util::fromQ).  A double
// transform carries that convolution exactly: with |x| <= 2^15, ||h_q||_2 < 2^22 (checked by the caller) and 4096-
// or 8192-sample blocks its error stays below 0.05 (eps * c log2 N * ||x||_2 ||h||_2), so rounding to the nearest
// integer recovers every sum bit for bit before the same wrap, shift and truncation are applied.
// the exact integer sum held in a double -> its value modulo 2^32: adding 1.5 * 2^52 leaves round-to-nearest-even(d) in the low dword of
// the sum -- one v_add_f64 instead of the software double -> int64 conversion (|d| < 2^51: sums stay below 2^45 under the caller's bound)
__device__ __forceinline__ int wrap_i32(double d) { return __double2loint(d + 6755399441055744.0); }
template <int IO>
struct StreamIo;
template <>
struct StreamIo<0> {
    static constexpr int EB = 16;
    template <int AUX> static __device__ __forceinline__ cd load(__amdgpu_buffer_rsrc_t rs, int voff, int soff)
    {
        return as_cd(__builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, AUX));
    }
    static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t ws, int voff, cd y, QShift)
    {
        __builtin_amdgcn_raw_buffer_store_b128(as_u4(y), ws, voff, 0, kAuxStream);
    }
};
template <>
struct StreamIo<1> {
    static constexpr int EB = 4;
    template <int AUX> static __device__ __forceinline__ cd load(__amdgpu_buffer_rsrc_t rs, int voff, int soff)
    {
        const unsigned t = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, AUX);
        return cd{(double)(short)(t & 0xffffu), (double)(short)(t >> 16)};
    }
    static __device__ __forceinline__ unsigned q(double d, QShift qs)   // wrap to the 32-bit Q accumulator, fromQ (pcx_qformat.hpp), truncate to int16
    {
        const int w = wrap_i32(d);
        return (unsigned)from_q_bits<int>(w, qs) & 0xffffu;
    }
    static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t ws, int voff, cd y, QShift qs)
    {
        __builtin_amdgcn_raw_buffer_store_b32(q(y.x, qs) | (q(y.y, qs) << 16), ws, voff, 0, kAuxStream);
    }
};
template <>
struct StreamIo<2> {
    static constexpr int EB = 2;
    template <int AUX> static __device__ __forceinline__ cd load(__amdgpu_buffer_rsrc_t rs, int voff, int soff)
    {
        const unsigned t = __builtin_amdgcn_raw_buffer_load_b16(rs, voff, soff, AUX);
        return cd{(double)(signed char)(t & 0xffu), (double)(signed char)((t >> 8) & 0xffu)};
    }
    static __device__ __forceinline__ unsigned q(double d, QShift qs)   // 16-bit Q accumulator, fromQ, truncate to int8
    {
        const short w = (short)wrap_i32(d);
        return (unsigned)from_q_bits<short>(w, qs) & 0xffu;
    }
    static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t ws, int voff, cd y, QShift qs)
    {
        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(q(y.x, qs) | (q(y.y, qs) << 8)), ws, voff, 0, kAuxStream);
    }
};

// DECIM: decimation M > 1 (interpolation 1): the block is evaluated at full rate and only the outputs the reference's
// decimator keeps -- full-rate index n with (n + 1) % M == 0, FIRFilter.cpp:291 -- are stored, at (n + 1) / M - 1.
// n_out counts full-rate outputs, n_dec the stored ones; magic = ceil(2^32 / M) (exact quotient for t * M < 2^32).
template <int LOG2N, int IO, bool DECIM>
__global__ __launch_bounds__(OlsPlan<LOG2N>::LPF) __attribute__((amdgpu_waves_per_eu(2))) void fir_cf64_ols_kernel(const unsigned char *__restrict__ in, size_t in_elems,
                                                                          unsigned char *__restrict__ out, size_t n_out, size_t n_dec, unsigned M, unsigned magic,
                                                                          const double2 *__restrict__ Hspec, int Kov, int pad,
                                                                          const double2 *__restrict__ twtab, size_t first_full,
                                                                          size_t nfull, size_t nblocks, QShift qs)
{
    typedef OlsPlan<LOG2N> P;
    typedef StreamIo<IO> SIO;
    constexpr int N = P::N, LPF = P::LPF, EB = SIO::EB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cd *lds = reinterpret_cast<cd *>(smem_raw);
    const int l = threadIdx.x;
    const size_t S = (size_t)(N - Kov);
    size_t b = blockIdx.x;
    if (b >= nblocks) return;
    const cd *tab = reinterpret_cast<const cd *>(twtab);
    cd t3[P::A >= 3 ? 15 : 1];
    if (P::A >= 3) {
#pragma unroll
        for (int p = 0; p < 15; p++) t3[p] = tab[P::T3_OFF + p * 256 + (l & 255)];
    }
    cd tf[P::NTWF > 0 ? P::NTWF : 1];
#pragma unroll
    for (int p = 0; p < P::NTWF; p++) tf[p] = tab[P::TF_OFF + p * LPF + l];
    for (int i = l; i < P::LDS_T2; i += LPF) lds[P::LDS_IMG + i] = tab[i];
    // the lane's 16 bins of H (64 VGPRs as doubles) are re-read from L2 at the multiply, as fir_real_ols_kernel does: with the pass-256
    // factors (60) and the final-pass factors (32) resident they do not fit beside the block (this kernel serves N = 8192 only since
    // round 6; held in registers it spilled 30-61 VGPRs)
    const cd *Hg = reinterpret_cast<const cd *>(Hspec) + l;
    auto tw3 = [&](int p) { return t3[p]; };
    const int nov = (Kov + LPF - 1) / LPF;   // window rows shared with a neighbouring block

    for (; b < nblocks; b += gridDim.x) {
        cd v[16];
        if (b >= first_full && b < nfull) {
            // overlap rows (the first and last of the window, shared with the neighbouring blocks) keep the
            // default cache policy, the rest of the window is touched once
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(in + (b * S - pad) * EB, N * EB);
#pragma unroll
            for (int s = 0; s < 16; s++)
                v[s] = s < nov || s >= 16 - nov ? SIO::template load<0>(rs, l * EB, s * LPF * EB) : SIO::template load<kAuxStream>(rs, l * EB, s * LPF * EB);
        } else {
            // ragged: block 0 when pad > 0 (samples before the buffer only feed dropped outputs) and the tail
            const size_t shift = b * S >= (size_t)pad ? 0 : (size_t)pad - b * S;
            const size_t first = b * S + shift - pad;
            const size_t left = in_elems > first ? in_elems - first : 0;
            const size_t want = (size_t)N - shift;
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(in + first * EB, (unsigned)((left < want ? left : want) * EB));
#pragma unroll
            for (int s = 0; s < 16; s++) v[s] = SIO::template load<0>(rs, (l + LPF * s - (int)shift) * EB, 0);
        }
        xform<LOG2N>(v, lds, l, tw3, tf);
        // u = conj(X .* H) in the first-pass layout of the next transform (register k <- bin k*LPF + l)
        const cd *Hb = Hg;
        asm volatile("" : "+v"(Hb));   // keeps the loads inside the loop (they are loop-invariant and would be hoisted back into registers)
        cd u[16];
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int k = P::NATURAL ? q : bin_of(q);
            const cd p = cmul(v[q], Hb[LPF * k]);
            u[k] = cd{p.x, -p.y};
        }
        xform<LOG2N>(u, lds, l, tw3, tf);
        if (DECIM) {
            // b*S = B0*M + base: full-rate output b*S + (i - Kov) is kept when t = base + (i - Kov) + 1 is a multiple of M,
            // and lands at B0 + t/M - 1
            const size_t B0 = (b * S) / M;
            const unsigned base = (unsigned)((b * S) - B0 * M);
            const size_t room = n_dec > B0 ? n_dec - B0 : 0;
            const __amdgpu_buffer_rsrc_t ws = make_rsrc(out + B0 * EB, (unsigned)((room < (size_t)(N / 2 + 2) ? room : (size_t)(N / 2 + 2)) * EB));
            const size_t full_left = n_out - b * S;           // full-rate outputs this block may produce
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int row = LPF * (P::NATURAL ? q : bin_of(q));
                if (row + LPF - 1 < Kov) continue;
                const int i = l + row;
                const unsigned t = base + (unsigned)(i - Kov) + 1u;
                const unsigned qt = __umulhi(t, magic);
                if (i >= Kov && (size_t)(i - Kov) < full_left && qt * M == t)
                    SIO::store(ws, (int)((qt - 1u) * (unsigned)EB), cd{u[q].x, -u[q].y}, qs);
            }
        } else {
        const size_t room = n_out - b * S;
        const __amdgpu_buffer_rsrc_t ws = make_rsrc(out + b * S * EB, (unsigned)((room < S ? room : S) * EB));
        const unsigned vbase = (unsigned)(l - Kov) * (unsigned)EB;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int row = LPF * (P::NATURAL ? q : bin_of(q));
            if (row + LPF - 1 < Kov) continue;                // whole row dropped: uniform skip
            SIO::store(ws, (int)(vbase + (unsigned)row * (unsigned)EB), cd{u[q].x, -u[q].y}, qs);
        }
        }
    }
}

template <int LOG2N, int IO>
int launch_ols(const void *in, size_t in_elems, void *out, size_t n_out, const void *Hspec, size_t K, const void *tw, size_t M, QShift qs, hipStream_t st)
{
    typedef OlsPlan<LOG2N> P;
    const size_t Km1 = K - 1;
    const size_t Kov = (Km1 + 15) / 16 * 16;
    if (Kov > (size_t)P::N / 2) { set_error("fir ols f64: K=%zu too long for %d-sample blocks", K, P::N); return PCX_ERR_UNSUPPORTED; }
    const size_t pad = Kov - Km1;
    const size_t S = P::N - Kov;
    const size_t nblocks = (n_out + S - 1) / S;
    const size_t first_full = pad > 0 ? 1 : 0;
    size_t nfull = n_out / S;
    while (nfull > first_full && (nfull - 1) * S - pad + P::N > in_elems) nfull--;
    if (nfull < first_full) nfull = first_full;
    const size_t lds = (size_t)(P::LDS_IMG + P::LDS_T2) * sizeof(cd);
    auto k = M > 1 ? fir_cf64_ols_kernel<LOG2N, IO, true> : fir_cf64_ols_kernel<LOG2N, IO, false>;
    const unsigned magic = M > 1 ? (unsigned)(((1ull << 32) + M - 1) / M) : 0u;
    if (lds > 64 * 1024) PCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // resident workgroups per CU: LDS (160 KiB) and 8 waves of <= 256 VGPRs
    unsigned per_cu = (unsigned)(160 * 1024 / lds);
    const unsigned by_waves = 8u * 64u / P::LPF;
    if (per_cu > by_waves) per_cu = by_waves;
    if (per_cu < 1) per_cu = 1;
    // about four blocks per workgroup whatever the call (pcx_internal.hpp rounds_grid): eight workgroups queued per slot measured
    // +5 % at 64 Mi samples (tools/ab_oversub.sh), which is this; PCX_OVERSUB (diagnostic library) brings the fixed factor back
    const unsigned grid = PCX_ENV_INT("PCX_OVERSUB", 0) > 0 ? persistent_grid(nblocks, 256 * per_cu, 1) : rounds_grid(nblocks, 256 * per_cu, 4);
    hipLaunchKernelGGL(k, dim3(grid), dim3(P::LPF), lds, st, (const unsigned char *)in, in_elems, (unsigned char *)out, n_out, n_out / M,
                       (unsigned)M, magic, (const double2 *)Hspec, (int)Kov, (int)pad, (const double2 *)tw, first_full, nfull, nblocks, qs);
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}

// --------------------------------------------------------------------------------- //
// REAL streams (float64, int16, int8; real taps; M = L = 1): two consecutive real blocks ride one complex transform
// as its real and imaginary parts -- real taps filter Re and Im independently (H is the transform of a real sequence) --
// exactly as fir_f32_ols4096_kernel does for float32.  Complex block c carries real blocks 2c and 2c+1; every window goes
// through the range-checked descriptor form (samples in front of the buffer read 0 and only feed dropped outputs; a
// missing second block reads 0 and stores nothing).  Integer element types round to the nearest integer and apply the
// reference's wrap / fromQ shift / truncation as StreamIo<1,2> do: bit-exact on the same error bound.
// --------------------------------------------------------------------------------- //
template <int IO>
struct RealIo;
template <>
struct RealIo<0> {   // float64
    static constexpr int EB = 8;
    static __device__ __forceinline__ double load(__amdgpu_buffer_rsrc_t rs, int voff)
    {
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, 0, 0));
    }
    static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t ws, int voff, double y, QShift)
    {
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, y), ws, voff, 0, kAuxStream);
    }
};
template <>
struct RealIo<1> {   // int16: 32-bit Q accumulator, >> 16
    static constexpr int EB = 2;
    static __device__ __forceinline__ double load(__amdgpu_buffer_rsrc_t rs, int voff)
    {
        return (double)(short)__builtin_amdgcn_raw_buffer_load_b16(rs, voff, 0, 0);
    }
    static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t ws, int voff, double y, QShift qs)
    {
        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)StreamIo<1>::q(y, qs), ws, voff, 0, kAuxStream);
    }
};
template <>
struct RealIo<2> {   // int8: 16-bit Q accumulator, >> 8
    static constexpr int EB = 1;
    static __device__ __forceinline__ double load(__amdgpu_buffer_rsrc_t rs, int voff)
    {
        return (double)(signed char)__builtin_amdgcn_raw_buffer_load_b8(rs, voff, 0, 0);
    }
    static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t ws, int voff, double y, QShift qs)
    {
        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)StreamIo<2>::q(y, qs), ws, voff, 0, kAuxStream);
    }
};

template <>
struct RealIo<3> {   // float32 (decimating real float32 filters: the undecimated stream has its own kernel, fir_ols.hip)
    static constexpr int EB = 4;
    static __device__ __forceinline__ double load(__amdgpu_buffer_rsrc_t rs, int voff)
    {
        return (double)__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, 0, 0));
    }
    static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t ws, int voff, double y, QShift)
    {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((float)y), ws, voff, 0, kAuxStream);
    }
};

// DECIM as in fir_cf64_ols_kernel: full-rate evaluation, one output in M stored
template <int LOG2N, int IO, bool DECIM>
__global__ __launch_bounds__(OlsPlan<LOG2N>::LPF) __attribute__((amdgpu_waves_per_eu(2))) void fir_real_ols_kernel(
    const unsigned char *__restrict__ in, size_t in_elems, unsigned char *__restrict__ out, size_t n_out, const double2 *__restrict__ Hspec,
    int Kov, int pad, const double2 *__restrict__ twtab, size_t nblocks_real, size_t n_dec, unsigned M, unsigned magic, QShift qs)
{
    typedef OlsPlan<LOG2N> P;
    typedef RealIo<IO> RIO;
    constexpr int N = P::N, LPF = P::LPF, EB = RIO::EB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cd *lds = reinterpret_cast<cd *>(smem_raw);
    const int l = threadIdx.x;
    const size_t S = (size_t)(N - Kov);
    const size_t nblocks = (nblocks_real + 1) / 2;
    size_t b = blockIdx.x;
    if (b >= nblocks) return;
    const cd *tab = reinterpret_cast<const cd *>(twtab);
    cd t3[P::A >= 3 ? 15 : 1];
    if (P::A >= 3) {
#pragma unroll
        for (int p = 0; p < 15; p++) t3[p] = tab[P::T3_OFF + p * 256 + (l & 255)];
    }
    // (the final-pass factors, 32 VGPRs at N = 8192, are re-read from the device table at their use: the decimating instantiations
    // spilled four registers with them resident)
    const cd *tfg = tab + P::TF_OFF + l;
    for (int i = l; i < P::LDS_T2; i += LPF) lds[P::LDS_IMG + i] = tab[i];
    // the lane's 16 bins of H (64 VGPRs as doubles) are re-read from L2 at the multiply: with two windows' descriptors
    // live the block loop has no room to keep them
    const cd *Hg = reinterpret_cast<const cd *>(Hspec) + l;
    auto tw3 = [&](int p) { return t3[p]; };

    for (; b < nblocks; b += gridDim.x) {
        // window of real block rb: samples rb*S - pad + i, i = 0..N-1, through a descriptor that starts at the first
        // sample inside the buffer and ends with the buffer (rb past the stream: zero records)
        __amdgpu_buffer_rsrc_t rs[2];
        int shift[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const size_t rb = 2 * b + h;
            const size_t sh = rb * S >= (size_t)pad ? 0 : (size_t)pad - rb * S;
            const size_t first = rb * S + sh - pad;
            const size_t left = (rb < nblocks_real && in_elems > first) ? in_elems - first : 0;
            const size_t want = (size_t)N - sh;
            rs[h] = make_rsrc(in + first * EB, (unsigned)((left < want ? left : want) * EB));
            shift[h] = (int)sh;
        }
        cd v[16];
#pragma unroll
        for (int s = 0; s < 16; s++)
            v[s] = cd{RIO::load(rs[0], (l + LPF * s - shift[0]) * EB), RIO::load(rs[1], (l + LPF * s - shift[1]) * EB)};
        const cd *tf = tfg;
        asm volatile("" : "+v"(tf));
        xform<LOG2N, decltype(tw3), LPF>(v, lds, l, tw3, tf);
        const cd *Hb = Hg;
        asm volatile("" : "+v"(Hb));   // keeps the loads inside the loop (they are loop-invariant and would be hoisted back into registers)
        cd u[16];
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int k = P::NATURAL ? q : bin_of(q);
            const cd p = cmul(v[q], Hb[LPF * k]);
            u[k] = cd{p.x, -p.y};
        }
        asm volatile("" : "+v"(tf));
        xform<LOG2N, decltype(tw3), LPF>(u, lds, l, tw3, tf);
        if (DECIM) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const size_t rb = 2 * b + h;
                if (rb * S >= n_out) continue;                // uniform
                const size_t B0 = (rb * S) / M;
                const unsigned base = (unsigned)((rb * S) - B0 * M);
                const size_t room = n_dec > B0 ? n_dec - B0 : 0;
                const __amdgpu_buffer_rsrc_t wd = make_rsrc(out + B0 * EB, (unsigned)((room < (size_t)(N / 2 + 2) ? room : (size_t)(N / 2 + 2)) * EB));
                const size_t full_left = n_out - rb * S;
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const int row = LPF * (P::NATURAL ? q : bin_of(q));
                    if (row + LPF - 1 < Kov) continue;
                    const int i = l + row;
                    const unsigned t = base + (unsigned)(i - Kov) + 1u;
                    const unsigned qt = __umulhi(t, magic);
                    if (i >= Kov && (size_t)(i - Kov) < full_left && qt * M == t)
                        RIO::store(wd, (int)((qt - 1u) * (unsigned)EB), h == 0 ? u[q].x : -u[q].y, qs);
                }
            }
        } else {
        __amdgpu_buffer_rsrc_t ws[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const size_t rb = 2 * b + h;
            const size_t room = rb * S < n_out ? n_out - rb * S : 0;
            ws[h] = make_rsrc(out + (room ? rb * S : 0) * EB, (unsigned)((room < S ? room : S) * EB));
        }
        const unsigned vbase = (unsigned)(l - Kov) * (unsigned)EB;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int row = LPF * (P::NATURAL ? q : bin_of(q));
            if (row + LPF - 1 < Kov) continue;                // whole row dropped: uniform skip
            // y = conj(u): block 2b is its real part, block 2b+1 its imaginary part
            RIO::store(ws[0], (int)(vbase + (unsigned)row * (unsigned)EB), u[q].x, qs);
            RIO::store(ws[1], (int)(vbase + (unsigned)row * (unsigned)EB), -u[q].y, qs);
        }
        }
    }
}

template <int LOG2N, int IO>
int launch_real_ols(const void *in, size_t in_elems, void *out, size_t n_out, const void *Hspec, size_t K, const void *tw, size_t M, QShift qs, hipStream_t st)
{
    typedef OlsPlan<LOG2N> P;
    const size_t Km1 = K - 1;
    const size_t Kov = (Km1 + 15) / 16 * 16;
    if (Kov > (size_t)P::N / 2) { set_error("fir ols (real): K=%zu too long for %d-sample blocks", K, P::N); return PCX_ERR_UNSUPPORTED; }
    const size_t pad = Kov - Km1;
    const size_t S = P::N - Kov;
    const size_t nblocks_real = (n_out + S - 1) / S;
    const size_t nblocks = (nblocks_real + 1) / 2;
    const size_t lds = (size_t)(P::LDS_IMG + P::LDS_T2) * sizeof(cd);
    auto k = M > 1 ? fir_real_ols_kernel<LOG2N, IO, true> : fir_real_ols_kernel<LOG2N, IO, false>;
    const unsigned magic = M > 1 ? (unsigned)(((1ull << 32) + M - 1) / M) : 0u;
    if (lds > 64 * 1024) PCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    unsigned per_cu = (unsigned)(160 * 1024 / lds);
    const unsigned by_waves = 8u * 64u / P::LPF;
    if (per_cu > by_waves) per_cu = by_waves;
    if (per_cu < 1) per_cu = 1;
    const unsigned grid = PCX_ENV_INT("PCX_OVERSUB", 0) > 0 ? persistent_grid(nblocks, 256 * per_cu, 1) : rounds_grid(nblocks, 256 * per_cu, 4);   // as fir_cf64_ols
    hipLaunchKernelGGL(k, dim3(grid), dim3(P::LPF), lds, st, (const unsigned char *)in, in_elems, (unsigned char *)out, n_out,
                       (const double2 *)Hspec, (int)Kov, (int)pad, (const double2 *)tw, nblocks_real, n_out / M, (unsigned)M, magic, qs);
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}


// --------------------------------------------------------------------------------- //
// N = 4096 on the in-place transform pair (fft_f64.hpp, ip4096)
// --------------------------------------------------------------------------------- //
// Element types as the transform pair's kernels see them: a RAW register image of one sample (what a fetch-ahead keeps: one
// dword for the integer streams), its conversion, and the store (the integer stores: wrap_i32 above).
// fromQ under any of the three roundings (pcx_qformat.hpp from_q_bits) without a branch: floor + ((rem + add) >> n) with
// add = 0 (FLOOR), 2^(n-1) (ROUND), the remainder mask for negative values (TOWARD_ZERO: + 1 iff q < 0 and rem != 0)
struct QRound {
    int shift;
    unsigned mask, add_round, tz_mask;
    __device__ __forceinline__ explicit QRound(QShift s)
        : shift(s.shift), mask((1u << s.shift) - 1u), add_round(s.mode == PCX_Q_ROUND ? 1u << (s.shift - 1) : 0u),
          tz_mask(s.mode == PCX_Q_TOWARD_ZERO ? (1u << s.shift) - 1u : 0u) {}
    __device__ __forceinline__ int apply(int q) const
    {
        const unsigned rem = (unsigned)q & mask;
        return (q >> shift) + (int)((rem + add_round + ((unsigned)(q >> 31) & tz_mask)) >> shift);
    }
};
template <int IO>
struct IpIo;
template <>
struct IpIo<0> {
    static constexpr int EB = 16;
    static constexpr bool kAhead = false;
    typedef u32x4 Raw;
    template <int AUX> static __device__ __forceinline__ Raw load(__amdgpu_buffer_rsrc_t rs, int voff, int soff) { return __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, AUX); }
    static __device__ __forceinline__ cd cvt(Raw t) { return as_cd(t); }
    template <bool FLOORQ> static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t ws, int voff, int soff, cd y, QShift)
    {
        __builtin_amdgcn_raw_buffer_store_b128(as_u4(y), ws, voff, soff, kAuxStream);
    }
};
template <>
struct IpIo<1> {
    static constexpr int EB = 4;
    static constexpr bool kAhead = true;
    typedef unsigned Raw;
    template <int AUX> static __device__ __forceinline__ Raw load(__amdgpu_buffer_rsrc_t rs, int voff, int soff) { return __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, AUX); }
    static __device__ __forceinline__ cd cvt(Raw t) { return cd{(double)(short)(t & 0xffffu), (double)(short)(t >> 16)}; }
    template <bool FLOORQ> static __device__ __forceinline__ unsigned q(double d, QShift qs)
    {
        const int w = wrap_i32(d);
        return (unsigned)(FLOORQ ? (w >> qs.shift) : QRound(qs).apply(w)) & 0xffffu;
    }
    template <bool FLOORQ> static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t ws, int voff, int soff, cd y, QShift qs)
    {
        // FLOORQ: q >> 16 on both parts and the two halves packed = the HIGH halves of the two wrapped sums side by side: one v_perm_b32
        const unsigned packed = FLOORQ ? __builtin_amdgcn_perm((unsigned)wrap_i32(y.y), (unsigned)wrap_i32(y.x), 0x07060302u)
                                       : (q<false>(y.x, qs) | (q<false>(y.y, qs) << 16));
        __builtin_amdgcn_raw_buffer_store_b32(packed, ws, voff, soff, kAuxStream);
    }
};
template <>
struct IpIo<2> {
    static constexpr int EB = 2;
    static constexpr bool kAhead = true;
    typedef unsigned short Raw;
    template <int AUX> static __device__ __forceinline__ Raw load(__amdgpu_buffer_rsrc_t rs, int voff, int soff) { return __builtin_amdgcn_raw_buffer_load_b16(rs, voff, soff, AUX); }
    static __device__ __forceinline__ cd cvt(Raw t) { return cd{(double)(signed char)(t & 0xffu), (double)(signed char)((t >> 8) & 0xffu)}; }
    template <bool FLOORQ> static __device__ __forceinline__ unsigned q(double d, QShift qs)
    {
        const int w = (int)(short)wrap_i32(d);     // the 16-bit Q accumulator, sign-extended: QRound works on it as on a 32-bit one
        return (unsigned)(FLOORQ ? (w >> qs.shift) : QRound(qs).apply(w)) & 0xffu;
    }
    template <bool FLOORQ> static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t ws, int voff, int soff, cd y, QShift qs)
    {
        // FLOORQ: (q >> 8) & 0xff of the two 16-bit sums = byte 1 of each wrapped sum
        const unsigned packed = FLOORQ ? __builtin_amdgcn_perm((unsigned)wrap_i32(y.y), (unsigned)wrap_i32(y.x), 0x0c0c0501u)
                                       : (q<false>(y.x, qs) | (q<false>(y.y, qs) << 8));
        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)packed, ws, voff, soff, kAuxStream);
    }
};

// the window of block b through ONE range-checked descriptor: it starts at the first sample inside the buffer and ends with the
// window or the buffer, whichever comes first; lanes in front of it (block 0 when pad > 0: samples that only feed dropped outputs)
// wrap their offset out of range and read 0, as does everything behind the stream's end.  Rows 0 and 15 of the window are shared
// with the neighbouring blocks and keep the default cache policy, the rest is touched once.
template <int IO, bool ANY>
__device__ __forceinline__ void ip_fetch(typename IpIo<IO>::Raw (&raw)[16], const unsigned char *in, size_t in_elems, size_t b, size_t nblocks, size_t S, int pad,
                                         int idx2)
{
    typedef IpIo<IO> SIO;
    constexpr int EB = SIO::EB;
    const size_t shift = b * S >= (size_t)pad ? 0 : (size_t)pad - b * S;
    const size_t first = b * S + shift - pad;
    const size_t left = (b < nblocks && in_elems > first) ? in_elems - first : 0;     // a block behind the last: an empty descriptor
    const size_t want = (size_t)4096 - shift;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(in + (left ? first : 0) * EB, (unsigned)((left < want ? left : want) * EB));
    if (!ANY || shift == 0) {
        // the lane's offset once, the row in the instruction's scalar offset.  !ANY: the caller knows b >= 1 (shift == 0) and wants
        // exactly sixteen loads, no branch: the fetch-ahead inside the block loop (fir_cf64_ip_kernel says why)
        const int voff = idx2 * EB;
#pragma unroll
        for (int s = 0; s < 16; s++)
            raw[s] = (s == 0 || s == 15) ? SIO::template load<0>(rs, voff, s * 256 * EB) : SIO::template load<kAuxStream>(rs, voff, s * 256 * EB);
    } else {
        // block 0 of a filter with pad > 0: a lane in front of the buffer in row 0 is inside it from row 1 on, so the whole offset
        // goes through the range check as one number (a wrapped lane offset stays out of range whatever scalar offset is added)
#pragma unroll
        for (int s = 0; s < 16; s++) raw[s] = SIO::template load<0>(rs, (idx2 + 256 * s - (int)shift) * EB, 0);
    }
}

// the outputs of block b: u[q] = conj(y[256 bin_of(q) + idx2]); the first Kov of the block are dropped.  DECIM as in the N = 8192
// kernel below: full-rate index n with (n + 1) % M == 0 is stored at (n + 1) / M - 1.
template <int IO, bool DECIM, bool FLOORQ>
__device__ __forceinline__ void ip_store(const cd (&u)[16], unsigned char *out, size_t n_out, size_t n_dec, unsigned M, unsigned magic, size_t b, size_t S,
                                         int Kov, int idx2, QShift qs)
{
    // EVERY row is stored, by every lane, unconditionally: what must not land gets an offset outside the descriptor.  A store
    // behind a branch (a dropped row skipped, a lane masked) makes the number of stores in flight unknown to the compiler, and
    // the wait for the fetched-ahead samples at the foot of the block loop then has to drain them all.
    typedef IpIo<IO> SIO;
    constexpr int EB = SIO::EB;
    if (DECIM) {
        const size_t B0 = (b * S) / M;
        const unsigned base = (unsigned)((b * S) - B0 * M);
        const size_t room = n_dec > B0 ? n_dec - B0 : 0;
        const __amdgpu_buffer_rsrc_t ws = make_rsrc(out + B0 * EB, (unsigned)((room < (size_t)(4096 / 2 + 2) ? room : (size_t)(4096 / 2 + 2)) * EB));
        const size_t full_left = n_out - b * S;           // full-rate outputs this block may produce
        const unsigned left = full_left < 4096 ? (unsigned)full_left : 4096u;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int i = idx2 + 256 * bin_of(q);
            const unsigned t = base + (unsigned)(i - Kov) + 1u;
            const unsigned qt = __umulhi(t, magic);
            const bool keep = (unsigned)(i - Kov) < left && qt * M == t;       // (i < Kov wraps: not kept)
            SIO::template store<FLOORQ>(ws, keep ? (int)((qt - 1u) * (unsigned)EB) : -1, 0, cd{u[q].x, -u[q].y}, qs);
        }
    } else {
        const size_t room = n_out - b * S;
        const __amdgpu_buffer_rsrc_t ws = make_rsrc(out + b * S * EB, (unsigned)((room < S ? room : S) * EB));
        // outputs in front of Kov: the offset wraps, out of range, dropped.  (The row cannot ride in the scalar offset here: a lane
        // offset that wrapped stays out of range whatever is added to it, tools/soffset_lab.hip, and the lanes in front of Kov in
        // one row are valid in the next.)
        const unsigned vbase = (unsigned)(idx2 - Kov) * (unsigned)EB;
#pragma unroll
        for (int q = 0; q < 16; q++)
            SIO::template store<FLOORQ>(ws, (int)(vbase + (unsigned)(256 * bin_of(q)) * (unsigned)EB), 0, cd{u[q].x, -u[q].y}, qs);
    }
}

// FLOORQ: the integer streams' fromQ is the plain arithmetic shift by HALF the accumulator (PCX_Q_FLOOR + PCX_Q_FRAC_HALF_Q, the default
// reading: 16 of 32 bits for int16, 8 of 16 for int8; every other reading takes the branch-free general form) -- a kernel of its own, not
// a branch in front of the stores: two alternative store sequences that join make the compiler's count of the stores in flight
// inexact, and its wait for the fetched-ahead loads at the foot of the loop then drains them.
// DYN: 512 persistent workgroups (two per CU) that DRAW their blocks (pcx_sched.hpp AheadDealer: the next block must be known at
// the head of a block, for the fetch-ahead, so blocks are dealt in strided pairs with the draw one block ahead) instead of ~4,400
// workgroups of four blocks each: the tables (H, the pass factors: 124 registers from L2) are loaded once per 34 blocks instead of
// once per 4, and the launch no longer ends on a half-empty ninth round of workgroups.
template <int IO, bool DECIM, bool FLOORQ, int PART = 0, bool DYN = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void fir_cf64_ip_kernel(
    const unsigned char *__restrict__ in, size_t in_elems, unsigned char *__restrict__ out, size_t n_out, size_t n_dec, unsigned M, unsigned magic,
    const double2 *__restrict__ Hspec, int Kov, int pad, const double2 *__restrict__ twtab, size_t nblocks, QShift qs, SchedState *__restrict__ sched)
{
    typedef IpIo<IO> SIO;
    constexpr bool PRIO = true;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cd *lds = reinterpret_cast<cd *>(smem_raw);
    __shared__ unsigned sched_slot;
    const int l = threadIdx.x;
    const size_t S = (size_t)(4096 - Kov);
    size_t b = blockIdx.x;
    AheadDealer deal;
    if (DYN) {
        if (!deal.begin(sched, &sched_slot, nblocks, l)) { deal.finish(l); return; }
        b = deal.block();
    } else if (b >= nblocks) return;
    const ip4096::Lane L = ip4096::make_lane(l);
    typename SIO::Raw raw[16];
    ip_fetch<IO, true>(raw, in, in_elems, b, nblocks, S, pad, L.idx2);       // the first block's samples ahead of the tables
    const cd *tab = reinterpret_cast<const cd *>(twtab);
    if (l < 240) lds[ip4096::kT2 + l] = tab[l];               // first read behind the first exchange's barrier
    ip4096::LaneTw pw;
    pw.load(tab, L);
    cd H[16];
#pragma unroll
    for (int q = 0; q < 16; q++) H[q] = reinterpret_cast<const cd *>(Hspec)[L.k0 + 256 * bin_of(q)];
    // every table register is "used" once HERE: the waits for the table loads then sit in front of the loop.  Left to the first
    // use inside it, they are vmcnt waits in the loop body, and from the second block on what such a wait drains is the previous
    // block's stores (vmcnt counts loads and stores in one queue, oldest first).
    pw.opaque();
#pragma unroll
    for (int q = 0; q < 16; q += 4)
        asm volatile("" : "+v"(H[q].x), "+v"(H[q].y), "+v"(H[q + 1].x), "+v"(H[q + 1].y), "+v"(H[q + 2].x), "+v"(H[q + 2].y), "+v"(H[q + 3].x), "+v"(H[q + 3].y));

    // the block's samples are converted at the FOOT of the loop, behind the stores: the fetch-ahead is older than the stores in
    // the vmcnt queue, so the conversion waits for the loads alone.  (Converted at the head, the compiler carried the fetched
    // registers over the back edge through copies behind s_waitcnt vmcnt(0): every block waited for its own stores.)
    cd v[16];
#pragma unroll
    for (int s = 0; s < 16; s++) v[s] = SIO::cvt(raw[s]);
    for (;;) {
        // the next block's samples (nb >= gridDim.x >= 1: never block 0); behind the last block an empty descriptor, no branch: the
        // compiler must be able to COUNT the loads and stores in flight, or its wait for these loads at the foot drains the stores too
        size_t nb = b + gridDim.x;
        const bool more = DYN ? deal.next(&nb) : nb < nblocks;
        if (!more) nb = nblocks;
        // (the dealer's "lane 0" is asked for through L.idx2, which is 0 for lane 0 alone and lives in a register across the loop anyway:
        // threadIdx.x itself did not, and came back from scratch behind an s_waitcnt vmcnt(0) at the head of every block)
        if (DYN) deal.draw(L.idx2);    // (ahead of the fetch in the in-order vmcnt queue.  One conditional operation there, younger than
                                       //  nothing the foot waits for: it makes that wait a notch stronger, it cannot make it drain the stores)
        if (SIO::kAhead) ip_fetch<IO, false>(raw, in, in_elems, nb, nblocks, S, pad, L.idx2);
        // Wave priority rises with the progress through a block: 1 for the forward transform, 2 from the backward one to the last
        // store, 0 again for the foot (the fetched-ahead samples' conversion).  A SIMD holds one wave of each of the CU's two
        // workgroups; without it the two drift into the same phase and wait on LDS together.  Interleaved on two boxes
        // (profiles/r06/ab_ip64_prio.txt): none 0.2873, second half at 1: 0.2818, at 3: 0.2775, 1 then 2: 0.2773 ms.
        if (PRIO) __builtin_amdgcn_s_setprio(1);
        ip4096::forward<PART>(v, lds, L, pw);
        // u = conj(X .* H), in the natural register order the backward passes start from
        cd u[16];
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const cd p = cmul(v[q], H[q]);
            u[bin_of(q)] = cd{p.x, -p.y};
        }
        if (DYN) deal.publish(L.idx2); // the backward passes' barriers come behind it
        if (PRIO) __builtin_amdgcn_s_setprio(2);
        ip4096::backward<PART>(u, lds, L, pw);
        ip_store<IO, DECIM, FLOORQ>(u, out, n_out, n_dec, M, magic, b, S, Kov, L.idx2, qs);
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        if (!more) break;
        if (DYN) (void)deal.advance();
        b = nb;
        if (!SIO::kAhead) ip_fetch<IO, false>(raw, in, in_elems, b, nblocks, S, pad, L.idx2);
#pragma unroll
        for (int s = 0; s < 16; s++) v[s] = SIO::cvt(raw[s]);
    }
    if (DYN) deal.finish(L.idx2);
}

template <int IO>
int launch_ip(const void *in, size_t in_elems, void *out, size_t n_out, const void *Hspec, size_t K, const void *tw, size_t M, QShift qs, void *sched,
              hipStream_t st)
{
    const size_t Km1 = K - 1;
    const size_t Kov = (Km1 + 15) / 16 * 16;
    if (Kov > 2048) { set_error("fir ols f64: K=%zu too long for 4096-sample blocks", K); return PCX_ERR_UNSUPPORTED; }
    const size_t pad = Kov - Km1;
    const size_t S = 4096 - Kov;
    const size_t nblocks = (n_out + S - 1) / S;
    const size_t lds = (size_t)ip4096::kLdsSlots * sizeof(cd);
    const bool floorq = IO == 0 || (qs.mode == PCX_Q_FLOOR && qs.shift == (IO == 1 ? 16 : 8));
    // dealt from 512 persistent workgroups when the call is long enough for it to matter and nothing asks for a small grid (a
    // link-bound host call, pcx_internal.hpp); PCX_SCHED_STATIC (diagnostic library) keeps the grid-stride walk for A/B
    // (M == 1: with the decimating store's index arithmetic on top the dealt form spills a register, and reloads it behind vmcnt(0))
    const bool dyn = sched && M == 1 && nblocks > 4 * 512 && nblocks < ((size_t)1 << 31) && !g_link_grid && !PCX_ENV_SET("PCX_SCHED_STATIC");
    auto k = M > 1 ? (floorq ? fir_cf64_ip_kernel<IO, true, true> : fir_cf64_ip_kernel<IO, true, IO == 0>)
                   : (floorq ? fir_cf64_ip_kernel<IO, false, true> : fir_cf64_ip_kernel<IO, false, IO == 0>);
    if (dyn) k = floorq ? fir_cf64_ip_kernel<IO, false, true, 0, true> : fir_cf64_ip_kernel<IO, false, IO == 0, 0, true>;
#ifdef PCX_DIAG
    // timing-only parts of the transform pair (fft_f64.hpp PART; wrong outputs): PCX_IP64_PART=1 no barriers, 2 arithmetic only
    if (IO == 1 && M == 1 && floorq && PCX_ENV_INT("PCX_IP64_PART", 0) == 1) k = dyn ? fir_cf64_ip_kernel<1, false, true, 1, true> : fir_cf64_ip_kernel<1, false, true, 1>;
    if (IO == 1 && M == 1 && floorq && PCX_ENV_INT("PCX_IP64_PART", 0) == 2) k = dyn ? fir_cf64_ip_kernel<1, false, true, 2, true> : fir_cf64_ip_kernel<1, false, true, 2>;
#endif
    const unsigned magic = M > 1 ? (unsigned)(((1ull << 32) + M - 1) / M) : 0u;
    PCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // two workgroups per CU (73 KB of LDS each); about four blocks per workgroup whatever the call (pcx_internal.hpp rounds_grid:
    // +5 % over one round at 64 Mi samples, tools/ab_oversub.sh)
    const unsigned grid = dyn ? 512u : PCX_ENV_INT("PCX_OVERSUB", 0) > 0 ? persistent_grid(nblocks, 256 * 2, 1) : rounds_grid(nblocks, 256 * 2, 4);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, (const unsigned char *)in, in_elems, (unsigned char *)out, n_out, n_out / M, (unsigned)M, magic,
                       (const double2 *)Hspec, (int)Kov, (int)pad, (const double2 *)tw, nblocks, qs, (SchedState *)sched);
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}

// REAL streams on the in-place pair: two consecutive real blocks as the real and imaginary part of one complex block, as
// fir_real_ols_kernel below.  H is re-read from L2 at the multiply (two windows' descriptors and offsets are live across the block).
// DYN: blocks drawn one at a time from 512 persistent workgroups (pcx_sched.hpp BlockDealer; no fetch-ahead here, so the next block need
// only be known at the end of the current one) instead of the grid-stride walk over ~4 blocks per workgroup
template <int IO, bool DECIM, bool DYN = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void fir_real_ip_kernel(
    const unsigned char *__restrict__ in, size_t in_elems, unsigned char *__restrict__ out, size_t n_out, const double2 *__restrict__ Hspec,
    int Kov, int pad, const double2 *__restrict__ twtab, size_t nblocks_real, size_t n_dec, unsigned M, unsigned magic, QShift qs,
    SchedState *__restrict__ sched)
{
    typedef RealIo<IO> RIO;
    constexpr int EB = RIO::EB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cd *lds = reinterpret_cast<cd *>(smem_raw);
    const int l = threadIdx.x;
    const size_t S = (size_t)(4096 - Kov);
    const size_t nblocks = (nblocks_real + 1) / 2;
    __shared__ unsigned sched_slot;
    const ip4096::Lane L = ip4096::make_lane(l);
    BlockWalk<DYN> walk;
    if (!walk.begin(sched, &sched_slot, nblocks, L.idx2)) { walk.finish(L.idx2); return; }     // (L.idx2 == 0 for lane 0 alone: the dealer's "lane")
    const cd *tab = reinterpret_cast<const cd *>(twtab);
    if (l < 240) lds[ip4096::kT2 + l] = tab[l];
    ip4096::LaneTw pw;
    pw.load(tab, L);
    const cd *Hg = reinterpret_cast<const cd *>(Hspec) + L.k0;
    pw.opaque();     // the waits for the table loads in front of the loop (fir_cf64_ip_kernel)

    for (;;) {
        const size_t b = walk.block();
        __amdgpu_buffer_rsrc_t rs[2];
        int shift[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const size_t rb = 2 * b + h;
            const size_t sh = rb * S >= (size_t)pad ? 0 : (size_t)pad - rb * S;
            const size_t first = rb * S + sh - pad;
            const size_t left = (rb < nblocks_real && in_elems > first) ? in_elems - first : 0;
            const size_t want = (size_t)4096 - sh;
            rs[h] = make_rsrc(in + first * EB, (unsigned)((left < want ? left : want) * EB));
            shift[h] = (int)sh;
        }
        cd v[16];
#pragma unroll
        for (int s = 0; s < 16; s++)
            v[s] = cd{RIO::load(rs[0], (L.idx2 + 256 * s - shift[0]) * EB), RIO::load(rs[1], (L.idx2 + 256 * s - shift[1]) * EB)};
        walk.draw(L.idx2);                  // (every load of the block has been issued; its value is parked in front of the backward passes' barriers)
        __builtin_amdgcn_s_setprio(1);      // (priority rising with the progress through a block: fir_cf64_ip_kernel says why)
        ip4096::forward(v, lds, L, pw);
        const cd *Hb = Hg;
        asm volatile("" : "+v"(Hb));   // keeps the loads inside the loop (they are loop-invariant and would be hoisted back into registers)
        cd u[16];
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const cd p = cmul(v[q], Hb[256 * bin_of(q)]);
            u[bin_of(q)] = cd{p.x, -p.y};
        }
        walk.publish(L.idx2);
        __builtin_amdgcn_s_setprio(2);
        ip4096::backward(u, lds, L, pw);
        if (DECIM) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const size_t rb = 2 * b + h;
                if (rb * S >= n_out) continue;                // uniform
                const size_t B0 = (rb * S) / M;
                const unsigned base = (unsigned)((rb * S) - B0 * M);
                const size_t room = n_dec > B0 ? n_dec - B0 : 0;
                const __amdgpu_buffer_rsrc_t wd = make_rsrc(out + B0 * EB, (unsigned)((room < (size_t)(4096 / 2 + 2) ? room : (size_t)(4096 / 2 + 2)) * EB));
                const size_t full_left = n_out - rb * S;
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const int row = 256 * bin_of(q);
                    if (row + 255 < Kov) continue;
                    const int i = L.idx2 + row;
                    const unsigned t = base + (unsigned)(i - Kov) + 1u;
                    const unsigned qt = __umulhi(t, magic);
                    if (i >= Kov && (size_t)(i - Kov) < full_left && qt * M == t)
                        RIO::store(wd, (int)((qt - 1u) * (unsigned)EB), h == 0 ? u[q].x : -u[q].y, qs);
                }
            }
        } else {
            __amdgpu_buffer_rsrc_t ws[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const size_t rb = 2 * b + h;
                const size_t room = rb * S < n_out ? n_out - rb * S : 0;
                ws[h] = make_rsrc(out + (room ? rb * S : 0) * EB, (unsigned)((room < S ? room : S) * EB));
            }
            const unsigned vbase = (unsigned)(L.idx2 - Kov) * (unsigned)EB;
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int row = 256 * bin_of(q);
                if (row + 255 < Kov) continue;                // whole row dropped: uniform skip
                // y = conj(u): block 2b is its real part, block 2b+1 its imaginary part
                RIO::store(ws[0], (int)(vbase + (unsigned)row * (unsigned)EB), u[q].x, qs);
                RIO::store(ws[1], (int)(vbase + (unsigned)row * (unsigned)EB), -u[q].y, qs);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if (!walk.advance()) break;
    }
    walk.finish(L.idx2);
}

template <int IO>
int launch_real_ip(const void *in, size_t in_elems, void *out, size_t n_out, const void *Hspec, size_t K, const void *tw, size_t M, QShift qs, void *sched,
                   hipStream_t st)
{
    const size_t Km1 = K - 1;
    const size_t Kov = (Km1 + 15) / 16 * 16;
    if (Kov > 2048) { set_error("fir ols (real): K=%zu too long for 4096-sample blocks", K); return PCX_ERR_UNSUPPORTED; }
    const size_t pad = Kov - Km1;
    const size_t S = 4096 - Kov;
    const size_t nblocks_real = (n_out + S - 1) / S;
    const size_t nblocks = (nblocks_real + 1) / 2;
    const size_t lds = (size_t)ip4096::kLdsSlots * sizeof(cd);
    const bool dyn = sched && M == 1 && nblocks > 4 * 512 && nblocks < ((size_t)1 << 31) && !g_link_grid && !PCX_ENV_SET("PCX_SCHED_STATIC");
    auto k = M > 1 ? fir_real_ip_kernel<IO, true> : dyn ? fir_real_ip_kernel<IO, false, true> : fir_real_ip_kernel<IO, false>;
    const unsigned magic = M > 1 ? (unsigned)(((1ull << 32) + M - 1) / M) : 0u;
    PCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned grid = dyn ? 512u : PCX_ENV_INT("PCX_OVERSUB", 0) > 0 ? persistent_grid(nblocks, 256 * 2, 1) : rounds_grid(nblocks, 256 * 2, 4);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, (const unsigned char *)in, in_elems, (unsigned char *)out, n_out, (const double2 *)Hspec, (int)Kov,
                       (int)pad, (const double2 *)tw, nblocks_real, n_out / M, (unsigned)M, magic, qs, (SchedState *)sched);
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}

}  // namespace

// log2n 12 (K <= 2049: the in-place kernels) or 13; Hspec = FFT_N(h)/N in double, natural bin order; tw = make_tw_ols64(log2n) (pcx_api.hip).  io: 0 complex_float64,
// 1 complex_int16, 2 complex_int8 (h = the Q-format integer taps; see StreamIo).  n_out = full-rate outputs; M > 1 keeps one in M
int launch_fir_cf64_ols(const void *in, size_t in_elems, void *out, size_t n_out, const void *Hspec, size_t K, int log2n,
                        const void *tw, int io, size_t M, QShift qs, hipStream_t st, void *sched)
{
    if (n_out == 0) return PCX_OK;
    if (M < 1 || M > 65535) { set_error("fir ols f64: decimation %zu outside 1..65535", M); return PCX_ERR_UNSUPPORTED; }
    if (log2n == 12)
        return io == 0   ? launch_ip<0>(in, in_elems, out, n_out, Hspec, K, tw, M, qs, sched, st)
               : io == 1 ? launch_ip<1>(in, in_elems, out, n_out, Hspec, K, tw, M, qs, sched, st)
                         : launch_ip<2>(in, in_elems, out, n_out, Hspec, K, tw, M, qs, sched, st);
    if (log2n == 13)
        return io == 0   ? launch_ols<13, 0>(in, in_elems, out, n_out, Hspec, K, tw, M, qs, st)
               : io == 1 ? launch_ols<13, 1>(in, in_elems, out, n_out, Hspec, K, tw, M, qs, st)
                         : launch_ols<13, 2>(in, in_elems, out, n_out, Hspec, K, tw, M, qs, st);
    set_error("fir ols f64: no plan for log2(N) = %d", log2n);
    return PCX_ERR_UNSUPPORTED;
}

}  // namespace pcx

namespace pcx {
// REAL streams on the same pipeline (real taps): io 0 float64, 1 int16, 2 int8, 3 float32; log2n 12 (K <= 2049) or 13 (K <= 4097);
// n_out = full-rate outputs, M > 1 keeps one in M
int launch_fir_real_ols(const void *in, size_t in_elems, void *out, size_t n_out, const void *Hspec, size_t K, int log2n, const void *tw,
                        int io, size_t M, QShift qs, hipStream_t st, void *sched)
{
    if (n_out == 0) return PCX_OK;
    if (M < 1 || M > 65535) { set_error("fir ols (real): decimation %zu outside 1..65535", M); return PCX_ERR_UNSUPPORTED; }
    if (log2n == 12)
        return io == 0   ? launch_real_ip<0>(in, in_elems, out, n_out, Hspec, K, tw, M, qs, sched, st)
               : io == 1 ? launch_real_ip<1>(in, in_elems, out, n_out, Hspec, K, tw, M, qs, sched, st)
               : io == 2 ? launch_real_ip<2>(in, in_elems, out, n_out, Hspec, K, tw, M, qs, sched, st)
                         : launch_real_ip<3>(in, in_elems, out, n_out, Hspec, K, tw, M, qs, sched, st);
    if (log2n == 13)
        return io == 0   ? launch_real_ols<13, 0>(in, in_elems, out, n_out, Hspec, K, tw, M, qs, st)
               : io == 1 ? launch_real_ols<13, 1>(in, in_elems, out, n_out, Hspec, K, tw, M, qs, st)
               : io == 2 ? launch_real_ols<13, 2>(in, in_elems, out, n_out, Hspec, K, tw, M, qs, st)
                         : launch_real_ols<13, 3>(in, in_elems, out, n_out, Hspec, K, tw, M, qs, st);
    set_error("fir ols (real): no plan for log2(N) = %d", log2n);
    return PCX_ERR_UNSUPPORTED;
}
}  // namespace pcx
