// fir_ols_r16.hip -- overlap-save /comms/fir_filter (complex_float32, M = L = 1) on transform
// sizes other than 4096: the same  y = IFFT(FFT(block) .* H)  evaluation of FIRFilter.cpp:294-300
// as fir_ols.hip, built from the radix-16 family passes of fft_r16.hip.
//
// Why more than one block size: a workgroup holds one block (N*8.5 bytes of LDS, N/16 lanes) and
// alternates between waiting for its 8N-byte window and transforming it, so a CU overlaps memory
// and arithmetic only ACROSS workgroups.  N = 4096 fits 4 workgroups per CU, N = 2048 fits 8,
// N = 1024 fits 16 (one wave each) -- more, smaller, independent customers of the two resources --
// at the price of a larger overlap fraction (K-1)/N of re-read samples.  Measured on MI355X at
// K = 255, 64 Mi samples (tools/ab_ols.py): dedicated 4096 kernel 0.2246 ms, this file's 4096 plan
// 0.2291, 2048 0.2413, 1024 0.2723, 8192 0.3100 -- the finer grain does NOT pay (more blocks, more
// re-read overlap, 8 spilled VGPRs at 2048), so K <= 2049 stays on fir_ols.hip and this file serves
// the taps too long for it: N = 8192 for K-1 <= 4096, N = 16384 for K-1 <= 8192 (pcx_api.hip).
//
// Round 6, the long-tap plans under the counters (K = 4097 on 8192-sample blocks, 64 Mi samples, profiles/r06/fir4097_*): 0.48 ms;
// HBM traffic 1.08 GB read + 0.54 GB written = 3.4 TB/s (every window is fetched in full: at this tap count half of it is overlap,
// and the second fetch misses the caches); VALU 39 % of SIMD time, 21 % of the LDS cycles bank conflicts, waves waiting on LDS 18 %
// of their time.  No single roof: half-overlap blocks cost twice the headline's arithmetic and 24 B of traffic per output.
// Tried and dropped: the next window fetched ahead under a 128-VGPR budget (H and the pass-256 factors then have to be re-read
// from L2 per block to make room): 140 -> 121 Gsamples/s at 4097 taps, 90 -> 81 at 8193; the second half of a block at wave
// priority 1 (what gave the 4096-sample resamplers 3-6 %): -1.5 % / +-0 here (four waves per SIMD of two workgroups).
//
// Lane l of a block holds x[l + s*LPF], s = 0..15, LPF = N/16.  A forward transform leaves
// X[l + k*LPF] in the lane (k = register index for a final radix-R pass, bin_of(q) after a
// final radix-16 pass): exactly the layout the next transform's first pass wants, so the
// spectrum is multiplied by the lane's 16 bins of H (32 VGPRs, loaded once per persistent
// workgroup) and inverse-transformed (conj . FFT . conj) without leaving registers.
#include "fft4096.hpp"
#include <cstdio>
#include <cstdlib>

#include "pcx_internal.hpp"

namespace pcx {

namespace {
using namespace fft4k;

template <int LOG2N>
struct OlsPlan {
    static constexpr int N = 1 << LOG2N;
    static constexpr int LPF = N / 16;                    // lanes per block = workgroup size
    static constexpr int A = LOG2N / 4;                   // radix-16 passes (2 or 3 here)
    static constexpr int R = 1 << (LOG2N % 4);            // final radix (1: none)
    static constexpr int NTWF = R > 1 ? (16 / R) * (R - 1) : 0;
    static constexpr int LDS_IMG = N + N / 16;            // padded image, in complex elements
    static constexpr int LDS_T2 = 240;                    // pass Ns=16 table behind the image
    // device table layout of make_tw_r16(): [15][16], [15][256] (A >= 3), [NTWF][LPF]
    static constexpr int T3_OFF = 15 * 16;
    static constexpr int TF_OFF = T3_OFF + (A >= 3 ? 15 * 256 : 0);
    static constexpr bool NATURAL = R > 1;                // register q holds bin q*LPF + l (else bin_of(q)*LPF + l)
};

// forward DFT_N of the block held as v[s] = x[l + s*LPF]; on exit v[q] = X[l + LPF*(NATURAL ? q : bin_of(q))]
// TFS: stride of the final-pass twiddles behind `tf` (1: the lane's register copy; LPF: the device table itself,
// re-read at each use by the 16384-sample plan, which has no registers left for them)
template <int LOG2N, int TFS = 1>
__device__ __forceinline__ void xform(cf (&v)[16], cf *lds, int l, const LaneTw &t3, const cf *tf)
{
    typedef OlsPlan<LOG2N> P;
    constexpr int LPF = P::LPF, A = P::A, R = P::R;
    // ---- pass Ns = 1 ----
    fft16_plain(v);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; q++) lds[17 * l + bin_of(q)] = v[q];
    // ---- pass Ns = 16: twiddles depend on l & 15 only, broadcast reads of the LDS table ----
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 16; s++) v[s] = lds[l + (l >> 4) + s * (LPF + LPF / 16)];
    {
        LaneTw t2;
        const cf *tb = lds + P::LDS_IMG + (l & 15);
#pragma unroll
        for (int p = 0; p < 3; p++) t2.a[p] = tb[p * 16];
#pragma unroll
        for (int p = 0; p < 12; p++) t2.c[p] = tb[(3 + p) * 16];
        fft16_tw(v, t2);
    }
    __syncthreads();
    {
        const int wb = (l >> 4) * 272 + (l & 15);
#pragma unroll
        for (int q = 0; q < 16; q++) lds[wb + 17 * bin_of(q)] = v[q];
    }
    // ---- pass Ns = 256 ----
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
    // ---- final radix-R pass ----
    if (R > 1) {
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 16; s++) v[s] = lds[padi(l + s * LPF)];
        constexpr int G = 16 / R;
#pragma unroll
        for (int t = 0; t < G; t++) {
#pragma unroll
            for (int r = 1; r < R; r++) v[t + r * G] = cmul1(v[t + r * G], tf[(t * (R - 1) + (r - 1)) * TFS]);
            if (R == 2) {
                const cf a = v[t], b = v[t + G];
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

// Block geometry as in fir_ols.hip: Kov >= K-1 (a multiple of 16) outputs dropped per block, the
// window of block b starts `pad` = Kov-(K-1) samples before sample b*S, S = N - Kov.
template <int LOG2N, int DIAG>
__global__ __launch_bounds__(OlsPlan<LOG2N>::LPF, 4) void fir_cf32_ols_r16_kernel(const float2 *__restrict__ in, size_t in_elems,
                                                                                 float2 *__restrict__ out, size_t n_out,
                                                                                 const float2 *__restrict__ Hspec, int Kov, int pad,
                                                                                 const float2 *__restrict__ twtab, size_t first_full,
                                                                                 size_t nfull, size_t nblocks)
{
    typedef OlsPlan<LOG2N> P;
    constexpr int N = P::N, LPF = P::LPF;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf *lds = reinterpret_cast<cf *>(smem_raw);
    const int l = threadIdx.x;
    const size_t S = (size_t)(N - Kov);
    size_t b = blockIdx.x;
    if (b >= nblocks) return;
    const cf *tab = reinterpret_cast<const cf *>(twtab);
    // loop invariants: final-pass / pass-256 twiddles and the lane's 16 bins of H in registers,
    // the 16-row table of the Ns = 16 pass in LDS
    LaneTw t3;
    if (P::A >= 3) {
#pragma unroll
        for (int p = 0; p < 3; p++) t3.a[p] = tab[P::T3_OFF + p * 256 + (l & 255)];
#pragma unroll
        for (int p = 0; p < 12; p++) t3.c[p] = tab[P::T3_OFF + (3 + p) * 256 + (l & 255)];
    }
    constexpr bool TFG = LOG2N >= 14;
    constexpr int TFS = TFG ? LPF : 1;
    cf tf[P::NTWF > 0 && !TFG ? P::NTWF : 1];
    if (!TFG) {
#pragma unroll
        for (int p = 0; p < P::NTWF; p++) tf[p] = tab[P::TF_OFF + p * LPF + l];
    }
    for (int i = l; i < P::LDS_T2; i += LPF) lds[P::LDS_IMG + i] = tab[i];
    // 8192- / 16384-sample plans run 8 / 16 waves per block at <= 128 VGPRs: the lane's 16 bins of H (32 VGPRs) are
    // re-read from L2 at the multiply instead of living in registers across the block loop (spilled VGPRs 24 -> see
    // the resource report; PCX_OLS_HREG=1 at build time keeps them in registers for A/B)
#ifdef PCX_OLS_HREG
    constexpr bool HG = false;
#else
    constexpr bool HG = LOG2N >= 13;
#endif
    const cf *Hg = reinterpret_cast<const cf *>(Hspec) + l;
    cf H[HG ? 1 : 16];
    if (!HG) {
#pragma unroll
        for (int k = 0; k < 16; k++) H[k] = Hg[LPF * k];
    }

    for (; b < nblocks; b += gridDim.x) {
        cf v[16];
        const size_t blk = DIAG == 1 ? first_full : b;
        if (blk >= first_full && blk < nfull) {
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(in + blk * S - pad, N * 8);
#pragma unroll
            for (int s = 0; s < 16; s++) {
                const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, l * 8, s * LPF * 8, 0);
                v[s] = cf{__uint_as_float(t.x), __uint_as_float(t.y)};
            }
        } else {
            // ragged: block 0 when pad > 0 (the samples before the buffer only feed dropped outputs
            // and read as 0 through the range check) and the tail
            const size_t shift = blk * S >= (size_t)pad ? 0 : (size_t)pad - blk * S;
            const size_t first = blk * S + shift - pad;
            const size_t left = in_elems > first ? in_elems - first : 0;
            const size_t want = (size_t)N - shift;
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(in + first, (unsigned)((left < want ? left : want) * 8));
#pragma unroll
            for (int s = 0; s < 16; s++) {
                const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, (l + LPF * s - (int)shift) * 8, 0, 0);
                v[s] = cf{__uint_as_float(t.x), __uint_as_float(t.y)};
            }
        }
        const cf *tfp = TFG ? tab + P::TF_OFF + l : tf;
        if (TFG) asm volatile("" : "+v"(tfp));
        if (DIAG != 2) xform<LOG2N, TFS>(v, lds, l, t3, tfp);
        // u = conj(X .* H) in the first-pass layout of the next transform (register k <- bin k*LPF + l)
        const cf *Hb = Hg;
        if (HG) asm volatile("" : "+v"(Hb));   // loop-invariant loads would be hoisted straight back into registers
        cf u[16];
#pragma unroll
        for (int q = 0; q < 16; q += 2) {
            const int k0 = P::NATURAL ? q : bin_of(q), k1 = P::NATURAL ? q + 1 : bin_of(q + 1);
            u[k0] = v[q];
            u[k1] = v[q + 1];
            if (HG) cmul2_conj(u[k0], u[k1], Hb[LPF * k0], Hb[LPF * k1]);
            else cmul2_conj(u[k0], u[k1], H[k0], H[k1]);
        }
        if (TFG) asm volatile("" : "+v"(tfp));
        if (DIAG != 2) xform<LOG2N, TFS>(u, lds, l, t3, tfp);
        // time sample i = l + k*LPF of the block is output b*S + i - Kov; i < Kov wraps past
        // num_records and is dropped by the range check, as are outputs past n_out
        const size_t bo = DIAG == 1 ? 0 : b;
        const size_t room = n_out - bo * S;
        const __amdgpu_buffer_rsrc_t ws = make_rsrc(out + bo * S, (unsigned)((room < S ? room : S) * 8));
        const unsigned vbase = (unsigned)(l - Kov) * 8u;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int row = LPF * (P::NATURAL ? q : bin_of(q));
            if (row + LPF - 1 < Kov) continue;                // whole row dropped: uniform skip
            store_cf<0>(ws, vbase + (unsigned)row * 8u, cf{u[q].x, -u[q].y});
        }
    }
}

template <int LOG2N>
int launch_ols(const void *in, size_t in_elems, void *out, size_t n_out, const void *Hspec, size_t K, const void *tw,
               hipStream_t st, int diag)
{
    typedef OlsPlan<LOG2N> P;
    const size_t Km1 = K - 1;
    const size_t Kov = (Km1 + 15) / 16 * 16;
    if (Kov > (size_t)P::N / 2) { set_error("fir ols: K=%zu too long for %d-sample blocks", K, P::N); return PCX_ERR_UNSUPPORTED; }
    const size_t pad = Kov - Km1;
    const size_t S = P::N - Kov;
    const size_t nblocks = (n_out + S - 1) / S;
    const size_t first_full = pad > 0 ? 1 : 0;
    size_t nfull = n_out / S;
    while (nfull > first_full && (nfull - 1) * S - pad + P::N > in_elems) nfull--;
    if (nfull < first_full) nfull = first_full;
    const size_t lds = (size_t)(P::LDS_IMG + P::LDS_T2) * sizeof(cf);
#ifdef PCX_DIAG
    auto k = diag == 1 ? fir_cf32_ols_r16_kernel<LOG2N, 1> : diag == 2 ? fir_cf32_ols_r16_kernel<LOG2N, 2> : fir_cf32_ols_r16_kernel<LOG2N, 0>;
#else
    (void)diag;   // the timing-only instantiations exist in the diagnostic library only
    auto k = fir_cf32_ols_r16_kernel<LOG2N, 0>;
#endif
    if (lds > 64 * 1024) PCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // resident workgroups per CU: LDS (160 KiB) and 16 waves of <= 128 VGPRs
    unsigned per_cu = (unsigned)(160 * 1024 / lds);
    const unsigned by_waves = 16u * 64u / P::LPF;
    if (per_cu > by_waves) per_cu = by_waves;
    const long r16_rounds = PCX_ENV_INT("PCX_R16_ROUNDS", 0);   // (diagnostic library: blocks per workgroup instead of the fixed factor, A/B)
    const unsigned grid = r16_rounds > 0 ? rounds_grid(nblocks, 256 * per_cu, (unsigned)r16_rounds)
                                         : persistent_grid(nblocks, 256 * per_cu, 4);   // four queued per slot: K = 4097 132 -> 141 Gsamples/s, 8193 +-0 (PCX_OVERSUB A/B)
    hipLaunchKernelGGL(k, dim3(grid), dim3(P::LPF), lds, st, (const float2 *)in, in_elems, (float2 *)out, n_out,
                       (const float2 *)Hspec, (int)Kov, (int)pad, (const float2 *)tw, first_full, nfull, nblocks);
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}

}  // namespace

// log2n in {10 .. 14}; Hspec = FFT_N(h)/N, tw = make_tw_r16(log2n) (pcx_api.hip)
int launch_fir_cf32_ols_r16(const void *in, size_t in_elems, void *out, size_t n_out, const void *Hspec, size_t K, int log2n,
                            const void *tw, hipStream_t st)
{
    if (n_out == 0) return PCX_OK;
    // PCX_OLS_DIAG (libpcx_hip_diag.so only; timing-only, wrong outputs): 1 compute floor, 2 memory floor
    const int diag = (int)PCX_ENV_INT("PCX_OLS_DIAG", 0);
#ifdef PCX_DIAG
    if (diag) fprintf(stderr, "pcx(diag): PCX_OLS_DIAG=%d selects a TIMING-ONLY build of the overlap-save FIR: its outputs are wrong\n", diag);
#endif
    switch (log2n) {
    case 10: return launch_ols<10>(in, in_elems, out, n_out, Hspec, K, tw, st, diag);
    case 11: return launch_ols<11>(in, in_elems, out, n_out, Hspec, K, tw, st, diag);
    case 12: return launch_ols<12>(in, in_elems, out, n_out, Hspec, K, tw, st, diag);
    case 13: return launch_ols<13>(in, in_elems, out, n_out, Hspec, K, tw, st, diag);
    case 14: return launch_ols<14>(in, in_elems, out, n_out, Hspec, K, tw, st, diag);
    }
    set_error("fir ols: no plan for log2(N) = %d", log2n);
    return PCX_ERR_UNSUPPORTED;
}

}  // namespace pcx
