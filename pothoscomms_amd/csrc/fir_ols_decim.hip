// fir_ols_decim.hip -- decimating /comms/fir_filter (complex_float32, interpolation 1, decimation M in {2, 4, 8, 16})
// in the frequency domain with the decimation folded into the spectrum.
//
// The reference computes  y[t] = sum_k h[k] x[M-1 + t M - k]  (FIRFilter.cpp:286-302: the decimator fires when
// (f+1) % M == 0) -- every M-th sample of the full-rate filter output.  fir_cf32_ols4096_poly_kernel evaluates that
// output at full rate (two 4096-point transforms per block) and stores one sample in M.  Decimating a sequence by M
// aliases its spectrum:  Yd[k'] = sum_m Y[k' + m N/M], k' < N/M,  so the inverse transform only has to be N/M points
// long: one 4096-point forward transform, the H multiply, a fold of M bins into one and a 4096/M-point inverse --
// (1 + 1/M) transforms' worth of arithmetic instead of 2, on a kernel that is bound by its arithmetic.
//
// Lane j holds X[j + 256 r], r = 0..15, after the forward passes, and N/M = 256 P with P = 16/M: the M bins that fold
// onto k' = j + 256 r' are the lane's own registers r = r' + P m.  The N/M-point inverse (as conj . FFT . conj) is a
// radix-P decimation-in-frequency stage over the lane's P folded values, the twiddle W_{N/M}^(j k1), and P independent
// 256-point transforms (one per k1) run by 16 lanes each on the radix-16 passes of fft4096.hpp; output sample
// n' = k1 + P k2 goes through LDS once more so that the stores are contiguous.
// The decimator's phase (outputs at full-rate indices == M-1 mod M) is a circular advance by M-1 samples, folded into
// H on the host (pcx_api.hip) together with the 1/N of the inverse.  Block geometry as in fir_ols.hip: overlap Kov =
// K-1 rounded up to 16 samples (a multiple of every M here), S = 4096 - Kov full-rate outputs = S/M stored per block.
#include "fft4096.hpp"
#include "pcx_sched.hpp"
#include <cstdlib>

#include "pcx_internal.hpp"

namespace pcx {

namespace {
using namespace fft4k;

// NOINV (diagnostic library, TIMING ONLY, wrong outputs): the block without its N/M-point inverse stage -- what batching that stage
// over several blocks could save at most
template <int LOG2M, bool DYN = false, bool NOINV = false, int OCC = 4>
__global__ __launch_bounds__(256, OCC) void fir_cf32_ols4096_decim_kernel(const float2 *__restrict__ in, size_t in_elems, float2 *__restrict__ out,
                                                                        size_t n_out, const float2 *__restrict__ Hspec, int Kov, int pad,
                                                                        const float2 *__restrict__ twtab, size_t first_full, size_t nfull,
                                                                        size_t nblocks, unsigned M2, unsigned magic2, size_t n_dec2,
                                                                        pcx::SchedState *__restrict__ sched)
{
    // M2 > 1: the decimation factor is M * M2 (M2 odd or any cofactor): the folded stream is decimated once more on the
    // store -- sample g of it is kept when (g + 1) % M2 == 0 and lands at (g + 1) / M2 - 1 (n_dec2 of them in all)
    constexpr int M = 1 << LOG2M, P = 16 / M;       // P folded values per lane = frames of the 256-point stage
    constexpr int FRAME = 272;                       // 256 + 256/16: padded sub-frame
    constexpr int OIMG = 8 * FRAME;                  // output image behind the (at most 8) sub-frames
    __shared__ cf lds[LDS_ELEMS];
    const int j = threadIdx.x;
    const size_t S = (size_t)(N - Kov), Sd = S >> LOG2M;
    __shared__ unsigned sched_slot;
    pcx::BlockWalk<DYN> walk;     // DYN: blocks dealt dynamically (pcx_sched.hpp)
    if (!walk.begin(sched, &sched_slot, nblocks, j)) { walk.finish(j); return; }
    // pass-3 lane constants (30 VGPRs): in registers across the block loop for M >= 8; for M = 2 / 4 the inverse stage needs
    // the room and they are re-read from L2 in every block, like H (measured: M = 2 217 -> 249, M = 4 275 -> 288 Gsamples/s;
    // M = 16 loses 9 % with the reload and keeps them)
    constexpr bool TW3_REG = M >= 8;
    LaneTw tw3r;
    if (TW3_REG) load_pass3_twiddles(tw3r, twtab, j);
    stage_pass2_twiddles(lds, twtab, j);
    // the lane's 16 bins of H are re-read from L2 at the multiply (not held in 32 VGPRs): the 256-point stage keeps a
    // second 16-point set alive across two barriers and the kernel has to stay within 128 VGPRs for 4 workgroups per CU
    const cf *Hg = reinterpret_cast<const cf *>(Hspec) + j;
    // W_{N/M}^j: the lane constant of the decimation-in-frequency stage (its powers k1 = 2 .. P-1 are rebuilt by
    // multiplication in every block: 6 packed multiplies against 12 more registers)
    cf td1;
    {
        float sn, cs;
        sincospif(-2.0f * (float)spec_lane(j) / (float)(256 * P), &sn, &cs);    // the lane's bins are js + 256 r (file header)
        td1 = cf{cs, sn};
    }
    const int js = spec_lane(j);
    const int fi = j >> 4, l = j & 15;               // sub-frame and lane inside it (lanes j < 16 P run the 256-point stage)
    const bool sub = j < 16 * P;

    for (;;) {
        const size_t b = walk.block();
        cf v[16];
        if (b >= first_full && b < nfull) {
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(in + b * S - pad, N * 8);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const u32x2 t = (r == 0 || r == 15) ? __builtin_amdgcn_raw_buffer_load_b64(rs, j * 8, 2048 * r, 0)
                                                    : __builtin_amdgcn_raw_buffer_load_b64(rs, j * 8, 2048 * r, 2);
                v[r] = cf{__uint_as_float(t.x), __uint_as_float(t.y)};
            }
        } else {
            const size_t shift = b * S >= (size_t)pad ? 0 : (size_t)pad - b * S;
            const size_t first = b * S + shift - pad;
            const size_t left = in_elems > first ? in_elems - first : 0;
            const size_t want = (size_t)N - shift;
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(in + first, (unsigned)((left < want ? left : want) * 8));
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, (j + 256 * r - (int)shift) * 8, 0, 0);
                v[r] = cf{__uint_as_float(t.x), __uint_as_float(t.y)};
            }
        }
        walk.draw(j);
        if (TW3_REG) {
            dif_a_math(v, tw3r);
        } else {
            const float2 *tp = twtab;
            asm volatile("" : "+v"(tp));
            LaneTw tw3;
            load_pass3_twiddles(tw3, tp, j);
            dif_a_math(v, tw3);
        }
        walk.publish(j);      // the first exchange opens with a barrier, and more follow before the block ends
        dif_rest(v, lds, j);
        // u[r] = conj(X[j + 256 r] * H'[j + 256 r]), natural r
        const cf *Hb = Hg;
        asm volatile("" : "+v"(Hb));   // loop-invariant: without this the loads are hoisted back into registers
        cf u[16];
#pragma unroll
        for (int q = 0; q < 16; q += 2) {
            const int k0 = bin_of(q), k1 = bin_of(q + 1);
            u[k0] = v[q];
            u[k1] = v[q + 1];
            cmul2_conj(u[k0], u[k1], Hb[256 * k0], Hb[256 * k1]);
        }
        // fold: the M bins that alias onto k' = j + 256 r'
        cf z[P];
#pragma unroll
        for (int r = 0; r < P; r++) {
            z[r] = u[r];
#pragma unroll
            for (int m = 1; m < M; m++) z[r] = z[r] + u[r + P * m];
        }
        // radix-P decimation-in-frequency stage over r', then the lane twiddle W^(j k1)
        if constexpr (P == 8) fft8(z[0], z[1], z[2], z[3], z[4], z[5], z[6], z[7]);
        else if constexpr (P == 4) fft4(z[0], z[1], z[2], z[3]);
        else if constexpr (P == 2) { const cf a = z[0], c = z[1]; z[0] = a + c; z[1] = a - c; }
        {
            cf t = td1;
#pragma unroll
            for (int k1 = 1; k1 < P; k1++) {
                z[k1] = cmul1(z[k1], t);
                if (k1 + 1 < P) t = cmul1(t, td1);
            }
        }
        __syncthreads();                                  // every lane is done with the forward image (pass 3 reads)
        if (NOINV) {
#pragma unroll
            for (int k1 = 0; k1 < P; k1++) { const int n = j + 256 * k1; lds[OIMG + n + (n >> 4)] = z[k1]; }
        }
#pragma unroll
        for (int k1 = 0; k1 < P; k1++) lds[k1 * FRAME + js + (js >> 4)] = z[k1];
        __syncthreads();
        // P independent 256-point transforms, 16 lanes each: radix 16 x 16 (Ns = 1, Ns = 16)
        cf w[16];
        cf *fr = lds + fi * FRAME;
        if (!NOINV) {
        if (sub) {
#pragma unroll
            for (int s = 0; s < 16; s++) w[s] = fr[l + 17 * s];
            fft16_plain(w);
        }
        wave_lds_order();                                 // a frame belongs to sixteen lanes of ONE wave from here to the stage's last read:
        if (sub) {                                        // no workgroup barrier between its passes (fft4096.hpp)
#pragma unroll
            for (int q = 0; q < 16; q++) fr[17 * l + bin_of(q)] = w[q];
        }
        wave_lds_order();
        if (sub) {
#pragma unroll
            for (int s = 0; s < 16; s++) w[s] = fr[l + 17 * s];
            LaneTw tl;
            const cf *t2 = lds + LDS_DATA + l;
#pragma unroll
            for (int p = 0; p < 3; p++) tl.a[p] = t2[p * 16];
#pragma unroll
            for (int p = 0; p < 12; p++) tl.c[p] = t2[(3 + p) * 16];
            fft16_tw_conj(w, tl);      // conjugated on its last additions: w = the decimated time samples (fft4096.hpp fft4_conj)
            // w[q] = bin k2 = l + 16 bin_of(q) of sub-transform k1 = fi: decimated time sample n' = k1 + P k2
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int n = fi + P * (l + 16 * bin_of(q));
                lds[OIMG + n + (n >> 4)] = w[q];
            }
        }
        __syncthreads();
        }
        // decimated sample n' of the block is output b*Sd + n' - Kov/M; n' < Kov/M wraps past num_records and is dropped
        if (M2 > 1) {
            const size_t B0 = (b * Sd) / M2;
            const unsigned base = (unsigned)(b * Sd - B0 * M2);
            const size_t room = n_dec2 > B0 ? n_dec2 - B0 : 0;
            const __amdgpu_buffer_rsrc_t ws = make_rsrc(out + B0, (unsigned)((room < (size_t)2050 ? room : (size_t)2050) * 8));
            const size_t left = n_out - b * Sd;               // samples of the folded stream this block may produce
#pragma unroll
            for (int i = 0; i < P; i++) {
                const int n = j + 256 * i, g = n - (Kov >> LOG2M);
                const unsigned t = base + (unsigned)g + 1u;
                const unsigned qt = __umulhi(t, magic2);
                if (g >= 0 && (size_t)g < left && qt * M2 == t) {
                    const cf y = lds[OIMG + n + (n >> 4)];
                    store_cf<2>(ws, (qt - 1u) * 8u, y);
                }
            }
        } else {
        const size_t room = n_out - b * Sd;
        const __amdgpu_buffer_rsrc_t ws = make_rsrc(out + b * Sd, (unsigned)((room < Sd ? room : Sd) * 8));
        const unsigned vbase = (unsigned)(j - (Kov >> LOG2M)) * 8u;
#pragma unroll
        for (int i = 0; i < P; i++) {
            const int n = j + 256 * i;
            const cf y = lds[OIMG + n + (n >> 4)];
            store_cf<2>(ws, vbase + (unsigned)(256 * i) * 8u, y);
        }
        }
        if (!walk.advance()) break;
    }
    walk.finish(j);
}

// --------------------------------------------------------------------------------- //
// The same filter with the N/M-point inverse stage BATCHED over G consecutive blocks.
//
// In the kernel above the inverse stage of a block keeps only 16 P of the 256 lanes busy (P = 16/M sub-transforms of 256
// points, 16 lanes each) through six barriers: a quarter of the block's time at M = 8 (0.192 -> 0.144 ms per 64 Mi samples
// without it, timing-only build PCX_DECIM_NOINV).  Here a workgroup takes G blocks at a time: forward passes, H and the fold
// for each of them, the P folded and twiddled values per lane parked in registers (G P <= 16 of them), then ONE inverse stage
// over all G P sub-transforms -- 16 G P lanes busy, the same six barriers once per group -- and the G blocks' outputs, which
// are contiguous in the output stream, stored together.  Plain decimation factors only (no cofactor M2: those calls keep the
// kernel above).  The parked values cost registers the forward passes need: the product runs it at THREE workgroups per CU
// (OCC = 3, 170 VGPRs; launch_decim below has the measurements), with H held in registers as well (HREG) where that fits.
// --------------------------------------------------------------------------------- //
// The pass-3 factors of a lane DERIVED instead of held or re-read (TWM = 2): four of the fifteen -- w, w^2, w^3 (= c[n2][0]) and w^4
// (= a[0]), eight registers instead of thirty -- stay in registers; per block a[1] = w^8, a[2] = w^12 take one product each and the nine
// c[n2][k1] = w^n2 W16^(n2 k1), k1 = 1..3, one product with a constant (W16^4 = -i: none): every factor is one multiplication away
// from a table value, ~1e-7 relative.  What it buys is the register budget of a FOURTH workgroup per CU without the 60 KB per block
// of table re-reads the TWM = 0 builds pay (the exchanges of these kernels are exposed at three per CU).
struct LaneTwSeed {
    cf w1, w2, w3, w4;
};
__device__ __forceinline__ void load_tw3_seed(LaneTwSeed &s, const float2 *__restrict__ tab, int j)
{
    const cf *tb = reinterpret_cast<const cf *>(tab) + LDS_TW2;
    s.w4 = tb[j];
    s.w1 = tb[(3 + 0) * 256 + j];
    s.w2 = tb[(3 + 4) * 256 + j];
    s.w3 = tb[(3 + 8) * 256 + j];
}
// a * w, w a compile-time constant riding in a scalar register pair
__device__ __forceinline__ cf cmul1k(cf a, cf w)
{
    cf t, r;
    asm("v_pk_mul_f32 %0, %2, %3 op_sel:[1,1] op_sel_hi:[0,1] neg_lo:[0,1]\n\t"
        "v_pk_fma_f32 %1, %2, %3, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]"
        : "=&v"(t), "=&v"(r)
        : "v"(a), "s"(w));
    return r;
}
__device__ __forceinline__ void expand_tw3(LaneTw &t, const LaneTwSeed &s)
{
    // W16^m = exp(-j 2 pi m / 16), m = n2 k1
    constexpr float C1 = 0.92387953251128673848f, S1 = 0.38268343236508977173f, R2 = 0.70710678118654752440f;
    const cf W1 = {C1, -S1}, W2 = {R2, -R2}, W3 = {S1, -C1}, W6 = {-R2, -R2}, W9 = {-C1, S1};
    t.a[0] = s.w4;
    t.a[1] = cmul1(s.w4, s.w4);
    t.a[2] = cmul1(t.a[1], s.w4);
    t.c[0] = s.w1; t.c[1] = cmul1k(s.w1, W1); t.c[2] = cmul1k(s.w1, W2); t.c[3] = cmul1k(s.w1, W3);
    t.c[4] = s.w2; t.c[5] = cmul1k(s.w2, W2); t.c[6] = cf{s.w2.y, -s.w2.x}; t.c[7] = cmul1k(s.w2, W6);     // W16^4 = -i
    t.c[8] = s.w3; t.c[9] = cmul1k(s.w3, W3); t.c[10] = cmul1k(s.w3, W6); t.c[11] = cmul1k(s.w3, W9);
}

template <int LOG2M, int LOG2G, int TWM, int OCC = 4, bool HREG = false>   // TWM: the pass-3 factors 0 re-read per block, 1 held, 2 derived from four (expand_tw3)
__global__ __launch_bounds__(256, OCC) void fir_cf32_ols4096_decim_batched_kernel(const float2 *__restrict__ in, size_t in_elems, float2 *__restrict__ out,
                                                                                size_t n_out, const float2 *__restrict__ Hspec, int Kov, int pad,
                                                                                const float2 *__restrict__ twtab, size_t first_full, size_t nfull,
                                                                                size_t nblocks)
{
    constexpr int M = 1 << LOG2M, P = 16 / M, G = 1 << LOG2G, T = G * P;
    static_assert(T <= 16, "at most 16 sub-transforms of 256 points fill the workgroup");
    constexpr int FRAME = 272;
    __shared__ cf lds[LDS_ELEMS];
    const int j = threadIdx.x;
    const size_t S = (size_t)(N - Kov), Sd = S >> LOG2M;
    const size_t ngroups = (nblocks + G - 1) / G;
    const int js = spec_lane(j);       // the lane's bins are js + 256 r (file header)
    LaneTw tw3r;
    LaneTwSeed tw3s;
    if (TWM == 1) load_pass3_twiddles(tw3r, twtab, j);
    if (TWM == 2) load_tw3_seed(tw3s, twtab, j);
    stage_pass2_twiddles(lds, twtab, j);
    const cf *Hg = reinterpret_cast<const cf *>(Hspec) + j;      // (turned on the host: lane j finds H[js + 256 r] at j + 256 r)
    cf Hr[16];
    if (HREG) {
#pragma unroll
        for (int k = 0; k < 16; k++) Hr[k] = Hg[256 * k];
    }
    cf td1;
    {
        float sn, cs;
        sincospif(-2.0f * (float)js / (float)(256 * P), &sn, &cs);
        td1 = cf{cs, sn};
    }
    const int fi = j >> 4, l = j & 15;               // sub-transform t = fi (block fi / P of the group, k1 = fi % P) and lane inside it
    const bool sub = j < 16 * T;

    for (size_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        cf zz[T];
#pragma unroll
        for (int g = 0; g < G; g++) {
            const size_t b = grp * G + g;
            if (b >= nblocks) {                       // (workgroup-uniform) the launch's last group may be short
#pragma unroll
                for (int k1 = 0; k1 < P; k1++) zz[g * P + k1] = cf{0.f, 0.f};
                continue;
            }
            cf v[16];
            if (b >= first_full && b < nfull) {
                const __amdgpu_buffer_rsrc_t rs = make_rsrc(in + b * S - pad, N * 8);
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const u32x2 t = (r == 0 || r == 15) ? __builtin_amdgcn_raw_buffer_load_b64(rs, j * 8, 2048 * r, 0)
                                                        : __builtin_amdgcn_raw_buffer_load_b64(rs, j * 8, 2048 * r, 2);
                    v[r] = cf{__uint_as_float(t.x), __uint_as_float(t.y)};
                }
            } else {
                const size_t shift = b * S >= (size_t)pad ? 0 : (size_t)pad - b * S;
                const size_t first = b * S + shift - pad;
                const size_t left = in_elems > first ? in_elems - first : 0;
                const size_t want = (size_t)N - shift;
                const __amdgpu_buffer_rsrc_t rs = make_rsrc(in + first, (unsigned)((left < want ? left : want) * 8));
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, (j + 256 * r - (int)shift) * 8, 0, 0);
                    v[r] = cf{__uint_as_float(t.x), __uint_as_float(t.y)};
                }
            }
            if (TWM == 1) {
                dif_a_math(v, tw3r);
            } else if (TWM == 2) {
                LaneTw tw3;
                expand_tw3(tw3, tw3s);
                dif_a_math(v, tw3);
            } else {
                const float2 *tp = twtab;
                asm volatile("" : "+v"(tp));
                LaneTw tw3;
                load_pass3_twiddles(tw3, tp, j);
                dif_a_math(v, tw3);
            }
            dif_rest(v, lds, j);
            const cf *Hb = Hg;
            asm volatile("" : "+v"(Hb));
            cf u[16];
#pragma unroll
            for (int q = 0; q < 16; q += 2) {
                const int k0 = bin_of(q), k1 = bin_of(q + 1);
                u[k0] = v[q];
                u[k1] = v[q + 1];
                if (HREG) cmul2_conj(u[k0], u[k1], Hr[k0], Hr[k1]);
                else cmul2_conj(u[k0], u[k1], Hb[256 * k0], Hb[256 * k1]);
            }
            cf z[P];
#pragma unroll
            for (int r = 0; r < P; r++) {
                z[r] = u[r];
#pragma unroll
                for (int m = 1; m < M; m++) z[r] = z[r] + u[r + P * m];
            }
            if constexpr (P == 8) fft8(z[0], z[1], z[2], z[3], z[4], z[5], z[6], z[7]);
            else if constexpr (P == 4) fft4(z[0], z[1], z[2], z[3]);
            else if constexpr (P == 2) { const cf a = z[0], c = z[1]; z[0] = a + c; z[1] = a - c; }
            {
                cf t = td1;
#pragma unroll
                for (int k1 = 1; k1 < P; k1++) {
                    z[k1] = cmul1(z[k1], t);
                    if (k1 + 1 < P) t = cmul1(t, td1);
                }
            }
#pragma unroll
            for (int k1 = 0; k1 < P; k1++) zz[g * P + k1] = z[k1];
        }
        // ---- one inverse stage for the whole group: T sub-transforms of 256 points, 16 lanes each ----
        __builtin_amdgcn_s_setprio(1);                    // (the second half of a group ahead of another workgroup's first: fir_ols_f64.hip, round 6)
        __syncthreads();                                  // the last block's pass-3 reads of the image are done
#pragma unroll
        for (int t = 0; t < T; t++) lds[t * FRAME + js + (js >> 4)] = zz[t];
        __syncthreads();
        cf w[16];
        cf *fr = lds + fi * FRAME;
        if (sub) {
#pragma unroll
            for (int s = 0; s < 16; s++) w[s] = fr[l + 17 * s];
            fft16_plain(w);
        }
        wave_lds_order();                                 // a frame belongs to sixteen lanes of ONE wave from here to the stage's last read:
        if (sub) {                                        // no workgroup barrier between its passes (fft4096.hpp)
#pragma unroll
            for (int q = 0; q < 16; q++) fr[17 * l + bin_of(q)] = w[q];
        }
        wave_lds_order();
        if (sub) {
#pragma unroll
            for (int s = 0; s < 16; s++) w[s] = fr[l + 17 * s];
            LaneTw tl;
            const cf *t2 = lds + LDS_DATA + l;
#pragma unroll
            for (int p = 0; p < 3; p++) tl.a[p] = t2[p * 16];
#pragma unroll
            for (int p = 0; p < 12; p++) tl.c[p] = t2[(3 + p) * 16];
            fft16_tw_conj(w, tl);      // conjugated on its last additions (fft4_conj)
        }
        __syncthreads();                                  // every frame has been read: the image is rewritten in output order
        if (sub) {
            // w[q] = bin k2 = l + 16 bin_of(q) of sub-transform (g, k1) = (fi / P, fi % P): decimated sample n' = k1 + P k2 of block g
            const int g = fi / P, k1 = fi % P;
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int n = g * (256 * P) + k1 + P * (l + 16 * bin_of(q));
                lds[n + (n >> 4)] = w[q];
            }
        }
        __syncthreads();
        // decimated sample n' of block b is output b*Sd + n' - Kov/M; n' < Kov/M wraps past num_records and is dropped
#pragma unroll
        for (int g = 0; g < G; g++) {
            const size_t b = grp * G + g;
            if (b >= nblocks) break;
            const size_t room = n_out - b * Sd;
            const __amdgpu_buffer_rsrc_t ws = make_rsrc(out + b * Sd, (unsigned)((room < Sd ? room : Sd) * 8));
            const unsigned vbase = (unsigned)(j - (Kov >> LOG2M)) * 8u;
#pragma unroll
            for (int i = 0; i < P; i++) {
                const int n = g * (256 * P) + j + 256 * i;
                const cf y = lds[n + (n >> 4)];
                store_cf<2>(ws, vbase + (unsigned)(256 * i) * 8u, y);
            }
        }
        __builtin_amdgcn_s_setprio(0);
    }
}

template <int LOG2M>
int launch_decim(const void *in, size_t in_elems, void *out, size_t n_iter, const void *Hspec, size_t K, const void *tw4096, size_t M2,
                 void *sched, hipStream_t st)
{
    constexpr size_t M = (size_t)1 << LOG2M;
    const size_t Km1 = K - 1;
    const size_t Kov = (Km1 + 15) / 16 * 16, pad = Kov - Km1;
    const size_t S = 4096 - Kov, Sd = S / M;
    const size_t n_out = n_iter / M;
    const size_t nblocks = (n_out + Sd - 1) / Sd;
    const size_t first_full = pad > 0 ? 1 : 0;
    size_t nfull = n_iter / S;
    while (nfull > first_full && (nfull - 1) * S - pad + 4096 > in_elems) nfull--;
    if (nfull < first_full) nfull = first_full;
    // dynamic dealing measured SLOWER here (tools/ab_sched.sh, M = 8: 0.2190 vs 0.1950 ms): the kernel is bound by its arithmetic and
    // re-reads H per block, and the grid stride keeps neighbouring blocks on neighbouring workgroups.  The product keeps the
    // stride; PCX_SCHED_RESAMPLERS (diagnostic library) selects the dealer for A/B.
    const bool dyn = sched && nblocks > 2 * 1024 && PCX_ENV_SET("PCX_SCHED_RESAMPLERS");
    const unsigned grid = dyn ? 1024u : persistent_grid(nblocks, 1024);
    const unsigned magic2 = M2 > 1 ? (unsigned)(((1ull << 32) + M2 - 1) / M2) : 0u;
    // Measured (tools/ab_decim.sh, tools/ab_decim_occ.sh; profiles/r02/ab_decim.txt, ab_decim_occ.txt).  At four workgroups per CU (128
    // VGPRs) the parked values push the forward passes into scratch and batching pays at M = 8 with two blocks only (+6-8 %).  At
    // THREE workgroups per CU (170 VGPRs) it pays everywhere -- the registers are worth more than the fourth workgroup:
    //   M = 2: 236 -> 274-276 Gsamples/s of input (two blocks per group);   M = 4: 302 -> 353 (two blocks);
    //   M = 8: 359 -> 403 (four blocks, H held in registers);                M = 16: 391 -> 445 (four blocks, H in registers).
    // (Also round 3: the NEXT block's samples prefetched into registers while this block is transformed -- 32 VGPRs, so H has to be
    // re-read per block instead of held -- 0.1686 ms at M = 8 against 0.1697-0.1709 for the same build without the prefetch and 0.1458
    // with H in registers: the load latency is not what this kernel waits for.  Not kept.)
    // Round 3, forward transform on the sixteen-lane exchange, every configuration re-timed in a process of its own behind 300 settling
    // launches (tools/resampler_sweep.py, profiles/r03/resampler_sweep.txt; ms per 64 Mi input samples, two runs each):
    //   M = 2: two blocks, H re-read 0.2021-0.2030 (H in registers 0.2274-0.2281)      M = 4: two blocks, H in registers 0.1650 (re-read 0.1726-0.1731)
    //   M = 8: two blocks, H in registers 0.1456-0.1460 (four blocks 0.1479-0.1500)    M = 16: four blocks, H in registers 0.1357-0.1375
    // -- the defaults below.
    // That is the product path for plain factors (M2 == 1); PCX_DECIM_UNBATCHED (diagnostic library) keeps the one-block kernel,
    // PCX_DECIM_G / PCX_DECIM_HREG / PCX_DECIM_OCC=4 the other configurations, for A/B.
    if (M2 == 1 && !PCX_ENV_SET("PCX_DECIM_UNBATCHED")) {
        constexpr int LGMAX = LOG2M == 1 ? 1 : LOG2M == 2 ? 2 : 3;      // G P <= 16, eight blocks at most
        int lg = (int)PCX_ENV_INT("PCX_DECIM_G", LOG2M >= 4 ? 2 : 1);
        if (lg > LGMAX) lg = LGMAX;
        if (lg < 1) lg = 1;
        const bool hreg = PCX_ENV_INT("PCX_DECIM_HREG", LOG2M >= 2 ? 1 : 0) != 0;
        const bool derived = PCX_ENV_INT("PCX_DECIM_TW3", 1) == 2;     // (diagnostic library: the product library's PCX_ENV_INT is the default) factors derived per block, four workgroups per CU
        const bool occ4 = derived || PCX_ENV_INT("PCX_DECIM_OCC", 3) == 4;
        const bool tw3 = PCX_ENV_INT("PCX_DECIM_TW3", LOG2M >= 3 ? 1 : 0) != 0;     // (four-per-CU builds only: pass-3 constants in registers)
        const size_t ngroups = (nblocks + ((size_t)1 << lg) - 1) >> lg;
        const unsigned bgrid = persistent_grid(ngroups, occ4 ? 1024 : 768);
#define PCX_DECIM_ARGS dim3(bgrid), dim3(256), 0, st, (const float2 *)in, in_elems, (float2 *)out, n_out, (const float2 *)Hspec, (int)Kov, (int)pad, \
                       (const float2 *)tw4096, first_full, nfull, nblocks
#ifdef PCX_DIAG
        // (diagnostic library) PCX_DECIM_TW3=2: the pass-3 factors derived per block from four held ones, four workgroups per CU, H held
        // (PCX_DECIM_HREG=1) or re-read -- the A/B of DESIGN.md 4.6's "a fourth workgroup would hide the exchanges"
#define PCX_DECIM_DERIVED(LG)                                                                                                          \
            if (derived && hreg) hipLaunchKernelGGL((fir_cf32_ols4096_decim_batched_kernel<LOG2M, LG, 2, 4, true>), PCX_DECIM_ARGS);    \
            else if (derived) hipLaunchKernelGGL((fir_cf32_ols4096_decim_batched_kernel<LOG2M, LG, 2, 4, false>), PCX_DECIM_ARGS);      \
            else
#else
#define PCX_DECIM_DERIVED(LG)
#endif
#define PCX_DECIM_LAUNCH(LG)                                                                                                      \
        do {                                                                                                                      \
            PCX_DECIM_DERIVED(LG)                                                                                                 \
            if (occ4 && tw3) hipLaunchKernelGGL((fir_cf32_ols4096_decim_batched_kernel<LOG2M, LG, 1, 4, false>), PCX_DECIM_ARGS);     \
            else if (occ4) hipLaunchKernelGGL((fir_cf32_ols4096_decim_batched_kernel<LOG2M, LG, 0, 4, false>), PCX_DECIM_ARGS);      \
            else if (hreg) hipLaunchKernelGGL((fir_cf32_ols4096_decim_batched_kernel<LOG2M, LG, 1, 3, true>), PCX_DECIM_ARGS);        \
            else hipLaunchKernelGGL((fir_cf32_ols4096_decim_batched_kernel<LOG2M, LG, 1, 3, false>), PCX_DECIM_ARGS);                 \
        } while (0)
        if (lg == 1) PCX_DECIM_LAUNCH(1);
        else if (lg == 2) PCX_DECIM_LAUNCH((LGMAX >= 2 ? 2 : 1));
        else PCX_DECIM_LAUNCH((LGMAX >= 3 ? 3 : 1));
#undef PCX_DECIM_LAUNCH
#undef PCX_DECIM_DERIVED
#undef PCX_DECIM_ARGS
        PCX_LAUNCH_CHECK();
        return PCX_OK;
    }
#ifdef PCX_DIAG
    if (PCX_ENV_SET("PCX_DECIM_NOINV")) {
        hipLaunchKernelGGL((fir_cf32_ols4096_decim_kernel<LOG2M, false, true>), dim3(grid), dim3(256), 0, st, (const float2 *)in, in_elems, (float2 *)out, n_out,
                           (const float2 *)Hspec, (int)Kov, (int)pad, (const float2 *)tw4096, first_full, nfull, nblocks, (unsigned)M2, magic2,
                           n_out / M2, (pcx::SchedState *)nullptr);
        PCX_LAUNCH_CHECK();
        return PCX_OK;
    }
#endif
    // (the factors with a cofactor M2 stay on this kernel) three workgroups per CU as above: PCX_DECIM_OCC=4 (diagnostic library) for A/B
    if (!dyn && PCX_ENV_INT("PCX_DECIM_OCC", 3) == 3) {
        hipLaunchKernelGGL((fir_cf32_ols4096_decim_kernel<LOG2M, false, false, 3>), dim3(persistent_grid(nblocks, 768)), dim3(256), 0, st, (const float2 *)in,
                           in_elems, (float2 *)out, n_out, (const float2 *)Hspec, (int)Kov, (int)pad, (const float2 *)tw4096, first_full, nfull, nblocks,
                           (unsigned)M2, magic2, n_out / M2, (pcx::SchedState *)nullptr);
        PCX_LAUNCH_CHECK();
        return PCX_OK;
    }
    if (dyn)
        hipLaunchKernelGGL((fir_cf32_ols4096_decim_kernel<LOG2M, true>), dim3(grid), dim3(256), 0, st, (const float2 *)in, in_elems, (float2 *)out, n_out,
                           (const float2 *)Hspec, (int)Kov, (int)pad, (const float2 *)tw4096, first_full, nfull, nblocks, (unsigned)M2, magic2,
                           n_out / M2, (pcx::SchedState *)sched);
    else
        hipLaunchKernelGGL((fir_cf32_ols4096_decim_kernel<LOG2M, false>), dim3(grid), dim3(256), 0, st, (const float2 *)in, in_elems, (float2 *)out, n_out,
                           (const float2 *)Hspec, (int)Kov, (int)pad, (const float2 *)tw4096, first_full, nfull, nblocks, (unsigned)M2, magic2,
                           n_out / M2, (pcx::SchedState *)nullptr);
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}

// --------------------------------------------------------------------------------- //
// The mirror image for INTERPOLATING filters (decimation 1, interpolation L in {2, 4, 8, 16}): the reference's polyphase
// rows (FIRFilter.cpp:286-302, taps_j[k] = taps[j + k L]) are the convolution of the zero-stuffed stream with the whole
// tap vector, and zero-stuffing by L REPLICATES the spectrum:  Xz[k] = Xs[k mod N/L],  Xs = FFT_{N/L}(input block).
// So a block is a 4096/L-point forward transform (16/L independent 256-point transforms of the input's polyphase
// components on 16 lanes each, then one twiddle and a radix-(16/L) stage in the lane, which leaves Xs[j + 256 r'] in lane
// j), 16 multiplies by H (all 4096 bins of the full tap vector), and the ordinary 4096-point inverse, whose output is
// the interleaved output stream itself: contiguous 2 KiB rows instead of the polyphase kernel's stride-L stores.
// Geometry at the input rate: Kov_in = K-1 rounded up so that Kov_in*L is a multiple of 16, S_in = 4096/L - Kov_in
// input samples = S_in*L outputs per block; output i of a block is valid from i >= Kov_in*L on (the wrapped positions a
// valid output still touches are zero-stuffing zeros).
// --------------------------------------------------------------------------------- //
// OCC = workgroups per CU the register budget is cut for (4: 128 VGPRs, 3: 170); HREG: the lane's 16 bins of H held in registers
// instead of re-read from L2 in every block (needs the 170)
template <int LOG2L, bool DYN = false, int OCC = 4, bool HREG = false>
__global__ __launch_bounds__(256, OCC) void fir_cf32_ols4096_interp_kernel(const float2 *__restrict__ in, size_t in_elems, float2 *__restrict__ out,
                                                                         size_t n_out, const float2 *__restrict__ Hspec, int Kov_in, int pad_in,
                                                                         const float2 *__restrict__ twtab, size_t nblocks,
                                                                         pcx::SchedState *__restrict__ sched)
{
    constexpr int L = 1 << LOG2L, P = 16 / L, LOG2P = 4 - LOG2L, ND = 256 * P;
    constexpr int FRAME = 272;
    __shared__ cf lds[LDS_ELEMS];
    const int j = threadIdx.x;
    const size_t S_in = (size_t)(ND - Kov_in), S_out = S_in << LOG2L;
    const int Kov_out = Kov_in << LOG2L;
    __shared__ unsigned sched_slot;
    pcx::BlockWalk<DYN> walk;
    if (!walk.begin(sched, &sched_slot, nblocks, j)) { walk.finish(j); return; }
    LaneTw tw3;
    load_pass3_twiddles(tw3, twtab, j);
    stage_pass2_twiddles(lds, twtab, j);
    const cf *Hg = reinterpret_cast<const cf *>(Hspec) + j;
    cf Hr[16];
    if (HREG) {
#pragma unroll
        for (int k = 0; k < 16; k++) Hr[k] = Hg[256 * k];
    }
    cf td1;
    {
        float sn, cs;
        sincospif(-2.0f * (float)spec_lane(j) / (float)ND, &sn, &cs);     // the lane takes the bins js + 256 r (decimator above)
        td1 = cf{cs, sn};
    }
    const int js = spec_lane(j);
    const int fi = j >> 4, l = j & 15;
    const bool sub = j < 16 * P;

    for (;;) {
        const size_t b = walk.block();
        // input window: ND samples from input index b*S_in - pad_in (those before the buffer read 0: they only feed dropped outputs)
        const size_t start = b * S_in;
        const size_t shift = start >= (size_t)pad_in ? 0 : (size_t)pad_in - start;
        const size_t first = start + shift - pad_in;
        const size_t left = in_elems > first ? in_elems - first : 0;
        const size_t want = (size_t)ND - shift;
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(in + first, (unsigned)((left < want ? left : want) * 8));
        cf x[P];
#pragma unroll
        for (int i = 0; i < P; i++) {
            const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, (j + 256 * i - (int)shift) * 8, 0, 0);
            x[i] = cf{__uint_as_float(t.x), __uint_as_float(t.y)};
        }
        walk.draw(j);
        __syncthreads();                                  // the previous block's inverse is done with the image
        // polyphase component n1 = n mod P of the window goes to sub-frame n1 at position n / P  (n = j + 256 i)
#pragma unroll
        for (int i = 0; i < P; i++) {
            const int n2 = (j >> LOG2P) + (256 >> LOG2P) * i;
            lds[(j & (P - 1)) * FRAME + n2 + (n2 >> 4)] = x[i];
        }
        __syncthreads();
        cf w[16];
        cf *fr = lds + fi * FRAME;
        if (sub) {
#pragma unroll
            for (int s = 0; s < 16; s++) w[s] = fr[l + 17 * s];
            fft16_plain(w);
        }
        wave_lds_order();                                 // a frame belongs to sixteen lanes of ONE wave from here to the stage's last read:
        if (sub) {                                        // no workgroup barrier between its passes (fft4096.hpp)
#pragma unroll
            for (int q = 0; q < 16; q++) fr[17 * l + bin_of(q)] = w[q];
        }
        wave_lds_order();
        if (sub) {
#pragma unroll
            for (int s = 0; s < 16; s++) w[s] = fr[l + 17 * s];
            LaneTw tl;
            const cf *t2 = lds + LDS_DATA + l;
#pragma unroll
            for (int p = 0; p < 3; p++) tl.a[p] = t2[p * 16];
#pragma unroll
            for (int p = 0; p < 12; p++) tl.c[p] = t2[(3 + p) * 16];
            fft16_tw(w, tl);
        }
        wave_lds_order();                                 // every lane of the frame has read its inputs (the frame's own sixteen lanes)
        if (sub) {
            // G_n1[k2], k2 = l + 16 bin_of(q), back into the frame in natural order
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int k2 = l + 16 * bin_of(q);
                fr[k2 + (k2 >> 4)] = w[q];
            }
        }
        __syncthreads();
        // lane j: Xs[j + 256 r'] = sum_n1 W_P^(n1 r') W_ND^(n1 j) G_n1[j]
        cf g[P];
#pragma unroll
        for (int n1 = 0; n1 < P; n1++) g[n1] = lds[n1 * FRAME + js + (js >> 4)];
        {
            cf t = td1;
#pragma unroll
            for (int n1 = 1; n1 < P; n1++) {
                g[n1] = cmul1(g[n1], t);
                if (n1 + 1 < P) t = cmul1(t, td1);
            }
        }
        if constexpr (P == 8) fft8(g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7]);
        else if constexpr (P == 4) fft4(g[0], g[1], g[2], g[3]);
        else if constexpr (P == 2) { const cf a = g[0], c = g[1]; g[0] = a + c; g[1] = a - c; }
        // replicated spectrum times H, conjugated for the inverse (conj . FFT . conj)
        const cf *Hb = Hg;
        asm volatile("" : "+v"(Hb));
        cf u[16];
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            u[r] = g[r & (P - 1)];
            u[r + 1] = g[(r + 1) & (P - 1)];
            if (HREG) cmul2_conj(u[r], u[r + 1], Hr[r], Hr[r + 1]);
            else cmul2_conj(u[r], u[r + 1], Hb[256 * r], Hb[256 * (r + 1)]);
        }
        walk.publish(j);                                  // two barriers follow
        lds_barrier();                                    // every lane has read its frame values: the inverse below reuses the image
        dit_back(u, lds, j, tw3);           // conjugated on the last additions: u = the time samples
        const size_t room = n_out - b * S_out;
        const __amdgpu_buffer_rsrc_t ws = make_rsrc(out + b * S_out, (unsigned)((room < S_out ? room : S_out) * 8));
        const unsigned vbase = (unsigned)(j - Kov_out) * 8u;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int row = 256 * bin_of(q);
            if (row + 255 < Kov_out) continue;                // whole row dropped: uniform skip
            store_cf<2>(ws, vbase + (unsigned)row * 8u, u[q]);
        }
        if (!walk.advance()) break;
    }
    walk.finish(j);
}

// --------------------------------------------------------------------------------- //
// The interpolating filter with its SHORT FORWARD stage batched over G consecutive blocks -- the mirror of
// fir_cf32_ols4096_decim_batched_kernel.  The N/L-point forward transform of a block keeps 16 P of the 256 lanes busy (P = 16/L
// sub-transforms of 256 points); here the input windows of G blocks are loaded together (G P values per lane in flight), all
// G P sub-transforms run side by side on 16 lanes each, the lane's values of every block are parked in registers, and the
// blocks then go through twiddle, radix-P stage, H and the full inverse one after the other.  Three workgroups per CU.
// --------------------------------------------------------------------------------- //
template <int LOG2L, int LOG2G, bool HREG>
__global__ __launch_bounds__(256, 3) void fir_cf32_ols4096_interp_batched_kernel(const float2 *__restrict__ in, size_t in_elems, float2 *__restrict__ out,
                                                                                 size_t n_out, const float2 *__restrict__ Hspec, int Kov_in, int pad_in,
                                                                                 const float2 *__restrict__ twtab, size_t nblocks)
{
    constexpr int L = 1 << LOG2L, P = 16 / L, LOG2P = 4 - LOG2L, ND = 256 * P, G = 1 << LOG2G, T = G * P;
    static_assert(T <= 16, "at most 16 sub-transforms of 256 points fill the workgroup");
    constexpr int FRAME = 272;
    __shared__ cf lds[LDS_ELEMS];
    const int j = threadIdx.x;
    const size_t S_in = (size_t)(ND - Kov_in), S_out = S_in << LOG2L;
    const int Kov_out = Kov_in << LOG2L;
    const size_t ngroups = (nblocks + G - 1) / G;
    LaneTw tw3;
    load_pass3_twiddles(tw3, twtab, j);
    stage_pass2_twiddles(lds, twtab, j);
    const cf *Hg = reinterpret_cast<const cf *>(Hspec) + j;
    cf Hr[16];
    if (HREG) {
#pragma unroll
        for (int k = 0; k < 16; k++) Hr[k] = Hg[256 * k];
    }
    cf td1;
    {
        float sn, cs;
        sincospif(-2.0f * (float)spec_lane(j) / (float)ND, &sn, &cs);     // the lane takes the bins js + 256 r (decimator above)
        td1 = cf{cs, sn};
    }
    const int js = spec_lane(j);
    const int fi = j >> 4, l = j & 15;
    const bool sub = j < 16 * T;

    for (size_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        // the G input windows (ND samples each, from input index b*S_in - pad_in; what lies before the buffer or behind it reads 0)
        cf xx[T];
#pragma unroll
        for (int g = 0; g < G; g++) {
            const size_t b = grp * G + g;
            const size_t start = b * S_in;
            const size_t shift = start >= (size_t)pad_in ? 0 : (size_t)pad_in - start;
            const size_t first = start + shift - pad_in;
            const size_t left = (b < nblocks && in_elems > first) ? in_elems - first : 0;     // a block past the end loads nothing
            const size_t want = (size_t)ND - shift;
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(in + (b < nblocks ? first : 0), (unsigned)((left < want ? left : want) * 8));
#pragma unroll
            for (int i = 0; i < P; i++) {
                const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, (j + 256 * i - (int)shift) * 8, 0, 0);
                xx[g * P + i] = cf{__uint_as_float(t.x), __uint_as_float(t.y)};
            }
        }
        __syncthreads();                                  // the previous group's last inverse is done with the image
        // polyphase component n1 = n mod P of block g's window goes to sub-frame g P + n1 at position n / P  (n = j + 256 i)
#pragma unroll
        for (int g = 0; g < G; g++) {
#pragma unroll
            for (int i = 0; i < P; i++) {
                const int n2 = (j >> LOG2P) + (256 >> LOG2P) * i;
                lds[(g * P + (j & (P - 1))) * FRAME + n2 + (n2 >> 4)] = xx[g * P + i];
            }
        }
        __syncthreads();
        cf w[16];
        cf *fr = lds + fi * FRAME;
        if (sub) {
#pragma unroll
            for (int s = 0; s < 16; s++) w[s] = fr[l + 17 * s];
            fft16_plain(w);
        }
        wave_lds_order();                                 // a frame belongs to sixteen lanes of ONE wave from here to the stage's last read:
        if (sub) {                                        // no workgroup barrier between its passes (fft4096.hpp)
#pragma unroll
            for (int q = 0; q < 16; q++) fr[17 * l + bin_of(q)] = w[q];
        }
        wave_lds_order();
        if (sub) {
#pragma unroll
            for (int s = 0; s < 16; s++) w[s] = fr[l + 17 * s];
            LaneTw tl;
            const cf *t2 = lds + LDS_DATA + l;
#pragma unroll
            for (int p = 0; p < 3; p++) tl.a[p] = t2[p * 16];
#pragma unroll
            for (int p = 0; p < 12; p++) tl.c[p] = t2[(3 + p) * 16];
            fft16_tw(w, tl);
        }
        wave_lds_order();                                 // every lane of a frame has read its inputs (the frame's own sixteen lanes)
        if (sub) {
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int k2 = l + 16 * bin_of(q);
                fr[k2 + (k2 >> 4)] = w[q];
            }
        }
        __syncthreads();
        __builtin_amdgcn_s_setprio(1);                    // (the group's inverse transforms and stores ahead of another workgroup's forward stage)
        // park the lane's values of every block: the inverse passes below reuse the image
        cf gg[T];
#pragma unroll
        for (int t = 0; t < T; t++) gg[t] = lds[t * FRAME + js + (js >> 4)];
#pragma unroll
        for (int g = 0; g < G; g++) {
            const size_t b = grp * G + g;
            if (b >= nblocks) break;                      // (workgroup-uniform)
            // lane j: Xs[j + 256 r'] = sum_n1 W_P^(n1 r') W_ND^(n1 j) G_n1[j]
            cf gv[P];
#pragma unroll
            for (int n1 = 0; n1 < P; n1++) gv[n1] = gg[g * P + n1];
            {
                cf t = td1;
#pragma unroll
                for (int n1 = 1; n1 < P; n1++) {
                    gv[n1] = cmul1(gv[n1], t);
                    if (n1 + 1 < P) t = cmul1(t, td1);
                }
            }
            if constexpr (P == 8) fft8(gv[0], gv[1], gv[2], gv[3], gv[4], gv[5], gv[6], gv[7]);
            else if constexpr (P == 4) fft4(gv[0], gv[1], gv[2], gv[3]);
            else if constexpr (P == 2) { const cf a = gv[0], c = gv[1]; gv[0] = a + c; gv[1] = a - c; }
            const cf *Hb = Hg;
            asm volatile("" : "+v"(Hb));
            cf u[16];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                u[r] = gv[r & (P - 1)];
                u[r + 1] = gv[(r + 1) & (P - 1)];
                if (HREG) cmul2_conj(u[r], u[r + 1], Hr[r], Hr[r + 1]);
                else cmul2_conj(u[r], u[r + 1], Hb[256 * r], Hb[256 * (r + 1)]);
            }
            lds_barrier();                                // every lane has parked its values / read the previous block's samples: the image is reused
            dit_back(u, lds, j, tw3);           // conjugated on the last additions: u = the time samples
            const size_t room = n_out - b * S_out;
            const __amdgpu_buffer_rsrc_t ws = make_rsrc(out + b * S_out, (unsigned)((room < S_out ? room : S_out) * 8));
            const unsigned vbase = (unsigned)(j - Kov_out) * 8u;
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int row = 256 * bin_of(q);
                if (row + 255 < Kov_out) continue;
                store_cf<2>(ws, vbase + (unsigned)row * 8u, u[q]);
            }
        }
        __builtin_amdgcn_s_setprio(0);
    }
}

template <int LOG2L>
int launch_interp(const void *in, size_t in_elems, void *out, size_t n_iter, const void *Hspec, size_t K, const void *tw4096, void *sched, hipStream_t st)
{
    constexpr size_t L = (size_t)1 << LOG2L, ND = 4096 / L, A = 16 / L;    // Kov_in * L must be a multiple of 16
    const size_t Kov_in = (K - 1 + A - 1) / A * A, pad_in = Kov_in - (K - 1);
    if (Kov_in > ND / 2) { set_error("fir ols (interpolating): %zu taps per phase too long for L=%zu", K, L); return PCX_ERR_UNSUPPORTED; }
    const size_t S_in = ND - Kov_in;
    const size_t nblocks = (n_iter + S_in - 1) / S_in;
    const bool dyn = sched && nblocks > 2 * 1024 && PCX_ENV_SET("PCX_SCHED_RESAMPLERS");   // measured +-0 (0.1960 vs 0.1950 ms at L = 4): stride kept
    const unsigned grid = dyn ? 1024u : persistent_grid(nblocks, 1024);
    // Three workgroups per CU (170 VGPRs) instead of four: the kernel spills at 128 (12-52 bytes per lane), and the registers are worth more
    // than the fourth workgroup, as for the decimator -- L = 2 / 4 / 8 / 16: 268 / 309 / 227 / 263 -> 309 / 329 / 250 / 277 Gsamples/s of
    // output, with H held in registers too at L >= 8 (tools/ab_interp.sh, profiles/r02/ab_interp.txt).  PCX_INTERP_OCC=4 (diagnostic
    // library) brings the four-per-CU build back for A/B, PCX_INTERP_HREG overrides the H choice.
    // The short forward stage batched over 2^lgi blocks (fir_cf32_ols4096_interp_batched_kernel) where it measured faster, three interleaved
    // repeats on one box (tools/ab_interp.sh, profiles/r02/ab_interp.txt): L = 8 two blocks 240-243 -> 250-255 Gsamples/s of output, L = 16
    // four blocks 277-280 -> 300-305; L = 2 loses 10 % and L = 4 is inside the noise, so they keep the one-block kernel.
    // PCX_INTERP_G (diagnostic library) overrides: 0 = one block, 1 / 2 / 3 = two / four / eight.
    // (round 3, with the inverse on the sixteen-lane exchange: L = 4 two blocks with H in registers 320 -> 344 Gsamples/s at 1020 taps,
    // bench.py --workload interp4 0.471 -> 0.503 of HBM; profiles/r03/ab_interp.txt)
    // (the same sweep: L = 2 one block, H in registers 0.1841-0.1846 ms per 32 Mi inputs against 0.1898-0.1902 re-read; L = 4 two blocks, H in
    // registers 0.1679-0.1687; L = 8 four blocks, H in registers 0.2014-0.2017 against 0.2063-0.2076 for two)
    const int lgi = (int)PCX_ENV_INT("PCX_INTERP_G", LOG2L == 2 ? 1 : LOG2L >= 3 ? 2 : 0);
    if (!dyn && lgi > 0) {
        constexpr int LGMAX = LOG2L == 1 ? 1 : LOG2L == 2 ? 2 : 3;
        const int lg = lgi > LGMAX ? LGMAX : lgi;
        const bool hreg = PCX_ENV_INT("PCX_INTERP_HREG", LOG2L >= 2 ? 1 : 0) != 0;
        const size_t ngroups = (nblocks + ((size_t)1 << lg) - 1) >> lg;
        const unsigned gb = persistent_grid(ngroups, 768);
#define PCX_INTERP_ARGS dim3(gb), dim3(256), 0, st, (const float2 *)in, in_elems, (float2 *)out, n_iter * L, (const float2 *)Hspec, (int)Kov_in, \
                        (int)pad_in, (const float2 *)tw4096, nblocks
#define PCX_INTERP_LAUNCH(LG)                                                                                                   \
        do {                                                                                                                    \
            if (hreg) hipLaunchKernelGGL((fir_cf32_ols4096_interp_batched_kernel<LOG2L, LG, true>), PCX_INTERP_ARGS);              \
            else hipLaunchKernelGGL((fir_cf32_ols4096_interp_batched_kernel<LOG2L, LG, false>), PCX_INTERP_ARGS);                  \
        } while (0)
        if (lg == 1) PCX_INTERP_LAUNCH(1);
        else if (lg == 2) PCX_INTERP_LAUNCH((LGMAX >= 2 ? 2 : 1));
        else PCX_INTERP_LAUNCH((LGMAX >= 3 ? 3 : 1));
#undef PCX_INTERP_LAUNCH
#undef PCX_INTERP_ARGS
        PCX_LAUNCH_CHECK();
        return PCX_OK;
    }
    if (!dyn && PCX_ENV_INT("PCX_INTERP_OCC", 3) == 3) {
        const unsigned g3 = persistent_grid(nblocks, 768);
        if (PCX_ENV_INT("PCX_INTERP_HREG", 1) != 0)
            hipLaunchKernelGGL((fir_cf32_ols4096_interp_kernel<LOG2L, false, 3, true>), dim3(g3), dim3(256), 0, st, (const float2 *)in, in_elems, (float2 *)out,
                               n_iter * L, (const float2 *)Hspec, (int)Kov_in, (int)pad_in, (const float2 *)tw4096, nblocks, (pcx::SchedState *)nullptr);
        else
            hipLaunchKernelGGL((fir_cf32_ols4096_interp_kernel<LOG2L, false, 3, false>), dim3(g3), dim3(256), 0, st, (const float2 *)in, in_elems, (float2 *)out,
                               n_iter * L, (const float2 *)Hspec, (int)Kov_in, (int)pad_in, (const float2 *)tw4096, nblocks, (pcx::SchedState *)nullptr);
        PCX_LAUNCH_CHECK();
        return PCX_OK;
    }
    if (dyn)
        hipLaunchKernelGGL((fir_cf32_ols4096_interp_kernel<LOG2L, true>), dim3(grid), dim3(256), 0, st, (const float2 *)in, in_elems, (float2 *)out,
                           n_iter * L, (const float2 *)Hspec, (int)Kov_in, (int)pad_in, (const float2 *)tw4096, nblocks, (pcx::SchedState *)sched);
    else
        hipLaunchKernelGGL((fir_cf32_ols4096_interp_kernel<LOG2L, false>), dim3(grid), dim3(256), 0, st, (const float2 *)in, in_elems, (float2 *)out,
                           n_iter * L, (const float2 *)Hspec, (int)Kov_in, (int)pad_in, (const float2 *)tw4096, nblocks, (pcx::SchedState *)nullptr);
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}

}  // namespace

// n_iter full-rate iterations (a multiple of M) -> n_iter / M outputs, M even: M = M1 * M2 with M1 the largest of 16 / 8 / 4 / 2
// dividing it (folded into the spectrum) and M2 the cofactor (kept one in M2 on the store).
// Hspec = FFT_4096(h)[k] * exp(+j 2 pi k (M1-1) / 4096) / 4096.
size_t fir_decim_fold_factor(size_t M) { return M % 16 == 0 ? 16 : M % 8 == 0 ? 8 : M % 4 == 0 ? 4 : M % 2 == 0 ? 2 : 1; }
int launch_fir_cf32_ols4096_decim(const void *in, size_t in_elems, void *out, size_t n_iter, const void *Hspec, size_t K, size_t M,
                                  const void *tw4096, void *sched, hipStream_t st)
{
    if (n_iter == 0) return PCX_OK;
    if (K < 1 || K > 2049) { set_error("fir ols (decimating): K=%zu outside 1..2049", K); return PCX_ERR_UNSUPPORTED; }
    const size_t M1 = fir_decim_fold_factor(M), M2 = M / M1;
    if (M2 > 65535) { set_error("fir ols (decimating): M=%zu too large", M); return PCX_ERR_UNSUPPORTED; }
    switch (M1) {
    case 2: return launch_decim<1>(in, in_elems, out, n_iter, Hspec, K, tw4096, M2, sched, st);
    case 4: return launch_decim<2>(in, in_elems, out, n_iter, Hspec, K, tw4096, M2, sched, st);
    case 8: return launch_decim<3>(in, in_elems, out, n_iter, Hspec, K, tw4096, M2, sched, st);
    case 16: return launch_decim<4>(in, in_elems, out, n_iter, Hspec, K, tw4096, M2, sched, st);
    }
    set_error("fir ols (decimating): M=%zu is odd", M);
    return PCX_ERR_UNSUPPORTED;
}

// n_iter input iterations -> n_iter * L outputs.  Hspec = FFT_4096(all taps) / 4096; K = taps per polyphase row.
int launch_fir_cf32_ols4096_interp(const void *in, size_t in_elems, void *out, size_t n_iter, const void *Hspec, size_t K, size_t L,
                                   const void *tw4096, void *sched, hipStream_t st)
{
    if (n_iter == 0) return PCX_OK;
    switch (L) {
    case 2: return launch_interp<1>(in, in_elems, out, n_iter, Hspec, K, tw4096, sched, st);
    case 4: return launch_interp<2>(in, in_elems, out, n_iter, Hspec, K, tw4096, sched, st);
    case 8: return launch_interp<3>(in, in_elems, out, n_iter, Hspec, K, tw4096, sched, st);
    case 16: return launch_interp<4>(in, in_elems, out, n_iter, Hspec, K, tw4096, sched, st);
    }
    set_error("fir ols (interpolating): L=%zu is not 2, 4, 8 or 16", L);
    return PCX_ERR_UNSUPPORTED;
}

}  // namespace pcx
