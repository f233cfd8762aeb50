// fir_ols_part.hip -- /comms/fir_filter, complex_float32, M = L = 1, 2049 < K <= 8193 taps: overlap-save with the TAPS PARTITIONED.
//
// The same y[n] = sum_k h[k] x[n-k] as FIRFilter.cpp:294-300.  fir_ols.hip evaluates it per 4096-sample block as
// IFFT(FFT(block) .* H) and stops at K - 1 = 2048: a block cannot be shorter than the filter.  Until round 6 longer filters
// went to 8192- / 16384-sample blocks (fir_ols_r16.hip, removed: 512 / 1024 lanes per block, six / eight LDS exchanges per
// transform pair, half of every window overlap) at 0.29 / 0.18 of the HBM rate.  Here the BLOCK stays 4096 samples and the filter is cut
// instead: h = h_0 + z^-B h_1 + ... + z^-(P-1)B h_(P-1), B = 2048 taps each (the last one up to B + 1), and
//     y_b = IFFT( X_b . H_0 + X_(b-1) . H_1 + ... + X_(b-P+1) . H_(P-1) ) [B .. 2B)        X_m = FFT( x[mB + off .. mB + off + 4096) )
// -- ONE forward and ONE inverse 4096-point transform per B outputs whatever K, plus P multiply-adds per bin against spectra the
// workgroup computed one, two, ... blocks ago and still holds in registers.  So a workgroup walks a CONTIGUOUS run of blocks
// (and computes the P - 1 spectra in front of its run first: forward transforms only), the transform pair is fir_ols.hip's
// (fft4096.hpp: three barriers per block, the spectrum digit-reversed across lanes), and the window's lower half -- the
// previous window's upper half -- is not fetched again either: it is kept in registers in the form it arrived in.
//
// Index algebra.  out[n] = sum_k h[k] in[n + K-1 - k] (the buffer carries K - 1 samples of history in front, as for every plan).
// Window m is in[mB + off + i], i = 0 .. 4095, off = K - 1 - B; partition p of output block b (outputs bB .. bB + B - 1) is
// the circular convolution of window b - p with h_p, whose samples i >= B are free of wrap-around (h_p has at most B + 1 taps)
// and sample B + t is exactly sum_k h_p[k] in[bB + t + K-1 - pB - k].  Windows in front of the buffer (m < 0 at the start of
// the stream) read as zero through the descriptor's range check; they only meet taps that do not exist.
//
// REAL float32 streams (real taps) ride the same kernel two at a time: a real filter does not mix the parts of a complex stream, so
// z[t] = x[t] + i x[t + D] -- the call's first half beside its second, D = half the outputs -- is filtered as ONE complex stream of
// half the length and y[t] = Re, y[t + D] = Im.  Only the fetches and the stores differ (two 4-byte accesses per element instead
// of one of 8); the second half's history is the end of the first half, real samples in the same buffer.
//
// DECIMATING filters (L = 1, M > 1, FIRFilter.cpp:286-302: the output with (n + 1) % M == 0 is kept) run the same blocks at the full
// rate and keep one output in M at the store -- 200 x the time-domain tile they fell to beyond 2049 taps.  (Up to 2049 taps the
// folded-spectrum kernel of fir_ols_decim.hip also shortens the inverse transform; here the spectra of P windows would have to be
// folded each.)  A real pair's halves then split at a multiple of M, so that both halves keep the same phase.
//
// Registers: P spectra of 16 bins (32 VGPRs each) + the product + the pass-3 factors (30) + the kept half window (16): every P
// runs two workgroups per CU on up to 256 VGPRs.  P = 2 holds its bins of H_0 and H_1 in registers as well (64); P = 3 and 4 read
// the H_p from L2 in every block (96 / 128 KB a table, the same for every workgroup), stored the way the lanes hold the spectrum
// (pcx_tables.hpp turn_spectrum_lanes), two partitions to a 16-byte entry.
#include "fft4096.hpp"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "pcx_internal.hpp"

namespace pcx {

namespace {
using namespace fft4k;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kHop = 2048;   // B: outputs per block, taps per partition

// (ra, rb) = (a wa, b wb): the first product of a bin pair
__device__ __forceinline__ void mac_first(cf &ra, cf &rb, cf a, cf b, cf wa, cf wb)
{
    asm("v_pk_mul_f32 %0, %2, %4 op_sel:[1,1] op_sel_hi:[0,1] neg_lo:[0,1]\n\t"   // (-a.y w.y, a.x w.y)
        "v_pk_mul_f32 %1, %3, %5 op_sel:[1,1] op_sel_hi:[0,1] neg_lo:[0,1]\n\t"
        "v_pk_fma_f32 %0, %2, %4, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"        // (a.x w.x, a.y w.x) + t
        "v_pk_fma_f32 %1, %3, %5, %1 op_sel:[0,0,0] op_sel_hi:[1,0,1]"
        : "=&v"(ra), "=&v"(rb)
        : "v"(a), "v"(b), "v"(wa), "v"(wb));
}
// (ra, rb) += (a wa, b wb);  CONJ: (ra, rb) = conj of that sum -- the inverse transform runs on the forward passes, on conj(Y)
template <bool CONJ>
__device__ __forceinline__ void mac_next(cf &ra, cf &rb, cf a, cf b, cf wa, cf wb)
{
    if (CONJ) {
        asm("v_pk_fma_f32 %0, %2, %4, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]\n\t"      // (r.x - a.y w.y, r.y + a.x w.y)
            "v_pk_fma_f32 %1, %3, %5, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]\n\t"
            "v_pk_fma_f32 %0, %2, %4, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_hi:[1,0,1]\n\t"      // (t.x + a.x w.x, -t.y - a.y w.x)
            "v_pk_fma_f32 %1, %3, %5, %1 op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_hi:[1,0,1]"
            : "+v"(ra), "+v"(rb)
            : "v"(a), "v"(b), "v"(wa), "v"(wb));
    } else {
        asm("v_pk_fma_f32 %0, %2, %4, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]\n\t"
            "v_pk_fma_f32 %1, %3, %5, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]\n\t"
            "v_pk_fma_f32 %0, %2, %4, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
            "v_pk_fma_f32 %1, %3, %5, %1 op_sel:[0,0,0] op_sel_hi:[1,0,1]"
            : "+v"(ra), "+v"(rb)
            : "v"(a), "v"(b), "v"(wa), "v"(wb));
    }
}

// the lane's bin k of partitions 2g and 2g + 1 (plane g of the table: [16][256] entries of 16 bytes; the last plane of an odd P
// holds one partition in 8-byte entries)
template <int P>
struct HTable {
    static constexpr int kPlanes = (P + 1) / 2;
    static constexpr size_t plane_bytes(int g) { return (size_t)4096 * ((2 * g + 1 < P) ? 16 : 8); }
    static constexpr size_t plane_off(int g) { return g == 0 ? 0 : plane_off(g - 1) + plane_bytes(g - 1); }
    static constexpr size_t kBytes = plane_off(kPlanes);
};

// A block of a run (`step` below).  X[(R + p) % P] holds: p = 0 the block's window as fetched (transformed in place), p >= 1 the
// spectrum of window b - p.  Behind the block the OLDEST slot is free: the next window is assembled in it, and the next step runs
// with R' = (R + P - 1) % P -- the block loop is unrolled P times so that every slot is a compile-time register range.
// KEEP: the window's lower half from registers -- the previous window's upper half, kept as it arrived (16 VGPRs) -- instead of
//   from memory again (+3.5 %).
// HREG: the lane's bins of every H_p in registers across the run (P = 2: 64 VGPRs) instead of from L2 in every block.
// BP: bin pairs per table batch (H from L2): the first batch is requested in front of the forward transform's last stage, each
//   further one in front of the multiply-adds of the batch before it -- read at their use, two bins at a time, every block waited
//   for eight L2 round trips.
// Measured, 64 Mi samples (tools/ab_upols.sh, profiles/r06/ab_upols.txt): K = 4097 -- three workgroups per CU with H from L2 0.345 ms,
// two with H in registers 0.327; the pass-3 factors in LDS instead of registers (to make room for the next window's upper half a
// whole block early) 0.345 either way: thirty more LDS reads per transform are not hidden at two waves per SIMD, and the earlier
// fetch bought nothing -- memory latency is not what a block waits for.  K = 8193 -- 0.403 ms.
// REAL: `in` / `out` are float streams, `half` = D (outputs [0, half) are the real part's, [half, n_out) the imaginary part's);
// n_out, in_elems in real samples, nblocks = blocks of the first half.
// DECIM: n_out counts full-rate outputs, n_dec the stored ones; magic = ceil(2^32 / M) (exact quotient for t * M < 2^32).
template <int P, bool KEEP, bool HREG, int BP, bool REAL, bool DECIM>
__global__ __launch_bounds__(256, 2) void fir_cf32_upols_kernel(const void *__restrict__ in_, size_t in_elems, void *__restrict__ out_,
                                                                  size_t n_out, const unsigned char *__restrict__ Hparts, long long off,
                                                                  const float2 *__restrict__ twtab, size_t nblocks, size_t half,
                                                                  size_t n_dec, unsigned M, unsigned magic)
{
    __shared__ cf lds[LDS_ELEMS];
    const int j = threadIdx.x;
    const float2 *in = static_cast<const float2 *>(in_);
    float2 *out = static_cast<float2 *>(out_);
    const float *inr = static_cast<const float *>(in_);
    float *outr = static_cast<float *>(out_);
    // this workgroup's run of blocks: a balanced partition of 0 .. nblocks - 1
    const size_t q = nblocks / gridDim.x, rem = nblocks % gridDim.x, w = blockIdx.x;
    const size_t b0 = w * q + (w < rem ? w : rem), b1 = b0 + q + (w < rem ? 1 : 0);
    if (b0 >= b1) return;

    LaneTw tw3;
    load_pass3_twiddles(tw3, make_rsrc(twtab, TW_TABLE_ELEMS * 8), j);
    stage_pass2_twiddles(lds, twtab, j);          // (in front of the first dif_rest's barriers)

    // Window m, whole (the prologue of a run).  Windows inside the buffer take one descriptor and scalar row offsets; a window that
    // starts in front of the buffer or ends behind it goes through the range check lane by lane (reads 0 outside).
    // Non-temporal, except the upper half of a window that the next block fetches again as its lower half (no KEEP).
    // one part of a real pair: 4-byte samples from inr + s_h, the range check lane by lane.  A lane in front of the buffer gets a fixed
    // out-of-range offset, not its (negative, wrapped) own: written as (j + 256 r - shift) * 4 the compiler keeps (j - shift) * 4 in
    // the register and 1024 r in the instruction's offset field, and a buffer_load_dword whose register part is wrapped DROPS the
    // valid lanes that share a four-lane group with lanes still out of range after the addition (tools/bufload_quad_lab.hip,
    // profiles/r06/bufload_quad_lab.txt: dword loads with a non-zero instruction offset only; dwordx2 loads are not affected)
    auto fetch_part = [&](float (&dst)[16], long long s_h) {
        const long long shift = s_h < 0 ? -s_h : 0, first = s_h + shift;
        const long long left = (long long)in_elems > first ? (long long)in_elems - first : 0, want = 4096 - shift;
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(inr + first, (unsigned)((left < want ? left : want) * 4));
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int e = j + 256 * r - (int)shift;
            dst[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, e < 0 ? 0x7ffffff0 : e * 4, 0, kAuxStream));
        }
    };
    auto fetch = [&](cf (&dst)[16], long long m) {
        constexpr int r0 = 0, d0 = 0;
        const long long s = m * kHop + off;
        if (REAL) {
            float re[16], im[16];
            fetch_part(re, s);
            fetch_part(im, s + (long long)half);
#pragma unroll
            for (int r = 0; r < 16; r++) dst[r] = cf{re[r], im[r]};
        } else if (s >= 0 && (size_t)s + 4096 <= in_elems) {
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(in + s, 4096 * 8);
#pragma unroll
            for (int r = r0; r < 16; r++) {
                const u32x2 t = (KEEP || r < 8) ? __builtin_amdgcn_raw_buffer_load_b64(rs, j * 8, 2048 * r, kAuxStream)
                                                : __builtin_amdgcn_raw_buffer_load_b64(rs, j * 8, 2048 * r, 0);
                dst[r - d0] = cf{__uint_as_float(t.x), __uint_as_float(t.y)};
            }
        } else {
            const long long shift = s < 0 ? -s : 0, first = s + shift;
            const long long left = (long long)in_elems > first ? (long long)in_elems - first : 0, want = 4096 - shift;
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(in + first, (unsigned)((left < want ? left : want) * 8));
#pragma unroll
            for (int r = r0; r < 16; r++) {
                const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, (j + 256 * r - (int)shift) * 8, 0, kAuxStream);
                dst[r - d0] = cf{__uint_as_float(t.x), __uint_as_float(t.y)};
            }
        }
    };
    // the same inside the block loop, where the window never starts in front of the buffer (m >= 1): ONE path, every load
    // unconditional and range-checked -- behind a branch the registers a fetch fills are a merge of old and new values, and the
    // compiler parks the new ones in copies behind `s_waitcnt vmcnt(0)` right where they were requested
    auto fetch_next = [&](auto &dst, size_t m, auto r0c, auto d0c) {
        constexpr int r0 = decltype(r0c)::value, d0 = decltype(d0c)::value;
        const size_t s = m * kHop + (size_t)off;
        if (REAL) {
            const size_t sb = s + half;
            const size_t la = in_elems > s ? in_elems - s : 0, lb = in_elems > sb ? in_elems - sb : 0;
            const __amdgpu_buffer_rsrc_t ra = make_rsrc(inr + (la ? s : 0), (unsigned)((la < 4096 ? la : 4096) * 4));
            const __amdgpu_buffer_rsrc_t rb = make_rsrc(inr + (lb ? sb : 0), (unsigned)((lb < 4096 ? lb : 4096) * 4));
#pragma unroll
            for (int r = r0; r < 16; r++) {
                const unsigned a = __builtin_amdgcn_raw_buffer_load_b32(ra, (j + 256 * r) * 4, 0, kAuxStream);
                const unsigned c = __builtin_amdgcn_raw_buffer_load_b32(rb, (j + 256 * r) * 4, 0, kAuxStream);
                dst[r - d0] = cf{__uint_as_float(a), __uint_as_float(c)};
            }
            return;
        }
        const size_t left = in_elems > s ? in_elems - s : 0;
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(in + (left ? s : 0), (unsigned)((left < 4096 ? left : 4096) * 8));
#pragma unroll
        for (int r = r0; r < 16; r++) {
            const u32x2 t = (KEEP || r < 8) ? __builtin_amdgcn_raw_buffer_load_b64(rs, (j + 256 * r) * 8, 0, kAuxStream)
                                             : __builtin_amdgcn_raw_buffer_load_b64(rs, (j + 256 * r) * 8, 0, 0);
            dst[r - d0] = cf{__uint_as_float(t.x), __uint_as_float(t.y)};
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I8 = std::integral_constant<int, 8>;
    const __amdgpu_buffer_rsrc_t hs = make_rsrc(Hparts, (unsigned)HTable<P>::kBytes);
    // the lane's bin k of every partition
    auto hbin = [&](cf (&h)[P], int k) {
#pragma unroll
        for (int g = 0; g < HTable<P>::kPlanes; g++) {
            if (2 * g + 1 < P) {
                const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(hs, j * 16, (int)HTable<P>::plane_off(g) + 4096 * k, 0);
                h[2 * g] = cf{__uint_as_float(t.x), __uint_as_float(t.y)};
                h[2 * g + 1] = cf{__uint_as_float(t.z), __uint_as_float(t.w)};
            } else {
                h[2 * g] = load_cf(hs, j * 8, (int)HTable<P>::plane_off(g) + 2048 * k);
            }
        }
    };
    constexpr int NB = 8 / BP;
    struct HBatch { cf h0[BP][P], h1[BP][P]; };
    auto hload = [&](HBatch &hb, auto bc) {
        constexpr int bi = decltype(bc)::value;
#pragma unroll
        for (int i = 0; i < BP; i++) {
            hbin(hb.h0[i], bin_of(2 * (bi * BP + i)));
            hbin(hb.h1[i], bin_of(2 * (bi * BP + i) + 1));
        }
    };
    auto first_pass = [&](cf (&v)[16]) { dif_a_math(v, tw3); };

    cf X[P][16];
    cf keep[KEEP ? 8 : 1];         // the upper half of the newest window as it arrived = the lower half of the next
    cf Hr[HREG ? P : 1][16];
    if (HREG) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            cf h[P];
            hbin(h, k);
#pragma unroll
            for (int p = 0; p < P; p++) Hr[p][k] = h[p];
        }
    }
    // the spectra in front of the run: windows b0 - P + 1 .. b0 - 1 into slots P - 1 .. 1
#pragma unroll
    for (int p = P - 1; p >= 1; p--) {
        fetch(X[p], (long long)b0 - p);
        first_pass(X[p]);
        dif_rest(X[p], lds, j);
    }
    fetch(X[0], (long long)b0);

    size_t b = b0;
    auto step = [&](auto rc) {
        constexpr int R = decltype(rc)::value, NX = (R + P - 1) % P;
        cf(&v)[16] = X[R];
        if (KEEP) {
#pragma unroll
            for (int r = 0; r < 8; r++) keep[r] = v[8 + r];
        }
        // u = conj( sum_p X_(b-p) . H_p ), bin pairs; natural register order for the inverse's first pass
        cf u[16];
        auto mac_pair = [&](int qq, const cf (&h0)[P], const cf (&h1)[P]) {
            const int k0 = bin_of(qq), k1 = bin_of(qq + 1);
            if (P == 1) {            // one partition: the plain product, conjugated
                u[k0] = v[qq];
                u[k1] = v[qq + 1];
                cmul2_conj(u[k0], u[k1], h0[0], h1[0]);
                return;
            }
            mac_first(u[k0], u[k1], v[qq], v[qq + 1], h0[0], h1[0]);
#pragma unroll
            for (int p = 1; p < P; p++) {
                if (p == P - 1) mac_next<true>(u[k0], u[k1], X[(R + p) % P][qq], X[(R + p) % P][qq + 1], h0[p], h1[p]);
                else mac_next<false>(u[k0], u[k1], X[(R + p) % P][qq], X[(R + p) % P][qq + 1], h0[p], h1[p]);
            }
        };
        first_pass(v);
        if (HREG) {
            dif_rest(v, lds, j);
#pragma unroll
            for (int qq = 0; qq < 16; qq += 2) {
                cf h0[P], h1[P];
#pragma unroll
                for (int p = 0; p < P; p++) { h0[p] = Hr[p][bin_of(qq)]; h1[p] = Hr[p][bin_of(qq + 1)]; }
                mac_pair(qq, h0, h1);
            }
        } else {
            HBatch hb[NB];
            dif_rest(v, lds, j, [&]() { hload(hb[0], I0()); });
            __builtin_amdgcn_sched_barrier(0);
            auto batch = [&](auto bc) {
                constexpr int bi = decltype(bc)::value, nb = bi + 1 < NB ? bi + 1 : bi;
                if (bi + 1 < NB) hload(hb[nb], std::integral_constant<int, nb>());
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < BP; i++) mac_pair(2 * (bi * BP + i), hb[bi].h0[i], hb[bi].h1[i]);
                __builtin_amdgcn_sched_barrier(0);
            };
            batch(I0());
            if (NB > 1) batch(std::integral_constant<int, (NB > 1 ? 1 : 0)>());
            if (NB > 2) { batch(std::integral_constant<int, (NB > 2 ? 2 : 0)>()); batch(std::integral_constant<int, (NB > 2 ? 3 : 0)>()); }
            if (NB > 4) { batch(std::integral_constant<int, (NB > 4 ? 4 : 0)>()); batch(std::integral_constant<int, (NB > 4 ? 5 : 0)>());
                          batch(std::integral_constant<int, (NB > 4 ? 6 : 0)>()); batch(std::integral_constant<int, (NB > 4 ? 7 : 0)>()); }
        }
        // the next window into the slot of the oldest spectrum, while the inverse transform runs
        cf(&nx)[16] = X[NX];
        const bool more = b + 1 < b1;
        if (KEEP) {
#pragma unroll
            for (int r = 0; r < 8; r++) nx[r] = keep[r];
            fetch_next(nx, b + 1, I8(), I0());
        } else {
            fetch_next(nx, b + 1, I0(), I0());
        }
        __builtin_amdgcn_s_setprio(1);
        dit_back(u, lds, j, tw3);
        // time sample i = j + 256 bin_of(q) >= B of the block is output bB + i - B; outputs past n_out fall to the range check
        if (DECIM) {
            // full-rate output n = o + t of the block (of either half of a real pair: `half` is a multiple of M) is kept when
            // (n + 1) % M == 0, at (n + 1) / M - 1; everything else gets an offset outside the descriptor, no branch
            const size_t o = b * kHop, B0 = o / M;
            const unsigned base = (unsigned)(o - B0 * M);
            constexpr size_t kMost = kHop / 2 + 2;             // outputs a block can keep (M >= 2)
            constexpr int EB = REAL ? 4 : 8;
            const size_t lim_a = REAL ? half : n_out, dec_a = REAL ? half / M : n_dec;
            const size_t left_a = lim_a > o ? lim_a - o : 0, room_a = dec_a > B0 ? dec_a - B0 : 0;
            const __amdgpu_buffer_rsrc_t wa = make_rsrc(static_cast<unsigned char *>(out_) + B0 * EB, (unsigned)((room_a < kMost ? room_a : kMost) * EB));
            const size_t left_b = REAL && n_out > half + o ? n_out - half - o : 0, B0b = half / M + B0, room_b = REAL && n_dec > B0b ? n_dec - B0b : 0;
            const __amdgpu_buffer_rsrc_t wb = make_rsrc(static_cast<unsigned char *>(out_) + (room_b ? B0b * EB : 0), (unsigned)((room_b < kMost ? room_b : kMost) * EB));
#pragma unroll
            for (int qq = 0; qq < 16; qq++) {
                const int row = bin_of(qq);
                if (row < 8) continue;
                const unsigned t = (unsigned)(j + 256 * (row - 8)), tt = base + t + 1u, qt = __umulhi(tt, magic);
                const bool hit = qt * M == tt;
                const int at = (int)((qt - 1u) * (unsigned)EB);
                if (REAL) {
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(u[qq].x), wa, hit && t < left_a ? at : 0x7ffffff0, 0, kAuxStream);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(u[qq].y), wb, hit && t < left_b ? at : 0x7ffffff0, 0, kAuxStream);
                } else {
                    store_cf<kAuxStream>(wa, (unsigned)(hit && t < left_a ? at : 0x7ffffff0), u[qq]);
                }
            }
        } else if (REAL) {
            const size_t o = b * kHop, na = half - o, nb = n_out - half > o ? n_out - half - o : 0;       // (o < half: nblocks covers the first half)
            const __amdgpu_buffer_rsrc_t wa = make_rsrc(outr + o, (unsigned)((na < (size_t)kHop ? na : (size_t)kHop) * 4));
            const __amdgpu_buffer_rsrc_t wb = make_rsrc(outr + (nb ? half + o : 0), (unsigned)((nb < (size_t)kHop ? nb : (size_t)kHop) * 4));
#pragma unroll
            for (int qq = 0; qq < 16; qq++) {
                const int row = bin_of(qq);
                if (row < 8) continue;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(u[qq].x), wa, (j + 256 * (row - 8)) * 4, 0, kAuxStream);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(u[qq].y), wb, (j + 256 * (row - 8)) * 4, 0, kAuxStream);
            }
        } else {
            const size_t room = n_out - b * kHop;
            const __amdgpu_buffer_rsrc_t ws = make_rsrc(out + b * kHop, (unsigned)((room < (size_t)kHop ? room : (size_t)kHop) * 8));
#pragma unroll
            for (int qq = 0; qq < 16; qq++) {
                const int row = bin_of(qq);
                if (row < 8) continue;
                store_cf<kAuxStream>(ws, (unsigned)j * 8u + (unsigned)(row - 8) * 2048u, u[qq]);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        b++;
        return more;
    };
    for (;;) {
        if (!step(I0())) break;
        if (P >= 2) { if (!step(std::integral_constant<int, (P - 1) % P>())) break; }
        if (P >= 3) { if (!step(std::integral_constant<int, (P >= 3 ? P - 2 : 0)>())) break; }
        if (P >= 4) { if (!step(std::integral_constant<int, (P >= 4 ? P - 3 : 0)>())) break; }
    }
}

template <int P, bool KEEP, bool HREG, int BP, bool REAL, bool DECIM>
int launch_parts(const void *in, size_t in_elems, void *out, size_t n_out, const void *Hparts, size_t K, const void *tw, size_t M, hipStream_t st)
{
    // REAL: the first `half` outputs beside the rest -- a multiple of 32 (whole 128-byte lines for the second half's rows) and of M
    // (both halves keep the same phase of the decimator)
    size_t half = 0;
    if (REAL) {
        size_t q = 32;
        if (DECIM) { size_t a = M, b = 32; while (b) { const size_t t = a % b; a = b; b = t; } q = M / a * 32; }
        half = std::min(n_out, ((n_out + 1) / 2 + q - 1) / q * q);
    }
    const size_t nblocks = ((REAL ? half : n_out) + kHop - 1) / kHop;
    // a run pays P - 1 forward transforms before its first output: runs of at least 4 (P - 1) blocks while the call has them
    const long oversub = PCX_ENV_INT("PCX_UPOLS_OVERSUB", 1);      // (diagnostic library: workgroups queued per slot, A/B)
    const unsigned slots = 256u * 2u * (unsigned)(oversub > 0 ? oversub : 1);      // two workgroups per CU
    const size_t cap = g_link_grid ? (size_t)g_link_grid : (size_t)slots;
    const size_t min_run = P > 1 ? 4 * (size_t)(P - 1) : 1;
    size_t grid = (nblocks + min_run - 1) / min_run;
    if (grid > cap) grid = cap;
    if (grid < 1) grid = 1;
    const unsigned magic = DECIM ? (unsigned)(((1ull << 32) + M - 1) / M) : 0u;
    hipLaunchKernelGGL((fir_cf32_upols_kernel<P, KEEP, HREG, BP, REAL, DECIM>), dim3((unsigned)grid), dim3(256), 0, st, in, in_elems, out, n_out,
                       (const unsigned char *)Hparts, (long long)(K - 1) - kHop, (const float2 *)tw, nblocks, half, n_out / (DECIM ? M : 1),
                       (unsigned)M, magic);
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}

}  // namespace

size_t fir_upols_table_bytes(int parts)
{
    return parts == 1 ? HTable<1>::kBytes : parts == 2 ? HTable<2>::kBytes : parts == 3 ? HTable<3>::kBytes : parts == 4 ? HTable<4>::kBytes : 0;
}

// parts = ceil((K - 1) / 2048) in 2 .. 4 (1, K <= 2049: real decimating filters only); Hparts = the partitions' spectra
// (pcx_tables.hpp make_hparts), tw = make_tw4096();
// real_stream: float32 samples and outputs (real taps), else complex_float32; M > 1: n_out full-rate outputs (a multiple of M), one in
// M stored
int launch_fir_cf32_upols(const void *in, size_t in_elems, void *out, size_t n_out, const void *Hparts, size_t K, int parts, const void *tw,
                          hipStream_t st, bool real_stream, size_t M)
{
    if (n_out == 0) return PCX_OK;
    if (K < 1 || std::max<size_t>(1, (K - 1 + kHop - 1) / kHop) != (size_t)parts) { set_error("fir partitioned ols: K=%zu does not make %d partitions", K, parts); return PCX_ERR_UNSUPPORTED; }
    if (M < 1 || M > 65535) { set_error("fir partitioned ols: decimation %zu outside 1 .. 65535", M); return PCX_ERR_UNSUPPORTED; }
    const int variant = (int)PCX_ENV_INT("PCX_UPOLS_VARIANT", 0);   // (diagnostic library: A/B)
#define PCX_UPOLS(P, KEEP, HREG, BP)                                                                                                  \
    do {                                                                                                                              \
        if (real_stream && M > 1) return launch_parts<P, KEEP, HREG, BP, true, true>(in, in_elems, out, n_out, Hparts, K, tw, M, st);   \
        if (real_stream) return launch_parts<P, KEEP, HREG, BP, true, false>(in, in_elems, out, n_out, Hparts, K, tw, M, st);           \
        if (M > 1) return launch_parts<P, KEEP, HREG, BP, false, true>(in, in_elems, out, n_out, Hparts, K, tw, M, st);                 \
        return launch_parts<P, KEEP, HREG, BP, false, false>(in, in_elems, out, n_out, Hparts, K, tw, M, st);                           \
    } while (0)
    const bool plain = !real_stream && M == 1;
    switch (parts) {
    case 1:
        // one partition = plain overlap-save advancing by 2048 whatever K <= 2049: kept for what has no better kernel -- REAL
        // decimating filters (two halves per transform, one output in M stored: twice the double-precision pipeline they ran on)
        if (real_stream && M > 1) return launch_parts<1, true, true, 8, true, true>(in, in_elems, out, n_out, Hparts, K, tw, M, st);
        break;
    case 2:
#ifdef PCX_DIAG
        if (variant == 1 && plain) return launch_parts<2, true, false, 4, false, false>(in, in_elems, out, n_out, Hparts, K, tw, M, st);
        if (variant == 2 && plain) return launch_parts<2, false, true, 8, false, false>(in, in_elems, out, n_out, Hparts, K, tw, M, st);
#endif
        PCX_UPOLS(2, true, true, 8);
    case 3:
#ifdef PCX_DIAG
        if (variant == 1 && plain) return launch_parts<3, true, false, 1, false, false>(in, in_elems, out, n_out, Hparts, K, tw, M, st);
#endif
        PCX_UPOLS(3, true, false, 2);
    case 4:
#ifdef PCX_DIAG
        if (variant == 1 && plain) return launch_parts<4, false, false, 2, false, false>(in, in_elems, out, n_out, Hparts, K, tw, M, st);
#endif
        // (decimating: the store's index arithmetic does not fit beside the kept half window in 256 registers; whole windows fetched)
        if (real_stream && M > 1) return launch_parts<4, false, false, 1, true, true>(in, in_elems, out, n_out, Hparts, K, tw, M, st);
        if (M > 1) return launch_parts<4, false, false, 1, false, true>(in, in_elems, out, n_out, Hparts, K, tw, M, st);
        if (real_stream) return launch_parts<4, true, false, 1, true, false>(in, in_elems, out, n_out, Hparts, K, tw, M, st);
        return launch_parts<4, true, false, 1, false, false>(in, in_elems, out, n_out, Hparts, K, tw, M, st);
    }
#undef PCX_UPOLS
    (void)variant; (void)plain;
    set_error("fir partitioned ols: no kernel for %d partitions", parts);
    return PCX_ERR_UNSUPPORTED;
}

}  // namespace pcx
