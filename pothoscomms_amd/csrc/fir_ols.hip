// fir_ols.hip -- frequency-domain overlap-save /comms/fir_filter for complex_float32,
// M = L = 1: the same y[n] = sum_k h[k] * x[n-k] as FIRFilter.cpp:294-300, evaluated per
// 4096-sample block as  IFFT( FFT(block) .* H ),  H = FFT(h zero-padded)/4096.
//
// Why: direct form costs 8K flop per 16 algorithmic bytes (K = 255: 127 flop/B) and is
// pinned at the FP32-FMA roof (~77 Gsamples/s); overlap-save costs ~134 flop per sample
// whatever K (<= 2049), which puts the filter back under the HBM roof.
//
// One persistent workgroup (256 lanes, 4 resident per CU) walks blocks of S = 4096-Kov outputs,
// Kov = K-1 rounded up to 16 samples (aligned rows, see the kernel):
//   load 4096 inputs (block b starts at xh[b*S - pad]; consecutive blocks overlap by Kov)
//   forward radix-16 x3 Stockham (fft4096.hpp)            -> lane j holds X[j + 256 k]
//   multiply by the lane's 16 bins of H (32 VGPRs, loaded once per workgroup)
//   inverse radix-16 x3: its pass 1 wants x[j + 256 r], exactly what the lane holds,
//   so the spectrum never leaves registers
//   store time samples i >= Kov (the first K-1 are circularly aliased) to y[b*S + i-Kov]
// HBM traffic per block: 32 KiB read + 8*S bytes written; LDS: one padded 34 KiB image.
// What bounds it: DESIGN.md 4.1 (profiles/r03/ols_lab3.md).
#include "fft4096.hpp"
#include "pcx_sched.hpp"
#include <cstdio>
#include <cstdlib>

#include <type_traits>

#include "pcx_internal.hpp"

namespace pcx {

// One launch covers the whole call.  Full blocks take the fast load path; the ragged last
// block (fewer than 4096 inputs left / fewer than S outputs wanted) takes the range-checked
// load path -- a wave-uniform choice per block.  Stores always go through the descriptor's
// range check, which drops both the K-1 aliased samples and anything past n_out.
// PREFETCH: keep the next block's 16 samples per lane in flight in registers (168 VGPRs, 3
// workgroups per CU) or load at the top of each block (128 VGPRs, 4 workgroups per CU).
// LAUX / SAUX: cache-policy bits of the stream loads / stores (LAUX >= 4: per-row policy, see fetch).
// Measured at K = 255, one box, 64 Mi samples: plain 0.2269-0.2280 ms; nt on every load 0.2311 with
// nt stores (consecutive blocks re-read the overlap, and those hits are lost when the first touch
// bypasses L2); nt stores alone 0.2267-0.2277; nt stores + nt loads on the rows that only this block
// reads 0.2204-0.2215 -- the default.
// DIAG (timing-only builds, wrong outputs): 1 = every block reads/writes block 0 (cache resident:
// the compute floor), 2 = no transforms (load, store: the memory floor), 3 = real stream, butterflies
// without the LDS exchanges, 4 = real stream, LDS exchanges without the butterflies (energy split)
// HGLOBAL: fetch the lane's 16 H bins from L2 in every block instead of holding them in 32 VGPRs
// -- room for the register prefetch at 4 workgroups per CU.
// XCH: 1 = the transform pair with its second exchange inside sixteen lanes (fft4096.hpp: three barriers per block), 0 = the
// Stockham passes both ways (eight)
template <bool PREFETCH, int LAUX = 0, int SAUX = 0, int CHUNKED = 0, int DIAG = 0, bool HGLOBAL = false, bool GATED = false, int XCH = 0>
__global__ __launch_bounds__(256, (PREFETCH && !HGLOBAL) ? 3 : 4) void fir_cf32_ols4096_kernel(const float2 *__restrict__ in, size_t in_elems,
                                                                  float2 *__restrict__ out, size_t n_out,
                                                                  const float2 *__restrict__ Hspec, int Kov, int pad,
                                                                  const float2 *__restrict__ twtab, size_t first_full,
                                                                  size_t nfull, size_t nblocks, SchedState *__restrict__ sched, Gate gate)
{
    // Kov >= K-1 outputs are dropped at the head of every block and the block's input window starts
    // `pad` = Kov-(K-1) samples before sample b*S: with Kov a multiple of 16 every row this kernel
    // STORES starts on a 128-byte line (a K-1 = 254 overlap leaves each 512-byte row straddling
    // lines: measured 11 % slower on the load+store floor).
    using namespace fft4k;
    __shared__ cf lds[LDS_ELEMS];
    const int j = threadIdx.x;
    const size_t S = (size_t)(N - Kov);
    // CHUNKED: each persistent workgroup walks a CONTIGUOUS run of blocks (balanced partition)
    // instead of a grid stride, so the K-1 samples block b+1 shares with block b were fetched
    // by the same CU a moment ago (L2/L1 hit instead of a second trip to the memory side).
    size_t b, bend, bstep;
    // CHUNKED == 3 (the product default): blocks dealt dynamically, one per draw from sixteen counters (pcx_sched.hpp) -- the CUs
    // do not all run at one rate, and a fixed share per workgroup made every launch wait for the slowest
    __shared__ unsigned sched_slot;
    BlockDealer deal;
    if (CHUNKED == 3) {
        static_assert(CHUNKED != 3 || !PREFETCH, "the register prefetch needs the next block's index a block early");
        if (!deal.begin(sched, &sched_slot, nblocks, j)) { deal.finish(j); return; }
        b = GATED ? nblocks - 1 - deal.block() : deal.block();     // GATED (a shard behind a halo): the front blocks last
        bend = nblocks; bstep = 0;
    } else if (CHUNKED == 2) {
        // XCD-aware walk: workgroup w runs on XCD w % 8 (round-robin dispatch).  Each XCD takes one
        // contiguous eighth of the blocks and its workgroups walk it side by side, so the window rows
        // block b+1 shares with block b are fetched by the same XCD at about the same time (an L2
        // hit instead of a second trip over the fabric: the per-XCD L2s do not see each other)
        const size_t per = (nblocks + 7) / 8, x = blockIdx.x & 7;
        b = x * per + (blockIdx.x >> 3);
        bend = (x + 1) * per < nblocks ? (x + 1) * per : nblocks;
        bstep = gridDim.x >> 3;
    } else if (CHUNKED) {
        const size_t q = nblocks / gridDim.x, rem = nblocks % gridDim.x, w = blockIdx.x;
        b = w * q + (w < rem ? w : rem);
        bend = b + q + (w < rem ? 1 : 0);
        bstep = 1;
    } else {
        b = blockIdx.x; bend = nblocks; bstep = gridDim.x;
    }
    if (b >= bend) return;
    // loop invariants of the persistent workgroup: the lane's pass-3 twiddles and its 16 bins
    // of H in registers, the pass-2 twiddle table in LDS.  Nothing but the stream itself is
    // loaded from global memory inside the loop.
    auto fetch = [&](cf (&dst)[16], size_t blk) {
        if (DIAG == 1) blk = first_full;
        if (blk >= first_full && blk < nfull) {
            if (LAUX >= 4) {
                // NOV = 1 << (LAUX-4) rows at either end of the window hold the (aligned) overlap with
                // the neighbouring blocks: cached normally so the second reader hits in L2.  The rows
                // between are touched by this block only: non-temporal.
                constexpr int NOV = 1 << (LAUX >= 4 ? LAUX - 4 : 0);
                const __amdgpu_buffer_rsrc_t rs = make_rsrc(in + blk * S - pad, N * 8);
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const u32x2 t = (r < NOV || r >= 16 - NOV) ? __builtin_amdgcn_raw_buffer_load_b64(rs, j * 8, 2048 * r, 0)
                                                               : __builtin_amdgcn_raw_buffer_load_b64(rs, j * 8, 2048 * r, 2);
                    dst[r] = cf{__uint_as_float(t.x), __uint_as_float(t.y)};
                }
            } else {
                load_frame<false, LAUX>(dst, make_rsrc(in + blk * S - pad, N * 8), j);
            }
        } else {
            // ragged: block 0 when pad > 0 (its window would start before the buffer: those samples
            // only feed dropped outputs, read as 0 through the range check) and the tail block
            const size_t shift = blk * S >= (size_t)pad ? 0 : (size_t)pad - blk * S;
            const size_t first = blk * S + shift - pad;
            const size_t left = in_elems > first ? in_elems - first : 0;
            const size_t want = (size_t)N - shift;
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(in + first, (unsigned)((left < want ? left : want) * 8));
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, (j + 256 * r - (int)shift) * 8, 0, LAUX >= 4 ? 0 : LAUX);
                dst[r] = cf{__uint_as_float(t.x), __uint_as_float(t.y)};
            }
        }
    };
    // XCH: the first block's samples are requested together with the workgroup's 64 KB of tables, not behind them (possible since the
    // first block is the blockIdx).  Oldest request first: H and the pass-2 table (they pass through LDS, so they are waited for
    // first), the samples, the pass-3 twiddles -- every one of them through a descriptor, so that no address registers live across
    // the prologue, and with the block loop rotated (the next block's fetch at the foot of this one).  Built the plain way -- the
    // fetch at the top of the loop, selected on the first trip -- the same idea spilled 27-29 registers, thirteen of them re-read
    // from scratch in every block.
    cf v[16];
    LaneTw tw3;
    cf H[16];
    if (XCH) {
        SpectrumLoad sl;
        spectrum_request(sl, Hspec, twtab, j);
        if (!PREFETCH) {
            if (CHUNKED == 3 && GATED && b < gate.blocks) gate_wait(gate, j);
            fetch(v, b);
        }
        load_pass3_twiddles(tw3, make_rsrc(twtab, TW_TABLE_ELEMS * 8), j);
        spectrum_turn(H, sl, lds, j);
    } else {
        load_pass3_twiddles(tw3, twtab, j);
        stage_pass2_twiddles(lds, twtab, j);
    }
    if (!HGLOBAL && !XCH) {
#pragma unroll
        for (int k = 0; k < 16; k++) H[k] = reinterpret_cast<const cf *>(Hspec)[j + 256 * k];
    }
    static_assert(!XCH || (DIAG == 0 && !HGLOBAL), "the sixteen-lane exchange is built for the product configuration only");
    cf nx[16];
    if (PREFETCH) fetch(nx, b);
    for (; b < bend; b += bstep) {
        if (PREFETCH) {
#pragma unroll
            for (int r = 0; r < 16; r++) v[r] = nx[r];
            const size_t bn = b + bstep;
            if (bn < bend) fetch(nx, bn);   // in flight during this block's math
        } else if (XCH) {
            // (requested at the foot of the previous trip -- or ahead of the tables)
        } else {
            if (CHUNKED == 3 && GATED && b < gate.blocks) gate_wait(gate, j);   // this block's window reaches into the halo slot
            fetch(v, b);
        }
        constexpr int PART = DIAG == 3 ? 1 : DIAG == 4 ? 2 : 0;
        if (XCH) dif_a_math(v, tw3);
        else if (DIAG != 2) pass1_math<PART>(v);
        if (CHUNKED == 3) deal.draw(j);     // behind the first butterflies: no load of this block is outstanding any more (pcx_sched.hpp)
        if (XCH) dif_rest(v, lds, j);
        else if (DIAG != 2) {
        pass1_exchange<PART>(v, lds, j);
        pass2<PART>(v, lds, j);
        pass3<PART>(v, lds, j, tw3);
        }
        // spectrum times H, re-ordered into natural register order for the next pass 1.
        // The inverse transform runs on the FORWARD passes: IFFT(z) = conj(FFT(conj(z))), so
        // one set of twiddles serves both directions.  u = conj(v * H); the final conj rides
        // on the last additions of the inverse's third pass.
        cf u[16];
        if (HGLOBAL) {
            const cf *Hp = reinterpret_cast<const cf *>(Hspec);
            asm volatile("" : "+s"(Hp));   // loop-invariant loads must stay in the loop (L2 hits, not registers)
#pragma unroll
            for (int k = 0; k < 16; k++) H[k] = Hp[j + 256 * k];
        }
#pragma unroll
        for (int q = 0; q < 16; q += 2) {
            const int k0 = bin_of(q), k1 = bin_of(q + 1);
            u[k0] = v[q];
            u[k1] = v[q + 1];
            cmul2_conj(u[k0], u[k1], H[k0], H[k1]);
        }
        if (CHUNKED == 3) deal.publish(j);  // the inverse passes' barriers stand between this and deal.advance()
        // second half of a block ahead of other workgroups' first halves (priority, then age, decides VALU issue between the four
        // waves of a SIMD): blocks already half done retire -- and free their loads' successors -- sooner.  Measured +1.7 %
        // (tools/ols_lab.hip "prio 1 on inverse + stores": 0.2060 -> 0.2023 ms); graded or higher levels measured the same.
        __builtin_amdgcn_s_setprio(1);
        if (XCH) dit_back(u, lds, j, tw3);
        else if (DIAG != 2) {
        pass1<PART>(u, lds, j);
        pass2<PART>(u, lds, j);
        pass3<PART, true>(u, lds, j, tw3);      // conj(FFT(u)) = the block's time samples: the conjugation rides on the last additions
        }
        // time sample i of the block is output b*S + i - (K-1).  For i < K-1 (circularly
        // aliased) the unsigned byte offset wraps far beyond num_records and the store is
        // dropped by the range check, as are outputs past n_out in the last block.
        const size_t bo = DIAG == 1 ? 0 : b;
        const size_t room = n_out - bo * S;
        const __amdgpu_buffer_rsrc_t ws = make_rsrc(out + bo * S, (unsigned)((room < S ? room : S) * 8));
        const unsigned vbase = (unsigned)(j - Kov) * 8u;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int row = 256 * bin_of(q);
            if (row + 255 < Kov) continue;                    // whole row dropped: uniform skip
            store_cf<SAUX>(ws, vbase + (unsigned)row * 8u, DIAG == 2 ? cf{u[q].x, -u[q].y} : u[q]);
        }
        __builtin_amdgcn_s_setprio(0);
        if (CHUNKED == 3) {
            if (!deal.advance()) break;
            b = GATED ? nblocks - 1 - deal.block() : deal.block();
        }
        if (XCH && !PREFETCH) {
            const size_t bn = CHUNKED == 3 ? b : b + bstep;
            if (bn < bend) {
                if (CHUNKED == 3 && GATED && bn < gate.blocks) gate_wait(gate, j);   // this block's window reaches into the halo slot
                fetch(v, bn);
            }
        }
    }
    if (CHUNKED == 3) deal.finish(j);
}

// gate_word != nullptr: the caller wants the blocks that read in[0 .. K-2] held until *gate_word reaches gate_value.  Only the
// dealt kernel has the gate; *gated says whether this launch honours it -- if not, NOTHING has been launched and the caller
// orders the halo in front of the call itself.
// slots: resident workgroups this launch may take (1024 = the whole device; a device that carries several shards of a stream
// gives each its share, pcx_shard.hip) -- a multiple of 128, so that the dealer's sixteen groups hold the same number of workgroups
// lead_valid: samples of the SAME stream readable in front of `in` (a call that is one chunk of a longer one, pcx_fir_process's
// drained output): with at least `pad` of them block 0 is a full block like any other and every output comes out exactly as
// the uncut call computes it
int launch_fir_cf32_ols4096(const void *in, size_t in_elems, void *out, size_t n_out, const void *Hspec, size_t K,
                            const void *tw4096, void *sched, hipStream_t st, const void *gate_word, unsigned gate_value, int *gated,
                            unsigned slots, size_t lead_valid)
{
    if (gated) *gated = 0;
    if (n_out == 0) return PCX_OK;
    if (K < 1 || K > 2049) { set_error("fir ols: K=%zu outside 1..2049", K); return PCX_ERR_UNSUPPORTED; }
    // PCX_OLS_VARIANT (libpcx_hip_diag.so only: the product library has no such switch): unset = default policy below;
    // 1 plain loads/stores, 0 register prefetch (3 workgroups/CU), 2/3 nt loads / nt loads+stores, 4 contiguous block runs,
    // 5 H from L2, 6 nt stores only, 7 nt interior loads only, 9 register prefetch + default cache policy, 14 XCD-aware block
    // walk; TIMING-ONLY (wrong outputs): 10/11 compute-only / memory-only (plain accesses), 12/13 the same with the default
    // policy, 15/16 butterflies-only / exchanges-only on the real stream.  PCX_OLS_ALIGN=0 keeps the minimal K-1 overlap.
    const int variant = (int)PCX_ENV_INT("PCX_OLS_VARIANT", -1);
    const int align = (int)PCX_ENV_INT("PCX_OLS_ALIGN", 1);
#ifdef PCX_DIAG
    if (variant == 10 || variant == 11 || variant == 12 || variant == 13 || variant == 15 || variant == 16) {
        static bool warned = false;
        if (!warned) fprintf(stderr, "pcx(diag): PCX_OLS_VARIANT=%d selects a TIMING-ONLY build of the overlap-save FIR: its outputs are wrong\n", variant);
        warned = true;
    }
#endif
    const size_t Km1 = K - 1;
    const size_t Kov = align ? (Km1 + 15) / 16 * 16 : Km1;   // <= 2048
    const size_t pad = Kov - Km1;
    const size_t S = 4096 - Kov;
    const size_t nblocks = (n_out + S - 1) / S;
    // full blocks: window inside the buffer and all S outputs wanted
    const size_t first_full = pad > lead_valid ? 1 : 0;
    size_t nfull = n_out / S;
    while (nfull > first_full && (nfull - 1) * S - pad + 4096 > in_elems) nfull--;
    if (nfull < first_full) nfull = first_full;
    const float2 *pi = (const float2 *)in, *ph = (const float2 *)Hspec, *pt = (const float2 *)tw4096;
    float2 *po = (float2 *)out;
    // slots below 128: a call on HOST memory over PCIe (pcx_api.hip host_grid): that many workgroups on the grid stride, several blocks each,
    // no dealer -- block k+1's fetch at the foot of block k then runs beside block k's stores, and the link carries both directions at once
    const bool host_grid = slots > 0 && slots < 128;
    // PCX_OLS_SLOTS (diagnostic): resident workgroups to use chip-wide (default 1024 = 4 per CU)
    if (!host_grid && (slots < 128 || slots > 1024 || slots % 128)) slots = 1024;
    const long env_static = PCX_ENV_INT("PCX_OLS_SLOTS", 0);
    const unsigned static_slots = env_static > 0 ? (unsigned)env_static : slots;
    const unsigned g4 = persistent_grid(nblocks, static_slots), g3 = persistent_grid(nblocks, 768);
    const unsigned gx = 8 * persistent_grid((nblocks + 7) / 8, 128);   // XCD-aware walk: equal rounds inside every XCD's eighth
    const bool dealt = !host_grid && sched && nblocks > 2 * (size_t)slots && !PCX_ENV_SET("PCX_SCHED_STATIC");   // (PCX_SCHED_STATIC, diag only: the grid stride, for A/B)
    if (gate_word && !dealt) return PCX_OK;       // no gate in the grid-stride kernels: *gated stays 0, nothing launched
    // the window of block b starts at sample b*S - pad: only block 0 reaches below K-1 (Kov <= 2048 <= S)
    const Gate gate{dealt ? (const unsigned *)gate_word : nullptr, gate_value, 1u};
    if (gated && gate.word) *gated = 1;
#define PCX_OLS_LAUNCH(KERN, GRID) hipLaunchKernelGGL(KERN, dim3(GRID), dim3(256), 0, st, pi, in_elems, po, n_out, ph, (int)Kov, (int)pad, pt, first_full, nfull, nblocks, (SchedState *)sched, gate)
#ifdef PCX_DIAG
    switch (variant) {
    case 0: PCX_OLS_LAUNCH((fir_cf32_ols4096_kernel<true>), g3); goto launched;
    case 1: PCX_OLS_LAUNCH((fir_cf32_ols4096_kernel<false>), g4); goto launched;
    case 2: PCX_OLS_LAUNCH((fir_cf32_ols4096_kernel<false, 2, 0>), g4); goto launched;
    case 3: PCX_OLS_LAUNCH((fir_cf32_ols4096_kernel<false, 2, 2>), g4); goto launched;
    case 4: PCX_OLS_LAUNCH((fir_cf32_ols4096_kernel<false, 0, 0, 1>), g4); goto launched;
    case 5: PCX_OLS_LAUNCH((fir_cf32_ols4096_kernel<true, 0, 0, 0, 0, true>), g4); goto launched;
    case 6: PCX_OLS_LAUNCH((fir_cf32_ols4096_kernel<false, 0, 2>), g4); goto launched;
    case 7: PCX_OLS_LAUNCH((fir_cf32_ols4096_kernel<false, 4, 0>), g4); goto launched;
    case 9: PCX_OLS_LAUNCH((fir_cf32_ols4096_kernel<true, 4, 2>), g3); goto launched;
    case 14: PCX_OLS_LAUNCH((fir_cf32_ols4096_kernel<false, 4, 2, 2>), gx); goto launched;
    // 20 / 21: static stride on a grid of PCX_ROUNDS blocks per workgroup (default 3), H held in registers / fetched from L2 per block
    case 20: PCX_OLS_LAUNCH((fir_cf32_ols4096_kernel<false, 4, 2>), rounds_grid(nblocks, 1024, 3)); goto launched;
    case 21: PCX_OLS_LAUNCH((fir_cf32_ols4096_kernel<false, 4, 2, 0, 0, true>), rounds_grid(nblocks, 1024, 3)); goto launched;
    case 15: PCX_OLS_LAUNCH((fir_cf32_ols4096_kernel<false, 4, 2, 0, 3>), g4); goto launched;
    case 16: PCX_OLS_LAUNCH((fir_cf32_ols4096_kernel<false, 4, 2, 0, 4>), g4); goto launched;
    case 12: PCX_OLS_LAUNCH((fir_cf32_ols4096_kernel<false, 4, 2, 0, 1>), g4); goto launched;
    case 13: PCX_OLS_LAUNCH((fir_cf32_ols4096_kernel<false, 4, 2, 0, 2>), g4); goto launched;
    // 30..34: the product kernel with other cache-policy bits on its stores (sc0 = 1, nt = 2, sc1 = 16)
    case 30: PCX_OLS_LAUNCH((fir_cf32_ols4096_kernel<false, 4, 0, 3, 0, false, false, 1>), g4); goto launched;
    case 31: PCX_OLS_LAUNCH((fir_cf32_ols4096_kernel<false, 4, 3, 3, 0, false, false, 1>), g4); goto launched;
    case 32: PCX_OLS_LAUNCH((fir_cf32_ols4096_kernel<false, 4, 18, 3, 0, false, false, 1>), g4); goto launched;
    case 33: PCX_OLS_LAUNCH((fir_cf32_ols4096_kernel<false, 4, 19, 3, 0, false, false, 1>), g4); goto launched;
    case 34: PCX_OLS_LAUNCH((fir_cf32_ols4096_kernel<false, 4, 16, 3, 0, false, false, 1>), g4); goto launched;
    case 35: PCX_OLS_LAUNCH((fir_cf32_ols4096_kernel<false, 4, 17, 3, 0, false, false, 1>), g4); goto launched;
    case 10: PCX_OLS_LAUNCH((fir_cf32_ols4096_kernel<false, 0, 0, 0, 1>), g4); goto launched;
    case 11: PCX_OLS_LAUNCH((fir_cf32_ols4096_kernel<false, 0, 0, 0, 2>), g4); goto launched;
    default: break;
    }
#else
    (void)variant; (void)g3; (void)gx;
#endif
    // non-temporal stores; non-temporal loads for the rows no other block reads; blocks dealt dynamically when the caller
    // brought the counter pair (every pcx_fir handle does), else the grid stride
    if (dealt && gate.word) {
        // (Leaving a few of the 1024 resident slots empty, so that the kernels the gate waits for -- the halo's receive or copy, the
        // one-thread signal -- find room beside this launch, measured no better: 0.2196 ms at 1024 workgroups, 0.2242 at 1008,
        // two shards on one device, profiles/r03/shard_probe.txt.  PCX_GATED_SLOTS (diag) sets the number.)
        const long env_gd = PCX_ENV_INT("PCX_GATED_SLOTS", 0);
        const unsigned gd = env_gd > 0 ? (unsigned)env_gd : slots;
        if (Kov <= 256) PCX_OLS_LAUNCH((fir_cf32_ols4096_kernel<false, 4, 2, 3, 0, false, true, 1>), gd);
        else if (Kov <= 512) PCX_OLS_LAUNCH((fir_cf32_ols4096_kernel<false, 5, 2, 3, 0, false, true, 1>), gd);
        else if (Kov <= 1024) PCX_OLS_LAUNCH((fir_cf32_ols4096_kernel<false, 6, 2, 3, 0, false, true, 1>), gd);
        else PCX_OLS_LAUNCH((fir_cf32_ols4096_kernel<false, 0, 2, 3, 0, false, true, 1>), gd);
    } else if (dealt) {
        const long env_gd = PCX_ENV_INT("PCX_DEALT_SLOTS", 0);
        const unsigned gd = env_gd > 0 ? (unsigned)env_gd : slots;   // 4 resident workgroups per CU, all of them drawing
#ifdef PCX_DIAG
        if (PCX_ENV_INT("PCX_OLS_XCH", 1) == 0) {       // the eight-barrier Stockham pair, for A/B
            if (Kov <= 256) PCX_OLS_LAUNCH((fir_cf32_ols4096_kernel<false, 4, 2, 3>), gd);
            else PCX_OLS_LAUNCH((fir_cf32_ols4096_kernel<false, 0, 2, 3>), gd);
        } else
#endif
        if (Kov <= 256) PCX_OLS_LAUNCH((fir_cf32_ols4096_kernel<false, 4, 2, 3, 0, false, false, 1>), gd);
        else if (Kov <= 512) PCX_OLS_LAUNCH((fir_cf32_ols4096_kernel<false, 5, 2, 3, 0, false, false, 1>), gd);
        else if (Kov <= 1024) PCX_OLS_LAUNCH((fir_cf32_ols4096_kernel<false, 6, 2, 3, 0, false, false, 1>), gd);
        else PCX_OLS_LAUNCH((fir_cf32_ols4096_kernel<false, 0, 2, 3, 0, false, false, 1>), gd);
    }
    else if (Kov <= 256) PCX_OLS_LAUNCH((fir_cf32_ols4096_kernel<false, 4, 2, 0, 0, false, false, 1>), g4);
    else if (Kov <= 512) PCX_OLS_LAUNCH((fir_cf32_ols4096_kernel<false, 5, 2, 0, 0, false, false, 1>), g4);
    else if (Kov <= 1024) PCX_OLS_LAUNCH((fir_cf32_ols4096_kernel<false, 6, 2, 0, 0, false, false, 1>), g4);
    else PCX_OLS_LAUNCH((fir_cf32_ols4096_kernel<false, 0, 2, 0, 0, false, false, 1>), g4);
#ifdef PCX_DIAG
launched:
#endif
#undef PCX_OLS_LAUNCH
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}

// --------------------------------------------------------------------------------- //
// REAL float32 streams with REAL taps (FIRFilterFactory's first row, FIRFilter.cpp:374): two
// overlap-save blocks share one complex transform.  With real taps the filter acts on the
// real and imaginary parts independently, so z = xA + i*xB (xA, xB = blocks 2p and 2p+1 of the
// real stream) gives Re(h*z) = h*xA and Im(h*z) = h*xB: same pipeline, half the transforms
// per sample.  Loads/stores are 4 bytes per lane (two rows per complex element).
// --------------------------------------------------------------------------------- //
template <int NOV, bool DYN = false>
__global__ __launch_bounds__(256, 4) void fir_f32_ols4096_kernel(const float *__restrict__ in, size_t in_elems,
                                                                 float *__restrict__ out, size_t n_out,
                                                                 const float2 *__restrict__ Hspec, int Kov, int pad,
                                                                 const float2 *__restrict__ twtab, size_t nblocks, SchedState *__restrict__ sched)
{
    // Kov = K-1 rounded up to a multiple of 32 samples (128-byte aligned 1 KiB output rows), the
    // window of block b starts `pad` = Kov-(K-1) samples before sample b*S (fir_cf32_ols4096_kernel)
    using namespace fft4k;
    __shared__ cf lds[LDS_ELEMS];
    const int j = threadIdx.x;
    const size_t S = (size_t)(N - Kov);
    const size_t npairs = (nblocks + 1) / 2;
    __shared__ unsigned sched_slot;
    BlockWalk<DYN> walk;     // the unit dealt is a PAIR of real blocks (one complex transform)
    if (!walk.begin(sched, &sched_slot, npairs, j)) { walk.finish(j); return; }
    LaneTw tw3;
    load_pass3_twiddles(tw3, make_rsrc(twtab, TW_TABLE_ELEMS * 8), j);
    cf H[16];
    load_spectrum_lanes(H, Hspec, twtab, lds, j);
    for (;;) {
        const size_t p = walk.block();
        const size_t bA = 2 * p, bB = 2 * p + 1;
        cf v[16];
        if (bA > 0 && bB * S - pad + N <= in_elems) {
            // both windows inside the buffer.  The NOV rows at either end of a window are shared with
            // the neighbouring blocks (cached normally); the rows between are read once (non-temporal)
            const __amdgpu_buffer_rsrc_t ra = make_rsrc(in + bA * S - pad, N * 4);
            const __amdgpu_buffer_rsrc_t rb = make_rsrc(in + bB * S - pad, N * 4);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                unsigned a, b;
                if (r < NOV || r >= 16 - NOV) {
                    a = __builtin_amdgcn_raw_buffer_load_b32(ra, j * 4, 1024 * r, 0);
                    b = __builtin_amdgcn_raw_buffer_load_b32(rb, j * 4, 1024 * r, 0);
                } else {
                    a = __builtin_amdgcn_raw_buffer_load_b32(ra, j * 4, 1024 * r, 2);
                    b = __builtin_amdgcn_raw_buffer_load_b32(rb, j * 4, 1024 * r, 2);
                }
                v[r] = cf{__uint_as_float(a), __uint_as_float(b)};
            }
        } else {
            // ragged: block 0 (its window starts `pad` samples before the buffer: read as 0 through the
            // range check, they only feed dropped outputs), the tail, a missing partner block
            const size_t shiftA = bA * S >= (size_t)pad ? 0 : (size_t)pad - bA * S;
            const size_t firstA = bA * S + shiftA - pad;
            const size_t leftA = in_elems > firstA ? in_elems - firstA : 0;
            const size_t wantA = (size_t)N - shiftA;
            const bool haveB = bB < nblocks;
            const size_t firstB = haveB ? bB * S - pad : 0;          // bB >= 1: S >= pad
            const size_t leftB = haveB && in_elems > firstB ? in_elems - firstB : 0;
            const __amdgpu_buffer_rsrc_t ra = make_rsrc(in + firstA, (unsigned)((leftA < wantA ? leftA : wantA) * 4));
            const __amdgpu_buffer_rsrc_t rb = make_rsrc(in + firstB, (unsigned)((leftB < (size_t)N ? leftB : (size_t)N) * 4));
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const unsigned a = __builtin_amdgcn_raw_buffer_load_b32(ra, (j + 256 * r - (int)shiftA) * 4, 0, 0);
                const unsigned b = __builtin_amdgcn_raw_buffer_load_b32(rb, (j + 256 * r) * 4, 0, 0);
                v[r] = cf{__uint_as_float(a), __uint_as_float(b)};
            }
        }
        dif_a_math(v, tw3);
        walk.draw(j);                       // behind the first butterflies (pcx_sched.hpp)
        dif_rest(v, lds, j);
        cf u[16];
#pragma unroll
        for (int q = 0; q < 16; q += 2) {
            const int k0 = bin_of(q), k1 = bin_of(q + 1);
            u[k0] = v[q];
            u[k1] = v[q + 1];
            cmul2_conj(u[k0], u[k1], H[k0], H[k1]);
        }
        walk.publish(j);
        __builtin_amdgcn_s_setprio(1);      // as fir_cf32_ols4096_kernel: the second half of a block first
        dit_back(u, lds, j, tw3);           // conjugated on the last additions: u = the time samples
        // real part -> block A's outputs, imaginary part -> block B's
        const size_t roomA = n_out - bA * S;
        const size_t roomB = bB < nblocks ? n_out - bB * S : 0;
        const __amdgpu_buffer_rsrc_t wa = make_rsrc(out + bA * S, (unsigned)((roomA < S ? roomA : S) * 4));
        const __amdgpu_buffer_rsrc_t wb = make_rsrc(out + (bB < nblocks ? bB * S : 0), (unsigned)((roomB < S ? roomB : S) * 4));
        const unsigned vbase = (unsigned)(j - Kov) * 4u;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int row = 256 * bin_of(q);
            if (row + 255 < Kov) continue;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(u[q].x), wa, (int)(vbase + (unsigned)row * 4u), 0, 2);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(u[q].y), wb, (int)(vbase + (unsigned)row * 4u), 0, 2);
        }
        __builtin_amdgcn_s_setprio(0);
        if (!walk.advance()) break;
    }
    walk.finish(j);
}

int launch_fir_f32_ols4096(const void *in, size_t in_elems, void *out, size_t n_out, const void *Hspec, size_t K,
                           const void *tw4096, void *sched, hipStream_t st)
{
    if (n_out == 0) return PCX_OK;
    if (K < 1 || K > 2049) { set_error("fir ols (real): K=%zu outside 1..2049", K); return PCX_ERR_UNSUPPORTED; }
    const size_t Km1 = K - 1;
    const size_t Kov = (Km1 + 31) / 32 * 32, pad = Kov - Km1;     // <= 2048
    const size_t S = 4096 - Kov;
    const size_t nblocks = (n_out + S - 1) / S;
    const size_t npairs = (nblocks + 1) / 2;
    const bool dyn = sched && npairs > 2 * 1024 && !PCX_ENV_SET("PCX_SCHED_STATIC");
    const unsigned grid = dyn ? 1024u : persistent_grid(npairs, 1024);
#define PCX_REAL_LAUNCH(NOV)                                                                                                                  \
    do {                                                                                                                                      \
        if (dyn) hipLaunchKernelGGL((fir_f32_ols4096_kernel<NOV, true>), dim3(grid), dim3(256), 0, st, (const float *)in, in_elems, (float *)out, n_out, \
                                    (const float2 *)Hspec, (int)Kov, (int)pad, (const float2 *)tw4096, nblocks, (SchedState *)sched);            \
        else hipLaunchKernelGGL((fir_f32_ols4096_kernel<NOV, false>), dim3(grid), dim3(256), 0, st, (const float *)in, in_elems, (float *)out, n_out,    \
                                (const float2 *)Hspec, (int)Kov, (int)pad, (const float2 *)tw4096, nblocks, (SchedState *)nullptr);              \
    } while (0)
    if (Kov <= 256) PCX_REAL_LAUNCH(1);
    else if (Kov <= 512) PCX_REAL_LAUNCH(2);
    else if (Kov <= 1024) PCX_REAL_LAUNCH(4);
    else PCX_REAL_LAUNCH(8);
#undef PCX_REAL_LAUNCH
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}

// --------------------------------------------------------------------------------- //
// Rational resampling (interpolation L, decimation M) on the same block pipeline.
//
// FIRFilter.cpp:286-302 walks the flat index f = n*L + j (input n, polyphase row j) and emits
// an output whenever (f+1) % M == 0, computed with row j's taps h_j[k] = taps[j + k*L].  Here
// one launch handles one row j: it filters the input with H_j at full rate in the frequency
// domain and stores only the samples the decimator keeps, y_j[n] -> out[(f+1)/M - 1].
// Cost: L passes over the input whatever M is -- against the generic kernel's K MACs per
// output read from global memory.  The division by M is a multiply by a host-computed
// reciprocal (x < M + 4096*L < 2^23, M < 2^17: exact).
// --------------------------------------------------------------------------------- //
__global__ __launch_bounds__(256, 4) void fir_cf32_ols4096_poly_kernel(const float2 *__restrict__ in, size_t in_elems,
                                                                       float2 *__restrict__ out, size_t n_iter,
                                                                       const float2 *__restrict__ Hspec, int Km1,
                                                                       const float2 *__restrict__ twtab, size_t nblocks,
                                                                       unsigned L, unsigned M, unsigned jrow, unsigned long long magic)
{
    using namespace fft4k;
    __shared__ cf lds[LDS_ELEMS];
    const int j = threadIdx.x;
    const size_t S = (size_t)(N - Km1);
    size_t b = blockIdx.x;
    if (b >= nblocks) return;
    LaneTw tw3;
    load_pass3_twiddles(tw3, make_rsrc(twtab, TW_TABLE_ELEMS * 8), j);
    cf H[16];
    load_spectrum_lanes(H, Hspec, twtab, lds, j);
    for (; b < nblocks; b += gridDim.x) {
        cf v[16];
        {
            const size_t left = in_elems - b * S;
            if (left >= (size_t)N) load_frame<false, 0>(v, make_rsrc(in + b * S, N * 8), j);   // row offsets in soffset
            else load_frame<true, 0>(v, make_rsrc(in + b * S, (unsigned)(left * 8)), j);      // ragged tail: range-checked
        }
        dif_a_math(v, tw3);
        dif_rest(v, lds, j);
        cf u[16];
#pragma unroll
        for (int q = 0; q < 16; q += 2) {
            const int k0 = bin_of(q), k1 = bin_of(q + 1);
            u[k0] = v[q];
            u[k1] = v[q + 1];
            cmul2_conj(u[k0], u[k1], H[k0], H[k1]);
        }
        dit_back(u, lds, j, tw3);           // conjugated on the last additions: u = the time samples
        // input index of time sample i is n = b*S + i - (K-1); flat index f = n*L + jrow
        const unsigned long long B0 = (unsigned long long)(b * S) * L + jrow + 1;   // f+1 at i' = i-(K-1) = 0 (wave-uniform)
        const unsigned long long q0 = B0 / M;
        const unsigned r0 = (unsigned)(B0 - q0 * M);
        const size_t n_left = n_iter - b * S;                                       // inputs of this call from the block start
        const unsigned nl = n_left < (size_t)N ? (unsigned)n_left : (unsigned)N;   // 32-bit bound for the lane tests
        float2 *ob = out + q0 - 1;
        asm volatile("" : "+s"(ob));   // keep ONE base pointer live instead of 16 hoisted row addresses
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int row = 256 * bin_of(q);
            if (row + 255 < Km1) continue;
            const unsigned ip = (unsigned)(row + j - Km1);      // negative wraps above nl
            if (ip >= nl) continue;
            const unsigned x = r0 + ip * L;
            const unsigned qq = (unsigned)(((unsigned long long)x * magic) >> 40);   // x / M
            if (x - qq * M == 0) ob[qq] = make_float2(u[q].x, u[q].y);
        }
    }
}

int launch_fir_cf32_ols4096_poly(const void *in, size_t in_elems, void *out, size_t n_iter, const void *Hspec_rows, size_t K,
                                 size_t L, size_t M, const void *tw4096, hipStream_t st)
{
    if (n_iter == 0) return PCX_OK;
    if (K < 1 || K > 2049 || M >= (1u << 17) || 4096 * L + M >= (1u << 23)) {
        set_error("fir ols (polyphase): K=%zu L=%zu M=%zu outside the kernel's range", K, L, M);
        return PCX_ERR_UNSUPPORTED;
    }
    const size_t S = 4096 - (K - 1);
    const size_t nblocks = (n_iter + S - 1) / S;
    const unsigned grid = persistent_grid(nblocks, 1024);
    const unsigned long long magic = ((1ull << 40) + M - 1) / M;   // ceil(2^40 / M)
    for (size_t jrow = 0; jrow < L; jrow++) {
        hipLaunchKernelGGL(fir_cf32_ols4096_poly_kernel, dim3(grid), dim3(256), 0, st, (const float2 *)in, in_elems, (float2 *)out,
                           n_iter, (const float2 *)Hspec_rows + jrow * 4096, (int)(K - 1), (const float2 *)tw4096, nblocks,
                           (unsigned)L, (unsigned)M, (unsigned)jrow, magic);
        PCX_LAUNCH_CHECK();
    }
    return PCX_OK;
}

// Interpolation by a factor the replicated-spectrum kernel does not take (L = 3, 5, 6 ...; decimation 1): every polyphase
// row runs the undecimated kernel above into its own contiguous row of a workspace -- 2 KiB stores instead of the polyphase
// kernel's stride-L 8-byte ones -- and this pass interleaves them: out[n L + j] = rows[j][n].
// M > 1 (rational resampling): output t is position o = t M + M - 1 of the interleaved stream -- the one the reference's
// decimator keeps (FIRFilter.cpp:291)
template <typename E>
__global__ __launch_bounds__(256) void interleave_rows_kernel(const E *__restrict__ rows, E *__restrict__ out, size_t n, unsigned L, unsigned M)
{
    const size_t total = n * L / M, gstride = (size_t)gridDim.x * blockDim.x;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gstride) {
        const size_t o = t * M + (M - 1);
        const size_t ni = o / L;
        const unsigned jr = (unsigned)(o - ni * L);
        const E v = __builtin_nontemporal_load(rows + (size_t)jr * n + ni);
        __builtin_nontemporal_store(v, out + t);
    }
}
// elem_bytes: 16 (complex_float64), 8 (complex_float32, float64), 4 (complex_int16, float32), 2 (complex_int8, int16), 1 (int8)
int launch_interleave_rows(const void *rows, void *out, size_t n, size_t L, size_t elem_bytes, size_t M, hipStream_t st)
{
    if (n == 0) return PCX_OK;
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef double d2 __attribute__((ext_vector_type(2)));
    const dim3 grid(stream_grid(n * L / M, 256)), block(256);
    switch (elem_bytes) {
    case 8: hipLaunchKernelGGL(interleave_rows_kernel<f2>, grid, block, 0, st, (const f2 *)rows, (f2 *)out, n, (unsigned)L, (unsigned)M); break;
    case 16: hipLaunchKernelGGL(interleave_rows_kernel<d2>, grid, block, 0, st, (const d2 *)rows, (d2 *)out, n, (unsigned)L, (unsigned)M); break;
    case 4: hipLaunchKernelGGL(interleave_rows_kernel<unsigned>, grid, block, 0, st, (const unsigned *)rows, (unsigned *)out, n, (unsigned)L, (unsigned)M); break;
    case 1: hipLaunchKernelGGL(interleave_rows_kernel<unsigned char>, grid, block, 0, st, (const unsigned char *)rows, (unsigned char *)out, n, (unsigned)L, (unsigned)M); break;
    case 2: hipLaunchKernelGGL(interleave_rows_kernel<unsigned short>, grid, block, 0, st, (const unsigned short *)rows, (unsigned short *)out, n, (unsigned)L, (unsigned)M); break;
    default: set_error("interleave: element size %zu", elem_bytes); return PCX_ERR_ARG;
    }
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}
int launch_interleave_rows_cf32(const void *rows, void *out, size_t n, size_t L, hipStream_t st) { return launch_interleave_rows(rows, out, n, L, 8, 1, st); }

// --------------------------------------------------------------------------------- //
// Fused Rotate -> FIR -> FreqDemod in the frequency domain (BASELINE configs[4]).
//
// Same block pipeline as above with H' = phasor * H (Rotate folded into the spectrum: FIR
// is linear), then FreqDemod (demod/FreqDemod.cpp:60-67) on the time samples before anything
// is stored:  d[m] = arg(y[m] * conj(y[m-1])).  Blocks overlap by K (not K-1) input samples so
// that every y[m-1] a block needs is one of its own valid outputs: block b reads
// xh[b*S - 1 + i], i = 0..4095, S = 4096 - K; time index i >= K-1 is FIR output m = b*S + i - K
// and i >= K yields d[m].  y[-1] (block 0, i = K-1) is the state carried from the previous
// call (*prev_in = conj(y_last), zero after reset); conj(y[n_out-1]) goes to *prev_out.
// The lane's neighbour sample y[m-1] lives in lane j-1 of the same register: a wave-shift DPP move
// fetches it, and only the 64 wave-boundary values cross through LDS.  12 algorithmic bytes per
// sample (8 in, 4 out).  The overlap is rounded up to a multiple of 32 samples (aligned 1 KiB
// output rows); 128 VGPRs, 4 workgroups per CU.
// --------------------------------------------------------------------------------- //
template <int OCC, int NOV, int SAUX, bool DYN = false, bool GATED = false>
__global__ __launch_bounds__(256, OCC) void fmchain_cf32_ols4096_kernel(const float2 *__restrict__ in, size_t in_elems,
                                                                        float *__restrict__ out, size_t n_out,
                                                                        const float2 *__restrict__ Hspec, int K, int pad,
                                                                        const float2 *__restrict__ twtab, size_t nblocks,
                                                                        const float2 *__restrict__ prev_in,
                                                                        float2 *__restrict__ prev_out, SchedState *__restrict__ sched, Gate gate)
{
    // K here = the block overlap: taps + pad, a multiple of 32 so that every 1 KiB row of outputs
    // this kernel stores starts on a 128-byte line (see fir_cf32_ols4096_kernel)
    using namespace fft4k;
    __shared__ cf lds[LDS_ELEMS];
    __shared__ cf bnd[64];   // last lane of each wave, per row: the demodulator's cross-wave neighbours
    __shared__ unsigned sched_slot;
    const int j = threadIdx.x;
    const size_t S = (size_t)(N - K);
    size_t b = blockIdx.x;
    BlockDealer deal;        // DYN: blocks dealt dynamically (pcx_sched.hpp)
    if (DYN) {
        if (!deal.begin(sched, &sched_slot, nblocks, j)) { deal.finish(j); return; }
        b = GATED ? nblocks - 1 - deal.block() : deal.block();     // GATED (a shard behind a halo): the front blocks last
    } else if (b >= nblocks) return;
    LaneTw tw3;
    load_pass3_twiddles(tw3, make_rsrc(twtab, TW_TABLE_ELEMS * 8), j);
    cf H[16];
    load_spectrum_lanes(H, Hspec, twtab, lds, j);
    // element i of block blk is xh[blk*S - 1 - pad + i].  Block 0 has no xh[-1-pad .. -1]: its
    // descriptor sits at xh[0] and the byte offsets of those elements wrap past num_records and
    // read 0 (they only feed dropped outputs and y[-1], which is replaced by the carried state).
    auto fetch = [&](cf (&dst)[16], size_t blk) {
        const size_t lead = (size_t)(1 + pad);
        if (blk > 0 && blk * S - lead + N <= in_elems) {
            // as in fir_cf32_ols4096_kernel: the NOV rows at either end are shared with the neighbouring
            // blocks (cached normally), the rows between are read once (non-temporal)
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(in + blk * S - lead, N * 8);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                // row r at byte 2048 r: the odd rows' 2048 goes into the instruction's 12-bit offset field, so eight scalar
                // offsets serve sixteen rows
                const u32x2 t = (r < NOV || r >= 16 - NOV) ? __builtin_amdgcn_raw_buffer_load_b64(rs, j * 8 + 2048 * (r & 1), 4096 * (r >> 1), 0)
                                                           : __builtin_amdgcn_raw_buffer_load_b64(rs, j * 8 + 2048 * (r & 1), 4096 * (r >> 1), 2);
                dst[r] = cf{__uint_as_float(t.x), __uint_as_float(t.y)};
            }
            return;
        }
        const int shift = blk == 0 ? 1 + pad : 0;
        const size_t first = blk * S - (lead - (size_t)shift);         // xh index of the descriptor base
        const size_t left = in_elems - first;
        const size_t want = (size_t)(N - shift);
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(in + first, (unsigned)((left < want ? left : want) * 8));
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, (j + 256 * r - shift) * 8, 0, 0);
            dst[r] = cf{__uint_as_float(t.x), __uint_as_float(t.y)};
        }
    };
    for (; b < nblocks; b += DYN ? 0 : gridDim.x) {
        cf v[16];
        if (DYN && GATED && b < gate.blocks) gate_wait(gate, j);   // this block's window reaches into the halo slot
        fetch(v, b);
        dif_a_math(v, tw3);
        if (DYN) deal.draw(j);              // behind the first butterflies (pcx_sched.hpp)
        dif_rest(v, lds, j);
        cf u[16];
#pragma unroll
        for (int q = 0; q < 16; q += 2) {
            const int k0 = bin_of(q), k1 = bin_of(q + 1);
            u[k0] = v[q];
            u[k1] = v[q + 1];
            cmul2_conj(u[k0], u[k1], H[k0], H[k1]);
        }
        if (DYN) deal.publish(j);
        __builtin_amdgcn_s_setprio(1);      // as fir_cf32_ols4096_kernel: the second half of a block first
        dit_back<false>(u, lds, j, tw3);
        // u[q] = conj(y) at time index i = j + 256*bin_of(q).  conj(y[i-1]) sits in lane j-1 of the
        // same register: a wave-shift DPP move brings it over, and only the first lane of each wave
        // needs the last lane of the wave before it (row k-1 for lane 0) -- 64 values through LDS
        // instead of a fourth trip of the whole block image.
        if (b == 0) {
            // block 0: the slot of y[-1] (time index K-1, itself a dropped output) takes the carried state
            const cf carried = cf{prev_in[0].x, prev_in[0].y};
            int slot_i = K - 1;
            asm volatile("" : "+s"(slot_i));     // computed HERE, once per launch: hoisted, the sixteen lane masks below sat in 32 scalar registers for the whole loop
#pragma unroll
            for (int q = 0; q < 16; q++)
                if (j + 256 * bin_of(q) == slot_i) u[q] = carried;
        }
        if ((j & 63) == 63) {
#pragma unroll
            for (int q = 0; q < 16; q++) bnd[(j >> 6) * 16 + bin_of(q)] = u[q];
        }
        lds_barrier();                      // orders bnd[] only: no global access is waited for (fft4096.hpp)
        const size_t room = n_out - b * S;
        const size_t cnt = room < S ? room : S;
        const __amdgpu_buffer_rsrc_t ws = make_rsrc(out + b * S, (unsigned)(cnt * 4));
        // lane 0 of wave w > 0 continues lane 63 of wave w-1 in the same row; lane 0 of wave 0 continues
        // lane 255 of row k-1 (bnd[47 + k]; for k = 0 that is time index -1: never a valid output)
        const cf *edge_row = bnd + ((j >> 6) > 0 ? ((j >> 6) - 1) * 16 : 47);
        // Two rows at a time so that the demodulator's polynomial and reflections run on packed pairs.  Every row is computed whole
        // and the descriptor's range check drops what must not be stored (time index < K: the byte offset wraps past
        // num_records) -- no run-time row tests, no branches.  (Leaving out the stores of rows that are dead for every K of
        // an instantiation made the compiler spill 6-12 registers in the long-filter instantiations; a pair of rows is
        // skipped only when both are dead.)
        // The atan2 constants ride in four scalar register PAIRS, either half picked with op_sel: (c5,c4) (c3,c2) (c1,c0) (pi/2,pi).
        // atan(t) = t (c0 + c1 s + ... + c5 s^5), s = t^2, t in [0,1]: minimax, 1.8e-6 rad in float32 -- 18x inside the
        // 1e-5*pi parity bar (the seven-term fit it replaces: 3e-7 rad, 100x).
        constexpr int NDEAD = NOV / 2;      // Kov in (256 NOV/2, 256 NOV] (NOV = 1: Kov <= 256): rows 0 .. NDEAD-1 are dead whatever K
        const cf K54 = {-0.01171913556754589f, 0.05264735221862793f}, K32 = {-0.116426482796669f, 0.19354037940502167f};
        const cf K10 = {-0.33262282609939575f, 0.9999772310256958f}, KPI = {1.57079632679489661923f, 3.14159265358979323846f};
        // Where a row goes.  Rows k >= NOV are valid whole whatever K (Kov <= 256 NOV): lane offset 4 j -- one loop-invariant register --
        // and the row's displacement 1024 k - 4 K in the instruction's SCALAR offset.  The range check covers voffset + soffset as one
        // wide unsigned sum on gfx950 (tools/soffset_lab.hip: 1024 records, soffset 2048 -> nothing written; a wrapped voffset
        // stays out of range whatever soffset is added), so the stream's last, partial block is clipped exactly as before.  Rows
        // below NOV keep the wrapping lane offset (j - K) * 4 + 1024 k in a vector register, which is what drops their time indices
        // below K.  For the short-filter instantiation (NOV = 1) that is row 0 alone: fifteen v_add per block fewer.
        const unsigned j4 = (unsigned)j * 4u;
        int k4 = 4 * K;
        asm volatile("" : "+s"(k4));        // HERE, per block: hoisted out of the block loop the fifteen displacements sit in scalar registers the loop does not have (10-16 spilled)
#pragma unroll
        for (int q = 0; q < 16; q += 2) {
            const int k0 = bin_of(q), k1 = bin_of(q + 1);
            if (k0 < NDEAD && k1 < NDEAD) continue;
            // conj(y[m-1]) for both rows; edge_row[k]: wave-uniform address, broadcast read
            auto prev = [&](cf a, cf edge) {
                cf p;
                p.x = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge.x), __float_as_int(a.x), 0x138, 0xf, 0xf, false));
                p.y = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge.y), __float_as_int(a.y), 0x138, 0xf, 0xf, false));
                return p;
            };
            const cf a0 = u[q], a1 = u[q + 1];
            const cf p0 = prev(a0, edge_row[k0]), p1 = prev(a1, edge_row[k1]);
            // z = y[m] * conj(y[m-1]) = conj(a) * p = (a.x p.x + a.y p.y, a.x p.y - a.y p.x): two packed instructions per sample.
            // (Written as plain multiply-adds the vectoriser packs them itself and then spends eight v_mov per pair building
            // the operand pairs; this form costs two v_max x, x per sample instead -- the compiler cannot know asm results to be
            // canonical floats in front of the min / max below -- which measured the cheaper of the two.)
            cf z0, z1;
            asm("v_pk_mul_f32 %0, %2, %3 op_sel:[1,1] op_sel_hi:[1,0] neg_hi:[1,0]\n\t"     // (a.y p.y, -a.y p.x)
                "v_pk_mul_f32 %1, %4, %5 op_sel:[1,1] op_sel_hi:[1,0] neg_hi:[1,0]\n\t"
                "v_pk_fma_f32 %0, %2, %3, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"       // (a.x p.x, a.x p.y) + that
                "v_pk_fma_f32 %1, %4, %5, %1 op_sel:[0,0,0] op_sel_hi:[0,1,1]"
                : "=&v"(z0), "=&v"(z1)
                : "v"(a0), "v"(p0), "v"(a1), "v"(p1));
            const float x0 = z0.x, y0 = z0.y, x1 = z1.x, y1 = z1.y;
            // atan2(y, x), two samples side by side.  |z| = 0 (the first output after activate(): arg of a signed zero,
            // FreqDemod.cpp:44-47,63-64) gives t = 0 / tiny = 0 and the quadrant from the sign bits alone, as atan2f does.
            const cf mx = {__builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(x0), __builtin_fabsf(y0)), 1e-37f),
                           __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(x1), __builtin_fabsf(y1)), 1e-37f)};
            // min(|x|, |y|) spelt as the instruction it is: through __builtin_fminf the compiler puts a canonicalising v_max x, x in
            // front of either operand (it cannot know the asm results above to be quiet) -- four instructions per pair of samples,
            // 32 of the tail's ~337 per wave and block
            cf mn;
            asm("v_min_f32_e64 %0, |%1|, |%2|" : "=v"(mn.x) : "v"(x0), "v"(y0));
            asm("v_min_f32_e64 %0, |%1|, |%2|" : "=v"(mn.y) : "v"(x1), "v"(y1));
            const cf rc = {__builtin_amdgcn_rcpf(mx.x), __builtin_amdgcn_rcpf(mx.y)};
            const cf t = mn * rc;
            const cf sq = t * t;
            cf pl, ro, rn;
            asm("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[0,0,1] op_sel_hi:[0,1,1]" : "=v"(pl) : "s"(K54), "v"(sq));            // c5 s + c4
            asm("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(pl) : "v"(sq), "s"(K32));            // .. s + c3
            asm("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,0,1] op_sel_hi:[1,1,1]" : "+v"(pl) : "v"(sq), "s"(K32));            // .. s + c2
            asm("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(pl) : "v"(sq), "s"(K10));            // .. s + c1
            asm("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,0,1] op_sel_hi:[1,1,1]" : "+v"(pl) : "v"(sq), "s"(K10));            // .. s + c0
            cf r = t * pl;
            asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(ro) : "s"(KPI), "v"(r));   // pi/2 - r
            r = cf{__builtin_fabsf(y0) > __builtin_fabsf(x0) ? ro.x : r.x, __builtin_fabsf(y1) > __builtin_fabsf(x1) ? ro.y : r.y};
            asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(rn) : "s"(KPI), "v"(r));   // pi - r
            r = cf{__float_as_int(x0) < 0 ? rn.x : r.x, __float_as_int(x1) < 0 ? rn.y : r.y};
            const float d0 = __builtin_copysignf(r.x, y0), d1 = __builtin_copysignf(r.y, y1);
            auto put = [&](int k, float d) {
                if (k >= NOV) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(d), ws, (int)j4, 1024 * k - k4, SAUX);
                else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(d), ws, (int)(j4 + (unsigned)(1024 * k - k4)), 0, SAUX);   // = ((j - K) * 4 + 1024 k) mod 2^32
            };
            put(k0, d0);
            put(k1, d1);
        }
        if (b == nblocks - 1) {
            // the stream's last output becomes the next call's carried state (kept conjugated)
            const int i_last = K + (int)cnt - 1;
#pragma unroll
            for (int q = 0; q < 16; q++)
                if (j + 256 * bin_of(q) == i_last) prev_out[0] = make_float2(u[q].x, u[q].y);
        }
        __builtin_amdgcn_s_setprio(0);
        if (DYN) {
            if (!deal.advance()) break;
            b = GATED ? nblocks - 1 - deal.block() : deal.block();
        }
    }
    if (DYN) deal.finish(j);
}

// gate_word / gated: as launch_fir_cf32_ols4096 (the halo of a sharded chain is K samples: the FIR's K-1 and the demodulator's one)
int launch_fmchain_cf32_ols4096(const void *in, size_t in_elems, void *out, size_t n_out, const void *Hspec, size_t K,
                                const void *tw4096, const void *prev_in, void *prev_out, void *sched, hipStream_t st,
                                const void *gate_word, unsigned gate_value, int *gated, unsigned slots)
{
    if (gated) *gated = 0;
    if (n_out == 0) return PCX_OK;
    if (K < 1 || K > 2048) { set_error("fm chain ols: K=%zu outside 1..2048", K); return PCX_ERR_UNSUPPORTED; }
    // PCX_FMCHAIN_OCC (libpcx_hip_diag.so only, A/B): 3 = 3 workgroups per CU, 5 = 4 per CU with plain loads/stores; default 4 + row policy
    const int occ = (int)PCX_ENV_INT("PCX_FMCHAIN_OCC", 4);
    const size_t Kov = (K + 31) / 32 * 32, pad = Kov - K;    // <= 2048
    const size_t S = 4096 - Kov;
    const size_t nblocks = (n_out + S - 1) / S;
    const bool host_grid = slots > 0 && slots < 128;      // a call on host memory over PCIe: few workgroups, several blocks each (launch_fir_cf32_ols4096)
    if (!host_grid && (slots < 128 || slots > 1024 || slots % 128)) slots = 1024;
    const bool dyn = !host_grid && sched && nblocks > 2 * (size_t)slots && !PCX_ENV_SET("PCX_SCHED_STATIC");   // dynamic dealing when the handle brought its counter pair and the launch is long
    if (gate_word && !dyn) return PCX_OK;         // no gate in the grid-stride kernel: *gated stays 0, nothing launched
    // the window of block b starts at sample b*S - 1 - pad: it reaches below sample K while b*S < Kov + 1 -- block 0, and block 1
    // too when S == Kov (2048 taps)
    const Gate gate{dyn ? (const unsigned *)gate_word : nullptr, gate_value, S < Kov + 1 ? 2u : 1u};
    if (gated && gate.word) *gated = 1;
#define PCX_FM_LAUNCH(OCC, NOV, SAUX, CAP)                                                                                       \
    do {                                                                                                                         \
        if (dyn && gate.word)                                                                                                    \
            hipLaunchKernelGGL((fmchain_cf32_ols4096_kernel<OCC, NOV, SAUX, true, true>), dim3(CAP), dim3(256), 0, st, (const float2 *)in, \
                               in_elems, (float *)out, n_out, (const float2 *)Hspec, (int)Kov, (int)pad, (const float2 *)tw4096,  \
                               nblocks, (const float2 *)prev_in, (float2 *)prev_out, (SchedState *)sched, gate);                  \
        else if (dyn)                                                                                                            \
            hipLaunchKernelGGL((fmchain_cf32_ols4096_kernel<OCC, NOV, SAUX, true>), dim3(CAP), dim3(256), 0, st, (const float2 *)in, \
                               in_elems, (float *)out, n_out, (const float2 *)Hspec, (int)Kov, (int)pad, (const float2 *)tw4096,  \
                               nblocks, (const float2 *)prev_in, (float2 *)prev_out, (SchedState *)sched, gate);                  \
        else                                                                                                                     \
            hipLaunchKernelGGL((fmchain_cf32_ols4096_kernel<OCC, NOV, SAUX>), dim3(persistent_grid(nblocks, CAP)), dim3(256), 0, st, \
                               (const float2 *)in, in_elems, (float *)out, n_out, (const float2 *)Hspec, (int)Kov, (int)pad,      \
                               (const float2 *)tw4096, nblocks, (const float2 *)prev_in, (float2 *)prev_out, (SchedState *)nullptr, gate); \
    } while (0)
#ifdef PCX_DIAG
    if (occ == 3) PCX_FM_LAUNCH(3, 8, 0, 768);            // A/B: 3 workgroups per CU, plain loads and stores
    else if (occ == 5) PCX_FM_LAUNCH(4, 8, 0, 1024);      // A/B: plain loads and stores
    else
#else
    (void)occ;
#endif
    if (Kov <= 256) PCX_FM_LAUNCH(4, 1, 2, slots);
    else if (Kov <= 512) PCX_FM_LAUNCH(4, 2, 2, slots);
    else if (Kov <= 1024) PCX_FM_LAUNCH(4, 4, 2, slots);
    else PCX_FM_LAUNCH(4, 8, 2, slots);
#undef PCX_FM_LAUNCH
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}

}  // namespace pcx
