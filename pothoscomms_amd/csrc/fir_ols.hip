// fir_ols.hip -- frequency-domain overlap-save /comms/fir_filter for complex_float32,
// M = L = 1: the same y[n] = sum_k h[k] * x[n-k] as FIRFilter.cpp:294-300, evaluated per
// 4096-sample block as  IFFT( FFT(block) .* H ),  H = FFT(h zero-padded)/4096.
//
// Why: direct form costs 8K flop per 16 algorithmic bytes (K = 255: 127 flop/B) and is
// pinned at the FP32-FMA roof (~77 Gsamples/s); overlap-save costs ~134 flop per sample
// whatever K (<= 2049), which puts the filter back under the HBM roof.
//
// One workgroup (256 lanes) per block of S = 4096-(K-1) outputs:
//   load 4096 inputs (block b starts at xh[b*S]; consecutive blocks overlap by K-1)
//   forward radix-16 x3 Stockham (fft4096.hpp)            -> lane j holds X[j + 256 k]
//   multiply by H[j + 256 k] (coalesced, L2-resident)
//   inverse radix-16 x3: its pass 1 wants x[j + 256 r], exactly what the lane holds,
//   so the spectrum never leaves registers
//   store time samples i >= K-1 (the first K-1 are circularly aliased) to y[b*S + i-(K-1)]
// HBM traffic per block: 32 KiB read + 8*S bytes written; LDS: one padded 34 KiB image.
#include "fft4096.hpp"
#include "pcx_internal.hpp"

namespace pcx {

// One launch covers the whole call.  Full blocks take the fast load path; the ragged last
// block (fewer than 4096 inputs left / fewer than S outputs wanted) takes the range-checked
// load path -- a wave-uniform choice per block.  Stores always go through the descriptor's
// range check, which drops both the K-1 aliased samples and anything past n_out.
__global__ __launch_bounds__(256, 3) void fir_cf32_ols4096_kernel(const float2 *__restrict__ in, size_t in_elems,
                                                                  float2 *__restrict__ out, size_t n_out,
                                                                  const float2 *__restrict__ Hspec, int Km1,
                                                                  const float2 *__restrict__ twtab, size_t nfull, size_t nblocks)
{
    using namespace fft4k;
    __shared__ cf lds[LDS_ELEMS];
    const int j = threadIdx.x;
    const size_t S = (size_t)(N - Km1);
    size_t b = blockIdx.x;
    if (b >= nblocks) return;
    // loop invariants of the persistent workgroup: the lane's pass-3 twiddles and its 16 bins
    // of H in registers, the pass-2 twiddle table in LDS.  Nothing but the stream itself is
    // loaded from global memory inside the loop.
    LaneTw tw3;
    load_pass3_twiddles(tw3, twtab, j);
    stage_pass2_twiddles(lds, twtab, j);
    cf H[16];
#pragma unroll
    for (int k = 0; k < 16; k++) H[k] = reinterpret_cast<const cf *>(Hspec)[j + 256 * k];
    auto fetch = [&](cf (&dst)[16], size_t blk) {
        const size_t left = in_elems - blk * S;   // samples from the block start to the end of the buffer
        if (blk < nfull) load_frame<false>(dst, make_rsrc(in + blk * S, N * 8), j);
        else load_frame<true>(dst, make_rsrc(in + blk * S, (unsigned)((left < (size_t)N ? left : (size_t)N) * 8)), j);
    };
    cf nx[16];
    fetch(nx, b);
    for (; b < nblocks; b += gridDim.x) {
        cf v[16];
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] = nx[r];
        const size_t bn = b + gridDim.x;
        if (bn < nblocks) fetch(nx, bn);   // in flight during this block's math
        pass1(v, lds, j);
        pass2(v, lds, j);
        pass3(v, lds, j, tw3);
        // spectrum times H, re-ordered into natural register order for the next pass 1.
        // The inverse transform runs on the FORWARD passes: IFFT(z) = conj(FFT(conj(z))), so
        // one set of twiddles serves both directions.  u = conj(v * H); the final conj rides
        // on the store.
        cf u[16];
#pragma unroll
        for (int q = 0; q < 16; q += 2) {
            const int k0 = bin_of(q), k1 = bin_of(q + 1);
            u[k0] = v[q];
            u[k1] = v[q + 1];
            cmul2_conj(u[k0], u[k1], H[k0], H[k1]);
        }
        pass1(u, lds, j);
        pass2(u, lds, j);
        pass3(u, lds, j, tw3);
        // time sample i of the block is output b*S + i - (K-1).  For i < K-1 (circularly
        // aliased) the unsigned byte offset wraps far beyond num_records and the store is
        // dropped by the range check, as are outputs past n_out in the last block.
        const size_t room = n_out - b * S;
        const __amdgpu_buffer_rsrc_t ws = make_rsrc(out + b * S, (unsigned)((room < S ? room : S) * 8));
        const unsigned vbase = (unsigned)(j - Km1) * 8u;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int row = 256 * bin_of(q);
            if (row + 255 < Km1) continue;                    // whole row aliased: uniform skip
            store_cf(ws, vbase + (unsigned)row * 8u, cf{u[q].x, -u[q].y});
        }
    }
}

int launch_fir_cf32_ols4096(const void *in, size_t in_elems, void *out, size_t n_out, const void *Hspec, size_t K,
                            const void *tw4096, hipStream_t st)
{
    if (n_out == 0) return PCX_OK;
    if (K < 1 || K > 2049) { set_error("fir ols: K=%zu outside 1..2049", K); return PCX_ERR_UNSUPPORTED; }
    const size_t S = 4096 - (K - 1);
    // full blocks: b*S + 4096 <= in_elems and (b+1)*S <= n_out
    size_t nfull = n_out / S;
    while (nfull > 0 && (nfull - 1) * S + 4096 > in_elems) nfull--;
    const size_t nblocks = (n_out + S - 1) / S;
    const unsigned grid = (unsigned)(nblocks < 768 ? nblocks : 768);   // 3 persistent workgroups per CU
    hipLaunchKernelGGL(fir_cf32_ols4096_kernel, dim3(grid), dim3(256), 0, st, (const float2 *)in, in_elems, (float2 *)out,
                       n_out, (const float2 *)Hspec, (int)(K - 1), (const float2 *)tw4096, nfull, nblocks);
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}

}  // namespace pcx
