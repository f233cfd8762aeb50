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

// TAIL=false: every block this launch touches is full (all 4096 inputs inside the buffer,
// all S outputs wanted).  TAIL=true is the ragged last block: same code, the buffer
// descriptors' range check zero-fills the missing inputs and drops the surplus outputs.
template <bool TAIL>
__global__ __launch_bounds__(256, 3) void fir_cf32_ols4096_kernel(const float2 *__restrict__ in, size_t in_elems,
                                                                  float2 *__restrict__ out, size_t n_out,
                                                                  const float2 *__restrict__ Hspec, int Km1,
                                                                  const float2 *__restrict__ twtab, size_t b0, size_t nblocks)
{
    using namespace fft4k;
    __shared__ cf lds[LDS_ELEMS];
    const int j = threadIdx.x;
    const size_t S = (size_t)(N - Km1);
    size_t b = b0 + blockIdx.x;
    if (b >= nblocks) return;
    // loop invariants of the persistent workgroup, in registers: the lane's twiddles and its
    // 16 bins of H.  Nothing but the stream itself is loaded inside the loop.
    Twiddles tw;
    load_twiddles(tw, twtab, j);
    cf H[16];
#pragma unroll
    for (int k = 0; k < 16; k++) H[k] = reinterpret_cast<const cf *>(Hspec)[j + 256 * k];
    auto in_rsrc = [&](size_t blk) {
        const size_t left = in_elems - blk * S;   // samples from the block start to the end of the buffer
        return make_rsrc(in + blk * S, (unsigned)((left < (size_t)N ? left : (size_t)N) * 8));
    };
    cf nx[16];
    load_frame<TAIL>(nx, in_rsrc(b), j);
    for (; b < nblocks; b += gridDim.x) {
        cf v[16];
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] = nx[r];
        const size_t bn = b + gridDim.x;
        if (bn < nblocks) load_frame<TAIL>(nx, in_rsrc(bn), j);  // in flight during this block's math
        pass1(v, lds, j, tw);
        pass2(v, lds, j, tw);
        pass3(v, lds, j, tw);
        // spectrum times H, re-ordered into natural register order for the next pass 1.
        // The inverse transform runs on the FORWARD passes: IFFT(z) = conj(FFT(conj(z))), so
        // one set of twiddles serves both directions (conjugated copies would double the
        // loop-invariant registers).  u = conj(v * H); the final conj rides on the store.
        cf u[16];
#pragma unroll
        for (int q = 0; q < 16; q++) u[bin_of(q)] = cmul_conj(v[q], H[bin_of(q)]);
        pass1(u, lds, j, tw);
        pass2(u, lds, j, tw);
        pass3(u, lds, j, tw);
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
    if (nfull > 0) {
        const unsigned grid = (unsigned)(nfull < 768 ? nfull : 768);   // 3 persistent workgroups per CU
        hipLaunchKernelGGL(fir_cf32_ols4096_kernel<false>, dim3(grid), dim3(256), 0, st, (const float2 *)in, in_elems,
                           (float2 *)out, n_out, (const float2 *)Hspec, (int)(K - 1), (const float2 *)tw4096, (size_t)0, nfull);
        PCX_LAUNCH_CHECK();
    }
    if (nblocks > nfull) {
        hipLaunchKernelGGL(fir_cf32_ols4096_kernel<true>, dim3((unsigned)(nblocks - nfull)), dim3(256), 0, st, (const float2 *)in,
                           in_elems, (float2 *)out, n_out, (const float2 *)Hspec, (int)(K - 1), (const float2 *)tw4096, nfull, nblocks);
        PCX_LAUNCH_CHECK();
    }
    return PCX_OK;
}

}  // namespace pcx
