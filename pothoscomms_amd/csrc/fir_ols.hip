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

__global__ __launch_bounds__(256) void fir_cf32_ols4096_kernel(const float2 *__restrict__ in, size_t in_elems,
                                                               float2 *__restrict__ out, size_t n_out,
                                                               const float2 *__restrict__ Hspec, int Km1,
                                                               fft4k::Tables tb, size_t nblocks)
{
    using namespace fft4k;
    __shared__ float2 lds[LDS_ELEMS];
    const int j = threadIdx.x;
    const size_t S = (size_t)(N - Km1);
    for (size_t b = blockIdx.x; b < nblocks; b += gridDim.x) {
        const size_t base = b * S;
        float2 v[16];
        if (base + N <= in_elems) {
#pragma unroll
            for (int r = 0; r < 16; r++) v[r] = in[base + j + 256 * r];
        } else {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const size_t gi = base + j + 256 * r;
                v[r] = gi < in_elems ? in[gi] : make_float2(0.f, 0.f);
            }
        }
        pass1<false>(v, lds, j);
        pass2<false>(v, lds, j, tb);
        pass3<false>(v, lds, j, tb);
        // spectrum times H, re-ordered into natural register order for the inverse pass 1
        float2 u[16];
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int k = bin_of(q);
            u[k] = cmul(v[q], Hspec[j + 256 * k]);
        }
        pass1<true>(u, lds, j);
        pass2<true>(u, lds, j, tb);
        pass3<true>(u, lds, j, tb);
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int i = j + 256 * bin_of(q);  // time index inside the block
            if (i >= Km1) {
                const size_t o = base + (size_t)(i - Km1);
                if (o < n_out) out[o] = u[q];
            }
        }
    }
}

int launch_fir_cf32_ols4096(const void *in, size_t in_elems, void *out, size_t n_out, const void *Hspec, size_t K,
                            const void *tw4096, hipStream_t st)
{
    if (n_out == 0) return PCX_OK;
    if (K < 1 || K > 2049) { set_error("fir ols: K=%zu outside 1..2049", K); return PCX_ERR_UNSUPPORTED; }
    fft4k::Tables tb;
    tb.tw2 = static_cast<const float2 *>(tw4096);
    tb.tw3 = tb.tw2 + 256;
    const size_t S = 4096 - (K - 1);
    const size_t nblocks = (n_out + S - 1) / S;
    const unsigned grid = (unsigned)(nblocks < 2048 ? nblocks : 2048);
    hipLaunchKernelGGL(fir_cf32_ols4096_kernel, dim3(grid), dim3(256), 0, st, (const float2 *)in, in_elems, (float2 *)out,
                       n_out, (const float2 *)Hspec, (int)(K - 1), tb, nblocks);
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}

}  // namespace pcx
