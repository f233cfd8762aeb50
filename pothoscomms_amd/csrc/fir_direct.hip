// fir_direct.hip -- time-domain /comms/fir_filter for complex_float32, M = L = 1:
// the convolution loop FIRFilter.cpp:294-300 as an overlap-save tile.
//
//   y[n] = sum_k h[k] * xh[n + K-1-k] = sum_m g[m] * xh[n + m],   g[m] = h[K-1-m]
// (xh = input buffer with the K-1 history samples in front, FIRFilter.cpp:281).
//
// One workgroup = 256 lanes = one tile of TILE = 2048 consecutive outputs.  The tile's
// inputs (TILE + Kp + 8 samples: tile + tap-length halo) are staged once in LDS; each lane
// produces R = 8 consecutive outputs from a register sliding window, so one LDS read of 8
// new samples feeds 64 complex MACs (256 FMA).  Taps are read with wave-uniform (scalar)
// loads, 8 taps per block of the m loop.  LDS image is padded by one sample per 8
// (phys(i) = i + i/8): lane stride becomes 9 samples = 18 dwords, which makes the
// 32-lane ds_read_b64 groups hit 64 distinct banks.
//
// Direct form is FP32-FMA-bound, not HBM-bound (8K flop per 16 algorithmic bytes;
// K=255: 127 flop/B against a machine balance of ~20): its ceiling is ~77 Gsamples/s.
// The frequency-domain kernel (fir_ols.hip) is the HBM-bound path for long filters;
// this one serves short filters and is the cross-check for it.
#include "pcx_internal.hpp"

namespace pcx {

namespace {
constexpr int kLanes = 256;
constexpr int kR = 8;                  // outputs per lane
constexpr int kTile = kLanes * kR;     // 2048 outputs per workgroup
}  // namespace

__device__ __forceinline__ int phys(int i) { return i + (i >> 3); }

// the tile's MAC loop: acc[r] = sum_m g[m] * s[t*8 + r + m]
template <bool CTAPS>
__device__ __forceinline__ void fir_tile_compute(const float2 *s, int t, const float *__restrict__ taps_rev, int Kp, float2 (&acc)[kR])
{
#pragma unroll
    for (int r = 0; r < kR; r++) acc[r] = make_float2(0.f, 0.f);

    // window w[q] = s[t*8 + 8*mb + q]; phys = t*9 + 9*mb + q for q < 8
    float2 w[2 * kR];
    const float2 *sp = s + t * 9;
#pragma unroll
    for (int q = 0; q < kR; q++) w[q] = sp[q];
    const int nblk = Kp / kR;
    for (int mb = 0; mb < nblk; mb++) {
        sp += 9;
#pragma unroll
        for (int q = 0; q < kR; q++) w[kR + q] = sp[q];
        const float *gp = taps_rev + (size_t)mb * kR * (CTAPS ? 2 : 1);
#pragma unroll
        for (int u = 0; u < kR; u++) {
            const float gr = CTAPS ? gp[2 * u] : gp[u];
            const float gi = CTAPS ? gp[2 * u + 1] : 0.f;
#pragma unroll
            for (int r = 0; r < kR; r++) {
                const float2 x = w[r + u];
                acc[r].x = __builtin_fmaf(gr, x.x, acc[r].x);
                acc[r].y = __builtin_fmaf(gr, x.y, acc[r].y);
                if (CTAPS) {
                    acc[r].x = __builtin_fmaf(-gi, x.y, acc[r].x);
                    acc[r].y = __builtin_fmaf(gi, x.x, acc[r].y);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < kR; q++) w[q] = w[kR + q];
    }
}

// CTAPS: complex taps (taps_rev holds float2) or real taps (taps_rev holds float)
template <bool CTAPS>
__global__ __launch_bounds__(kLanes) void fir_cf32_direct_kernel(const float2 *__restrict__ in, size_t in_elems,
                                                                 float2 *__restrict__ out, size_t n_out,
                                                                 const float *__restrict__ taps_rev, int Kp)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *s = reinterpret_cast<float2 *>(smem_raw);
    const int t = threadIdx.x;
    const size_t tile0 = (size_t)blockIdx.x * kTile;

    // stage tile + halo: s[phys(i)] = xh[tile0 + i], zero beyond the buffer
    const int n_stage = kTile + Kp + 8;
    for (int i = t; i < n_stage; i += kLanes) {
        const size_t gi = tile0 + (size_t)i;
        s[phys(i)] = gi < in_elems ? in[gi] : make_float2(0.f, 0.f);
    }
    __syncthreads();

    float2 acc[kR];
    fir_tile_compute<CTAPS>(s, t, taps_rev, Kp, acc);

    const size_t o0 = tile0 + (size_t)t * kR;
    if (o0 + kR <= n_out) {
        float4 *op = reinterpret_cast<float4 *>(out + o0);
        if ((reinterpret_cast<uintptr_t>(op) & 15) == 0) {
#pragma unroll
            for (int r = 0; r < kR; r += 2) op[r / 2] = make_float4(acc[r].x, acc[r].y, acc[r + 1].x, acc[r + 1].y);
        } else {
#pragma unroll
            for (int r = 0; r < kR; r++) out[o0 + r] = acc[r];
        }
    } else {
#pragma unroll
        for (int r = 0; r < kR; r++)
            if (o0 + r < n_out) out[o0 + r] = acc[r];
    }
}

int launch_fir_cf32_direct(const void *in, size_t in_elems, void *out, size_t n_out, const void *taps_rev, size_t K,
                           size_t Kp, hipStream_t st)
{
    (void)K;
    if (n_out == 0) return PCX_OK;
    const int n_stage = kTile + (int)Kp + 8;
    const size_t lds = (size_t)(n_stage + n_stage / 8 + 2) * sizeof(float2);
    if (lds > 64 * 1024) { set_error("fir direct: %zu taps exceed the LDS tile plan", K); return PCX_ERR_UNSUPPORTED; }
    const size_t grid = (n_out + kTile - 1) / kTile;
    hipLaunchKernelGGL(fir_cf32_direct_kernel<true>, dim3((unsigned)grid), dim3(kLanes), lds, st, (const float2 *)in, in_elems,
                       (float2 *)out, n_out, (const float *)taps_rev, (int)Kp);
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}


// --------------------------------------------------------------------------------- //
// fused Rotate -> FIR -> FreqDemod (BASELINE configs[4]): one pass, cf32 in, f32 out.
// Rotate's phasor is folded into the taps on the host (FIR is linear), the FIR tile is
// the loop above, and FreqDemod (FreqDemod.cpp:60-67) runs on the accumulators before
// anything is stored: d[n] = arg(y[n] * conj(y[n-1])).  Tiles overlap by ONE output so
// every y[n-1] a tile needs is produced inside the tile (stride kTile-1); y[-1] is the
// state carried from the previous call (*prev_in, already conjugated; zero after reset).
// --------------------------------------------------------------------------------- //
__global__ __launch_bounds__(kLanes) void fmchain_cf32_kernel(const float2 *__restrict__ in, size_t in_elems,
                                                              float *__restrict__ out, size_t n_out,
                                                              const float *__restrict__ taps_rev, int Kp,
                                                              const float2 *__restrict__ prev_in, float2 *__restrict__ prev_out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *s = reinterpret_cast<float2 *>(smem_raw);
    const int n_stage = kTile + Kp + 8;
    float2 *last = s + ((n_stage + n_stage / 8 + 2 + 1) & ~1);  // after the tile image, 16-byte aligned
    const int t = threadIdx.x;
    const long long base = (long long)blockIdx.x * (kTile - 1) - 1;  // first FIR output index of the tile

    for (int i = t; i < n_stage; i += kLanes) {
        const long long gi = base + i;
        s[phys(i)] = (gi >= 0 && (size_t)gi < in_elems) ? in[gi] : make_float2(0.f, 0.f);
    }
    __syncthreads();

    float2 acc[kR];
    fir_tile_compute<true>(s, t, taps_rev, Kp, acc);

    last[t] = acc[kR - 1];
    __syncthreads();
    const long long e0 = base + (long long)t * kR;  // FIR output index of acc[0]
    float2 pv;                                       // _prev for acc[0]
    if (t > 0) pv = make_float2(last[t - 1].x, -last[t - 1].y);
    else pv = make_float2(0.f, 0.f);
    if (e0 == -1) {  // tile 0, lane 0: acc[0] stands for the carried sample
        pv = prev_in[0];
    }
#pragma unroll
    for (int r = 0; r < kR; r++) {
        const long long n = e0 + r;
        const bool first_of_tile = (t == 0 && r == 0);
        if (!first_of_tile && n >= 0 && (size_t)n < n_out) {
            const float a = acc[r].x, b = acc[r].y, c = pv.x, d = pv.y;
            const float re = a * c - b * d, im = a * d + b * c;
            out[n] = atan2f(im, re);
            if ((size_t)n == n_out - 1) prev_out[0] = make_float2(a, -b);
        }
        if (!(e0 == -1 && r == 0)) pv = make_float2(acc[r].x, -acc[r].y);
    }
}

int launch_fmchain_cf32(const void *in, size_t in_elems, void *out, size_t n_out, const void *taps_rev, size_t K,
                        size_t Kp, const void *prev_in, void *prev_out, hipStream_t st)
{
    if (n_out == 0) return PCX_OK;
    const int n_stage = kTile + (int)Kp + 8;
    const size_t lds = (size_t)(((n_stage + n_stage / 8 + 2 + 1) & ~1) + kLanes) * sizeof(float2);
    if (lds > 60 * 1024) { set_error("fm chain: %zu taps exceed the LDS tile plan", K); return PCX_ERR_UNSUPPORTED; }
    // tile b covers FIR outputs b*(kTile-1)-1 .. ; need the last tile to reach n_out-1
    const size_t grid = (n_out + (kTile - 1) - 1) / (kTile - 1);
    hipLaunchKernelGGL(fmchain_cf32_kernel, dim3((unsigned)grid), dim3(kLanes), lds, st, (const float2 *)in, in_elems,
                       (float *)out, n_out, (const float *)taps_rev, (int)Kp, (const float2 *)prev_in, (float2 *)prev_out);
    PCX_LAUNCH_CHECK();
    return PCX_OK;
}

}  // namespace pcx
