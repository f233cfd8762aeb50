// wave_lab.hip -- premise test for a ONE-WAVE-PER-BLOCK overlap-save pipeline (radix-64 x 64, no barriers).
//
// The four-wave workgroup of fir_cf32_ols4096_kernel pays for its LDS rendezvous (profiles/r02/ols_lab.md).  A block held by
// a single wave (64 lanes x 64 points) needs no barrier at all, but leaves one wave per SIMD.  Before building the transform,
// this proxy runs the SKELETON of such a kernel -- the same bytes, the same instruction volume, no FFT -- to see what a lone
// wave per SIMD can overlap:
//     per block and wave: 64 row loads of 512 B (the next block's, into registers of their own: one wave per SIMD owns 512),
//     VALU packed FMAs on the current block's 64 values per lane (independent chains, real data),
//     LDS two 64 x 64 transposes through a wave-private 33 KB image (64 ds_write_b64 + 64 ds_read_b64 each),
//     60 row stores.  Blocks are drawn two at a time from one counter (no barrier needed: lane 0 draws, readfirstlane).
// Build: make -C tools wave_lab     Run: tools/wave_lab [seconds]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float cf __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, unsigned bytes)
{
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo), 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

constexpr int N = 4096, ROWS = 64;
constexpr int IMG = 64 * 65;            // padded 64 x 64 image of cf (33,280 B)

// VOPS: packed FMAs per lane and block (the pipeline's ~2700 + register moves); XCH: LDS transposes per block; PREF: prefetch
template <int VOPS, int XCH, bool PREF, int LDS_PAD>
__global__ __launch_bounds__(64, 1) void wave_kernel(const float2 *__restrict__ in, float2 *__restrict__ out, size_t n_out, int Kov,
                                                    size_t nblocks, unsigned *__restrict__ ctr, unsigned ctr_base)
{
    __shared__ cf lds[IMG + LDS_PAD];    // LDS_PAD sizes the allocation so that exactly 4 (or 3) such workgroups fit a CU
    const int l = threadIdx.x;
    const size_t S = (size_t)(N - Kov);
    const unsigned nchunks = (unsigned)((nblocks + 1) / 2);
    auto draw = [&]() -> unsigned {
        unsigned v = 0;
        if (l == 0) v = atomicAdd(ctr, 1u) - ctr_base;
        return __builtin_amdgcn_readfirstlane(v);
    };
    auto fetch = [&](cf (&dst)[ROWS], size_t blk) {
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(in + blk * S, N * 8);
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            const u32x2 t = (r < 4 || r >= 60) ? __builtin_amdgcn_raw_buffer_load_b64(rs, l * 8, 512 * r, 0)
                                               : __builtin_amdgcn_raw_buffer_load_b64(rs, l * 8, 512 * r, 2);
            dst[r] = cf{__uint_as_float(t.x), __uint_as_float(t.y)};
        }
    };
    unsigned chunk = draw();
    if (chunk >= nchunks) return;
    unsigned sub = 0;
    size_t b = chunk;
    cf nx[ROWS];
    if (PREF) fetch(nx, b);
    for (;;) {
        cf v[ROWS];
        // which block follows (known one block ahead: the draw for the next chunk is made at the first block of a chunk)
        const bool pair = (size_t)chunk + nchunks < nblocks;
        unsigned nextchunk = 0;
        size_t bn;
        bool more;
        if (sub == 0 && pair) { bn = (size_t)chunk + nchunks; more = true; nextchunk = chunk; }
        else { nextchunk = draw(); bn = nextchunk; more = nextchunk < nchunks; }
        if (PREF) {
#pragma unroll
            for (int r = 0; r < ROWS; r++) v[r] = nx[r];
            if (more) fetch(nx, bn);
        } else {
            fetch(v, b);
        }
        // ---- the work of two radix-64 passes per transform, twice: VALU volume + LDS transposes
        cf m0 = v[1], m1 = v[2];
#pragma unroll 1
        for (int rep = 0; rep < XCH; rep++) {
#pragma unroll 1
            for (int it = 0; it < VOPS / (XCH ? XCH : 1) / ROWS; it++) {
#pragma unroll
                for (int r = 0; r < ROWS; r++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[r]) : "v"(m0), "v"(m1));
            }
            // transpose: lane l writes row l, reads column l (wave-private: LDS keeps a wave's operations in order)
#pragma unroll
            for (int k = 0; k < ROWS; k++) lds[65 * l + k] = v[k];
#pragma unroll
            for (int r = 0; r < ROWS; r++) v[r] = lds[65 * r + l];
        }
        const size_t room = n_out - b * S;
        const __amdgpu_buffer_rsrc_t ws = make_rsrc(out + b * S, (unsigned)((room < S ? room : S) * 8));
        const unsigned vbase = (unsigned)(l - Kov) * 8u;
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            if (64 * r + 63 < Kov) continue;
            const u32x2 t = {__float_as_uint(v[r].x), __float_as_uint(v[r].y)};
            __builtin_amdgcn_raw_buffer_store_b64(t, ws, (int)(vbase + (unsigned)(64 * r) * 8u), 0, 2);
        }
        if (!more) break;
        if (sub == 0 && pair) sub = 1; else { chunk = nextchunk; sub = 0; }
        b = bn;
    }
}

__global__ void fill_kernel(float *p, size_t n, unsigned long long seed)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = (i + seed) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        p[i] = (float)((double)(z >> 40) / 8388608.0 - 1.0) * 1e-3f;
    }
}

typedef void (*KFn)(const float2 *, float2 *, size_t, int, size_t, unsigned *, unsigned);
struct Cfg { const char *name; KFn k; unsigned grid; };

int main(int argc, char **argv)
{
    const double secs = argc > 1 ? atof(argv[1]) : 3.0;
    const size_t n = 64ull << 20;
    const int Kov = 256;
    const size_t S = 4096 - Kov, nblocks = (n + S - 1) / S, in_elems = nblocks * S + 4096;
    float2 *x, *y;
    unsigned *ctr;
    CK(hipMalloc(&x, in_elems * 8)); CK(hipMalloc(&y, (nblocks * S + 64) * 8)); CK(hipMalloc(&ctr, 64)); CK(hipMemset(ctr, 0, 64));
    hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, (float *)x, in_elems * 2, 2ull);
    unsigned base = 0;
    // LDS per workgroup: 33,280 B image + pad.  4 per CU: <= 40,960 B.  (3 per CU: 53,000)
    const Cfg cfgs[] = {
        {"copy only, prefetch, 4 waves/CU", wave_kernel<0, 0, true, 900>, 1024},
        {"copy only, no prefetch, 4 waves/CU", wave_kernel<0, 0, false, 900>, 1024},
        {"2 transposes, no VALU, prefetch", wave_kernel<0, 2, true, 900>, 1024},
        {"4 transposes, no VALU, prefetch", wave_kernel<0, 4, true, 900>, 1024},
        {"4 transposes + 1536 pk FMA, prefetch", wave_kernel<1536, 4, true, 900>, 1024},
        {"4 transposes + 3072 pk FMA, prefetch", wave_kernel<3072, 4, true, 900>, 1024},
        {"4 transposes + 3072 pk FMA, NO prefetch", wave_kernel<3072, 4, false, 900>, 1024},
        {"4 transposes + 4096 pk FMA, prefetch", wave_kernel<4096, 4, true, 900>, 1024},
        {"4 transposes + 3072 pk FMA, prefetch, grid 2048", wave_kernel<3072, 4, true, 900>, 2048},
    };
    printf("# one wave per block skeleton: %zu blocks of %zu outputs, %zu samples, %.1f s per configuration\n", nblocks, S, n, secs);
    printf("%-52s %9s %8s\n", "config", "ms/launch", "TB/s");
    for (const Cfg &c : cfgs) {
        auto launch = [&] {
            hipLaunchKernelGGL(c.k, dim3(c.grid), dim3(64), 0, 0, x, y, n, Kov, nblocks, ctr, base);
            // draws per launch: one per chunk + one failed draw per workgroup that ends by drawing (not all do) -> reset instead
            CK(hipMemsetAsync(ctr, 0, 4, 0));
        };
        CK(hipDeviceSynchronize());
        const auto w0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count() < secs * 0.5) {
            for (int i = 0; i < 100; i++) launch();
            CK(hipDeviceSynchronize());
        }
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        size_t iters = 0;
        CK(hipEventRecord(e0, 0));
        const auto w1 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - w1).count() < secs * 0.5) {
            for (int i = 0; i < 100; i++) launch();
            iters += 100;
            CK(hipStreamSynchronize(0));
        }
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double per = ms / (double)iters;
        printf("%-52s %9.4f %8.3f\n", c.name, per, 16.0 * (double)n / (per * 1e-3) / 1e12);
        fflush(stdout);
    }
    return 0;
}
