// ubench.hip -- access-pattern and issue-rate microbenchmarks used to size the kernels
// (DESIGN.md cites the numbers).  Build: hipcc -O3 --offload-arch=gfx950 tools/ubench.hip -o tools/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void copy16(const float4* __restrict__ in, float4* __restrict__ out, size_t n)
{
    size_t stride = (size_t)gridDim.x * 256 * 4;
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n; i += stride) {
        float4 a = in[i], b = in[i + 256], c = in[i + 512], d = in[i + 768];
        out[i] = a; out[i + 256] = b; out[i + 512] = c; out[i + 768] = d;
    }
}
typedef float f4 __attribute__((ext_vector_type(4)));
template <bool NT, int U>
__global__ __launch_bounds__(256) void copy16u(const f4* __restrict__ in, f4* __restrict__ out, size_t n)
{
    size_t stride = (size_t)gridDim.x * 256 * U;
    for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i < n; i += stride) {
        f4 a[U];
#pragma unroll
        for (int u = 0; u < U; u++) a[u] = NT ? __builtin_nontemporal_load(&in[i + 256 * u]) : in[i + 256 * u];
#pragma unroll
        for (int u = 0; u < U; u++) { if (NT) __builtin_nontemporal_store(a[u], &out[i + 256 * u]); else out[i + 256 * u] = a[u]; }
    }
}
template <bool NT>
__global__ __launch_bounds__(256) void read16(const f4* __restrict__ in, float* __restrict__ out, size_t n)
{
    size_t stride = (size_t)gridDim.x * 256 * 4;
    float s = 0;
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n; i += stride) {
#pragma unroll
        for (int u = 0; u < 4; u++) { f4 a = NT ? __builtin_nontemporal_load(&in[i + 256 * u]) : in[i + 256 * u]; s += a.x + a.y + a.z + a.w; }
    }
    if (s == 123.456f) out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <bool NT>
__global__ __launch_bounds__(256) void write16(f4* __restrict__ out, size_t n)
{
    size_t stride = (size_t)gridDim.x * 256 * 4;
    const f4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n; i += stride) {
#pragma unroll
        for (int u = 0; u < 4; u++) { if (NT) __builtin_nontemporal_store(v, &out[i + 256 * u]); else out[i + 256 * u] = v; }
    }
}
// frame-structured copy: one 4096 x float2 frame per workgroup pass, lane j touches j + 256 r (8 B per lane)
template <int WAVES_HINT>
__global__ __launch_bounds__(256) void frame_copy8(const float2* __restrict__ in, float2* __restrict__ out, size_t nframes)
{
    for (size_t f = blockIdx.x; f < nframes; f += gridDim.x) {
        const float2* x = in + f * 4096; float2* y = out + f * 4096;
        float2 v[16];
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] = x[threadIdx.x + 256 * r];
#pragma unroll
        for (int r = 0; r < 16; r++) y[threadIdx.x + 256 * r] = v[r];
    }
}
// same frame, 16 B per lane: 128 lanes per frame, 2 frames per 256-lane workgroup
__global__ __launch_bounds__(256) void frame_copy16(const float4* __restrict__ in, float4* __restrict__ out, size_t nframes)
{
    const int t = threadIdx.x & 127, h = threadIdx.x >> 7;
    for (size_t f = 2 * (size_t)blockIdx.x + h; f < nframes; f += 2 * (size_t)gridDim.x) {
        const float4* x = in + f * 2048; float4* y = out + f * 2048;
        float4 v[16];
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] = x[t + 128 * r];
#pragma unroll
        for (int r = 0; r < 16; r++) y[t + 128 * r] = v[r];
    }
}
// the same two with non-temporal accesses (what the FFT / FIR kernels use on rows nobody else reads)
typedef float f2v __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void frame_copy8nt(const f2v* __restrict__ in, f2v* __restrict__ out, size_t nframes)
{
    for (size_t f = blockIdx.x; f < nframes; f += gridDim.x) {
        const f2v* x = in + f * 4096; f2v* y = out + f * 4096;
        f2v v[16];
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] = __builtin_nontemporal_load(&x[threadIdx.x + 256 * r]);
#pragma unroll
        for (int r = 0; r < 16; r++) __builtin_nontemporal_store(v[r], &y[threadIdx.x + 256 * r]);
    }
}
__global__ __launch_bounds__(256) void frame_copy16nt(const f4* __restrict__ in, f4* __restrict__ out, size_t nframes)
{
    // one frame per workgroup, 16 B per lane: lane j touches elements (2j, 2j+1) + 512 r, r < 8
    for (size_t f = blockIdx.x; f < nframes; f += gridDim.x) {
        const f4* x = in + f * 2048; f4* y = out + f * 2048;
        f4 v[8];
#pragma unroll
        for (int r = 0; r < 8; r++) v[r] = __builtin_nontemporal_load(&x[threadIdx.x + 256 * r]);
#pragma unroll
        for (int r = 0; r < 8; r++) __builtin_nontemporal_store(v[r], &y[threadIdx.x + 256 * r]);
    }
}
// frame copy through LDS with barriers (2 exchanges) to mimic the FFT's phase structure
__global__ __launch_bounds__(256) void frame_copy8_lds(const float2* __restrict__ in, float2* __restrict__ out, size_t nframes)
{
    __shared__ float2 lds[4096 + 256];
    const int j = threadIdx.x;
    for (size_t f = blockIdx.x; f < nframes; f += gridDim.x) {
        const float2* x = in + f * 4096; float2* y = out + f * 4096;
        float2 v[16];
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] = x[j + 256 * r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; r++) lds[17 * j + r] = v[r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] = lds[j + (j >> 4) + 272 * r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; r++) lds[(j >> 4) * 272 + (j & 15) + 17 * r] = v[r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] = lds[j + (j >> 4) + 272 * r];
#pragma unroll
        for (int r = 0; r < 16; r++) y[j + 256 * r] = v[r];
    }
}
// VALU issue rate: dependent-free FMA chains
template <bool PK>
__global__ __launch_bounds__(256) void fma_rate(float* out, int iters)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a[16];
    for (int i = 0; i < 16; i++) a[i] = f2{(float)threadIdx.x + i, 1.0f};
    f2 m = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (PK) a[i] = __builtin_elementwise_fma(a[i], m, c);
            else { a[i].x = __builtin_fmaf(a[i].x, m.x, c.x); a[i].y = __builtin_fmaf(a[i].y, m.y, c.y); }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; i++) s += a[i].x + a[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F>
static float time_ms(F launch, int reps)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; i++) launch();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main()
{
    const size_t nframes = 32768;            // 1 GiB in + 1 GiB out
    const size_t bytes = nframes * 4096 * 8;
    void *in, *out;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes));
    CK(hipMemset(in, 1, bytes));
    const double gb = 2.0 * bytes / 1e9;
    for (int grid : {1024, 2048, 4096}) {
        float ms = time_ms([&] { hipLaunchKernelGGL(copy16, dim3(grid), dim3(256), 0, 0, (const float4*)in, (float4*)out, bytes / 16); }, 10);
        printf("copy16          grid %5d: %.3f ms  %.0f GB/s\n", grid, ms, gb / ms * 1e3);
    }
    for (int grid : {2048, 8192}) {
        float ms = time_ms([&] { hipLaunchKernelGGL((copy16u<false, 8>), dim3(grid), dim3(256), 0, 0, (const f4*)in, (f4*)out, bytes / 16); }, 10);
        printf("copy16 x8       grid %5d: %.3f ms  %.0f GB/s\n", grid, ms, gb / ms * 1e3);
        ms = time_ms([&] { hipLaunchKernelGGL((copy16u<true, 4>), dim3(grid), dim3(256), 0, 0, (const f4*)in, (f4*)out, bytes / 16); }, 10);
        printf("copy16 nt x4    grid %5d: %.3f ms  %.0f GB/s\n", grid, ms, gb / ms * 1e3);
        ms = time_ms([&] { hipLaunchKernelGGL((copy16u<true, 8>), dim3(grid), dim3(256), 0, 0, (const f4*)in, (f4*)out, bytes / 16); }, 10);
        printf("copy16 nt x8    grid %5d: %.3f ms  %.0f GB/s\n", grid, ms, gb / ms * 1e3);
        ms = time_ms([&] { hipLaunchKernelGGL(read16<false>, dim3(grid), dim3(256), 0, 0, (const f4*)in, (float*)out, bytes / 16); }, 10);
        printf("read16          grid %5d: %.3f ms  %.0f GB/s (read only)\n", grid, ms, gb / 2 / ms * 1e3);
        ms = time_ms([&] { hipLaunchKernelGGL(read16<true>, dim3(grid), dim3(256), 0, 0, (const f4*)in, (float*)out, bytes / 16); }, 10);
        printf("read16 nt       grid %5d: %.3f ms  %.0f GB/s (read only)\n", grid, ms, gb / 2 / ms * 1e3);
        ms = time_ms([&] { hipLaunchKernelGGL(write16<false>, dim3(grid), dim3(256), 0, 0, (f4*)out, bytes / 16); }, 10);
        printf("write16         grid %5d: %.3f ms  %.0f GB/s (write only)\n", grid, ms, gb / 2 / ms * 1e3);
        ms = time_ms([&] { hipLaunchKernelGGL(write16<true>, dim3(grid), dim3(256), 0, 0, (f4*)out, bytes / 16); }, 10);
        printf("write16 nt      grid %5d: %.3f ms  %.0f GB/s (write only)\n", grid, ms, gb / 2 / ms * 1e3);
    }
    for (int grid : {1024, 2048, 4096, 32768}) {
        float ms = time_ms([&] { hipLaunchKernelGGL(frame_copy8<0>, dim3(grid), dim3(256), 0, 0, (const float2*)in, (float2*)out, nframes); }, 10);
        printf("frame_copy8     grid %5d: %.3f ms  %.0f GB/s\n", grid, ms, gb / ms * 1e3);
    }
    for (int grid : {512, 1024, 2048, 16384}) {
        float ms = time_ms([&] { hipLaunchKernelGGL(frame_copy16, dim3(grid), dim3(256), 0, 0, (const float4*)in, (float4*)out, nframes); }, 10);
        printf("frame_copy16    grid %5d: %.3f ms  %.0f GB/s\n", grid, ms, gb / ms * 1e3);
    }
    for (int grid : {1024, 4096, 8192, 32768}) {
        float ms = time_ms([&] { hipLaunchKernelGGL(frame_copy8nt, dim3(grid), dim3(256), 0, 0, (const f2v*)in, (f2v*)out, nframes); }, 300);
        printf("frame_copy8 nt  grid %5d: %.3f ms  %.0f GB/s   (300 launches)\n", grid, ms, gb / ms * 1e3);
        ms = time_ms([&] { hipLaunchKernelGGL(frame_copy16nt, dim3(grid), dim3(256), 0, 0, (const f4*)in, (f4*)out, nframes); }, 300);
        printf("frame_copy16 nt grid %5d: %.3f ms  %.0f GB/s   (300 launches)\n", grid, ms, gb / ms * 1e3);
        ms = time_ms([&] { hipLaunchKernelGGL((copy16u<true, 4>), dim3(grid * 2), dim3(256), 0, 0, (const f4*)in, (f4*)out, bytes / 16); }, 300);
        printf("copy16 nt x4    grid %5d: %.3f ms  %.0f GB/s   (300 launches)\n", grid * 2, ms, gb / ms * 1e3);
    }
    for (int grid : {1024, 2048, 32768}) {
        float ms = time_ms([&] { hipLaunchKernelGGL(frame_copy8_lds, dim3(grid), dim3(256), 0, 0, (const float2*)in, (float2*)out, nframes); }, 10);
        printf("frame_copy8_lds grid %5d: %.3f ms  %.0f GB/s\n", grid, ms, gb / ms * 1e3);
    }
    {
        const int iters = 4096, grid = 256 * 8;
        float ms = time_ms([&] { hipLaunchKernelGGL(fma_rate<false>, dim3(grid), dim3(256), 0, 0, (float*)out, iters); }, 5);
        double fl = (double)grid * 256 * iters * 16 * 2 * 2;
        printf("v_fma_f32    : %.3f ms  %.1f TFLOP/s\n", ms, fl / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(fma_rate<true>, dim3(grid), dim3(256), 0, 0, (float*)out, iters); }, 5);
        printf("v_pk_fma_f32 : %.3f ms  %.1f TFLOP/s\n", ms, fl / ms / 1e9);
    }
    return 0;
}
