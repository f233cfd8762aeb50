// clk_lab.hip -- does s_memtime follow the shader clock on gfx950?  (diagnostic, not part of the product: make -C tools clk_lab)
// One wave reads s_memtime (clock64) and s_memrealtime (wall_clock64, 100 MHz) around a spin of dependent FMAs; if s_memtime counts
// shader cycles, delta_memtime / delta_realtime * 100 MHz is the clock the wave ran at.  Run idle, and directly behind a burst of
// heavy launches on the same stream (DVFS reacts in milliseconds: the probe then sees the loaded clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void clk_probe(unsigned long long *out, int spins)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < spins; i++) a = __builtin_fmaf(a, b, 1e-7f);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = (unsigned long long)(a > 1e30f); }
}
__global__ void heavy(float4 *p, size_t n, int reps)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    float4 v = p[i % n];
    for (int r = 0; r < reps; r++) { v.x = __builtin_fmaf(v.x, v.y, v.z); v.y = __builtin_fmaf(v.y, v.z, v.w); v.z = __builtin_fmaf(v.z, v.w, v.x); v.w = __builtin_fmaf(v.w, v.x, v.y); }
    p[i % n] = v;
}
int main()
{
    unsigned long long *d, h[3];
    float4 *buf;
    const size_t n = 64u << 20;
    hipMalloc(&d, 64); hipMalloc(&buf, n * sizeof(float4)); hipMemset(buf, 0, n * sizeof(float4));
    for (int spins : {2000, 20000}) {
        for (int rep = 0; rep < 3; rep++) {
            hipDeviceSynchronize();
            clk_probe<<<1, 64>>>(d, spins);
            hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
            printf("idle            spins %6d: memtime %llu realtime %llu -> %.1f MHz if memtime is the shader clock (%.1f us)\n", spins, h[0], h[1], 100.0 * h[0] / h[1], h[1] / 100.0);
        }
        for (int rep = 0; rep < 3; rep++) {
            for (int k = 0; k < 400; k++) heavy<<<n / 256, 256>>>(buf, n, 64);
            clk_probe<<<1, 64>>>(d, spins);
            hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
            printf("behind 400 heavy spins %6d: memtime %llu realtime %llu -> %.1f MHz if memtime is the shader clock (%.1f us)\n", spins, h[0], h[1], 100.0 * h[0] / h[1], h[1] / 100.0);
        }
    }
    return 0;
}
