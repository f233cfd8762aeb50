// hwid_probe.hip -- where do the 1024 workgroups of a persistent launch with the headline kernel's footprint (256 lanes, 128 VGPRs,
// 36.7 KB LDS) land?  Every workgroup records HW_ID / XCC_ID; the host prints the (xcc, se, cu) histogram and where blockIdx 0..15 went.
// Build: make -C tools hwid_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <tuple>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256, 4) void probe(unsigned *out, float *sink, int spin)
{
    __shared__ float lds[9184];                     // 36,736 B
    float acc[96];                                   // keeps the register budget near the real kernel's
#pragma unroll
    for (int i = 0; i < 96; i++) acc[i] = (float)(threadIdx.x + i);
    lds[threadIdx.x] = 1.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) {     // stay resident until every workgroup is placed
#pragma unroll
        for (int i = 0; i < 96; i++) acc[i] = acc[i] * 1.0001f + lds[(threadIdx.x + i) & 255];
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 96; i++) s += acc[i];
    if (s == 12345.f) sink[0] = s;
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // HW_REG_HW_ID, 32 bits
        out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID
    }
}

int main()
{
    const int G = 1024;
    unsigned *d; float *sink;
    CK(hipMalloc(&d, G * 8)); CK(hipMalloc(&sink, 4));
    hipLaunchKernelGGL(probe, dim3(G), dim3(256), 0, 0, d, sink, 20000);       // 200 us
    CK(hipDeviceSynchronize());
    std::vector<unsigned> h(2 * G);
    CK(hipMemcpy(h.data(), d, G * 8, hipMemcpyDeviceToHost));
    std::map<std::tuple<unsigned, unsigned, unsigned, unsigned>, int> cnt;
    for (int w = 0; w < G; w++) {
        const unsigned id = h[2 * w], xcc = h[2 * w + 1] & 15;
        const unsigned cu = (id >> 8) & 15, sh = (id >> 12) & 1, se = (id >> 13) & 7;
        cnt[{xcc, se, sh, cu}]++;
        if (w < 24 || (w % 256) < 2) printf("wg %4d: xcc %u se %u sh %u cu %u  (HW_ID %08x XCC_ID %08x)\n", w, xcc, se, sh, cu, id, h[2 * w + 1]);
    }
    printf("%zu distinct (xcc, se, sh, cu)\n", cnt.size());
    std::map<int, int> hist;
    for (auto &kv : cnt) hist[kv.second]++;
    for (auto &kv : hist) printf("  %d CUs hold %d workgroups\n", kv.second, kv.first);
    for (auto &kv : cnt) if (std::get<0>(kv.first) == 0) printf("  xcc0 se %u sh %u cu %u: %d\n", std::get<1>(kv.first), std::get<2>(kv.first), std::get<3>(kv.first), kv.second);
    return 0;
}
