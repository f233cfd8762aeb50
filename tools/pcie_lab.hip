// pcie_lab.hip -- do H2D and D2H copies of page-locked buffers overlap on this platform, and at what chunk size?
// (decides whether the host-pointer entry points pipeline their staging).  Build: make -C tools pcie_lab
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(const float4 *in, float4 *out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
int main()
{
    const size_t bytes = 128ull << 20;
    void *hin, *hout, *din, *dout, *pg;
    CK(hipHostMalloc(&hin, bytes, hipHostMallocDefault));
    CK(hipHostMalloc(&hout, bytes, hipHostMallocDefault));
    pg = malloc(bytes);
    memset(hin, 1, bytes); memset(hout, 0, bytes); memset(pg, 1, bytes);
    CK(hipMalloc(&din, bytes)); CK(hipMalloc(&dout, bytes));
    hipStream_t s0, s1, s2;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    auto t = [&](const char *name, auto fn) {
        fn(); CK(hipDeviceSynchronize());
        const double t0 = now();
        for (int i = 0; i < 5; i++) fn();
        CK(hipDeviceSynchronize());
        const double dt = (now() - t0) / 5;
        printf("%-58s %8.3f ms  %6.1f GB/s per direction\n", name, dt * 1e3, bytes / dt / 1e9);
    };
    t("H2D 128 MiB pinned", [&] { CK(hipMemcpyAsync(din, hin, bytes, hipMemcpyHostToDevice, s0)); });
    t("D2H 128 MiB pinned", [&] { CK(hipMemcpyAsync(hout, dout, bytes, hipMemcpyDeviceToHost, s0)); });
    t("H2D 128 MiB pageable", [&] { CK(hipMemcpyAsync(din, pg, bytes, hipMemcpyHostToDevice, s0)); });
    t("D2H 128 MiB pageable", [&] { CK(hipMemcpyAsync(pg, dout, bytes, hipMemcpyDeviceToHost, s0)); });
    t("H2D + D2H serial, one stream", [&] { CK(hipMemcpyAsync(din, hin, bytes, hipMemcpyHostToDevice, s0)); CK(hipMemcpyAsync(hout, dout, bytes, hipMemcpyDeviceToHost, s0)); });
    t("H2D || D2H, two streams", [&] { CK(hipMemcpyAsync(din, hin, bytes, hipMemcpyHostToDevice, s0)); CK(hipMemcpyAsync(hout, dout, bytes, hipMemcpyDeviceToHost, s1)); });
    t("H2D || D2H, two streams, host waits for each pair", [&] { CK(hipMemcpyAsync(din, hin, bytes, hipMemcpyHostToDevice, s0)); CK(hipMemcpyAsync(hout, dout, bytes, hipMemcpyDeviceToHost, s1)); CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1)); });
    for (size_t chunk : {1ull << 20, 4ull << 20, 16ull << 20, 32ull << 20}) {
        char nm[128];
        snprintf(nm, sizeof nm, "pipeline H2D->kernel->D2H, 3 streams, %zu MiB chunks", chunk >> 20);
        t(nm, [&] {
            hipStream_t ss[3] = {s0, s1, s2};
            size_t k = 0;
            for (size_t off = 0; off < bytes; off += chunk, k++) {
                hipStream_t s = ss[k % 3];
                CK(hipMemcpyAsync((char *)din + off, (char *)hin + off, chunk, hipMemcpyHostToDevice, s));
                hipLaunchKernelGGL(touch, dim3(1024), dim3(256), 0, s, (const float4 *)((char *)din + off), (float4 *)((char *)dout + off), chunk / 16);
                CK(hipMemcpyAsync((char *)hout + off, (char *)dout + off, chunk, hipMemcpyDeviceToHost, s));
            }
        });
    }
    // kernels that read/write pinned host memory directly (zero-copy over PCIe)
    t("kernel reads pinned host, writes pinned host (zero-copy)", [&] { hipLaunchKernelGGL(touch, dim3(2048), dim3(256), 0, s0, (const float4 *)hin, (float4 *)hout, bytes / 16); });
    t("kernel reads pinned host -> device", [&] { hipLaunchKernelGGL(touch, dim3(2048), dim3(256), 0, s0, (const float4 *)hin, (float4 *)dout, bytes / 16); });
    t("kernel device -> pinned host", [&] { hipLaunchKernelGGL(touch, dim3(2048), dim3(256), 0, s0, (const float4 *)din, (float4 *)hout, bytes / 16); });
    // ---- round 5: can the two directions be split between a kernel and a copy engine? ----
    t("kernel pinned->device (s0) || D2H copy of other data (s1)", [&] {
        hipLaunchKernelGGL(touch, dim3(2048), dim3(256), 0, s0, (const float4 *)hin, (float4 *)din, bytes / 16);
        CK(hipMemcpyAsync(hout, dout, bytes, hipMemcpyDeviceToHost, s1));
    });
    t("H2D copy (s0) || kernel device->pinned (s1)", [&] {
        CK(hipMemcpyAsync(din, hin, bytes, hipMemcpyHostToDevice, s0));
        hipLaunchKernelGGL(touch, dim3(2048), dim3(256), 0, s1, (const float4 *)dout, (float4 *)hout, bytes / 16);
    });
    t("kernel pinned->device (s0) || kernel device->pinned (s1)", [&] {
        hipLaunchKernelGGL(touch, dim3(1024), dim3(256), 0, s0, (const float4 *)hin, (float4 *)din, bytes / 16);
        hipLaunchKernelGGL(touch, dim3(1024), dim3(256), 0, s1, (const float4 *)dout, (float4 *)hout, bytes / 16);
    });
    hipEvent_t ev[512];
    for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (size_t chunk : {1ull << 20, 2ull << 20, 4ull << 20, 8ull << 20, 32ull << 20}) {
        char nm[128];
        snprintf(nm, sizeof nm, "chunks of %zu MiB: kernel pinned->device (s0), event, D2H copy (s1)", chunk >> 20);
        t(nm, [&] {
            size_t k = 0;
            for (size_t off = 0; off < bytes; off += chunk, k++) {
                hipLaunchKernelGGL(touch, dim3(1024), dim3(256), 0, s0, (const float4 *)((char *)hin + off), (float4 *)((char *)dout + off), chunk / 16);
                CK(hipEventRecord(ev[k], s0));
                CK(hipStreamWaitEvent(s1, ev[k], 0));
                CK(hipMemcpyAsync((char *)hout + off, (char *)dout + off, chunk, hipMemcpyDeviceToHost, s1));
            }
        });
        snprintf(nm, sizeof nm, "chunks of %zu MiB: H2D copy (s0), event, kernel device->pinned (s1)", chunk >> 20);
        t(nm, [&] {
            size_t k = 0;
            for (size_t off = 0; off < bytes; off += chunk, k++) {
                CK(hipMemcpyAsync((char *)din + off, (char *)hin + off, chunk, hipMemcpyHostToDevice, s0));
                CK(hipEventRecord(ev[k], s0));
                CK(hipStreamWaitEvent(s1, ev[k], 0));
                hipLaunchKernelGGL(touch, dim3(1024), dim3(256), 0, s1, (const float4 *)((char *)din + off), (float4 *)((char *)hout + off), chunk / 16);
            }
        });
        snprintf(nm, sizeof nm, "chunks of %zu MiB: H2D copy (s0), event, kernel dev->dev (s1), event, D2H copy (s2)", chunk >> 20);
        t(nm, [&] {
            size_t k = 0;
            for (size_t off = 0; off < bytes; off += chunk, k++) {
                CK(hipMemcpyAsync((char *)din + off, (char *)hin + off, chunk, hipMemcpyHostToDevice, s0));
                CK(hipEventRecord(ev[2 * k], s0));
                CK(hipStreamWaitEvent(s1, ev[2 * k], 0));
                hipLaunchKernelGGL(touch, dim3(1024), dim3(256), 0, s1, (const float4 *)((char *)din + off), (float4 *)((char *)dout + off), chunk / 16);
                CK(hipEventRecord(ev[2 * k + 1], s1));
                CK(hipStreamWaitEvent(s2, ev[2 * k + 1], 0));
                CK(hipMemcpyAsync((char *)hout + off, (char *)dout + off, chunk, hipMemcpyDeviceToHost, s2));
            }
        });
    }
    return 0;
}
