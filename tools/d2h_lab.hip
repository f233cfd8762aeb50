// d2h_lab -- does "kernel, then hipMemcpyAsync to PAGEABLE host memory on the same non-blocking stream, then
// hipStreamSynchronize" always deliver what the kernel wrote, when several processes share one GPU?
//
// The fuzz soak (eight pytest workers on one MI355X) showed host-pointer calls of the library whose output had holes:
// the first 8192 bytes and the last few bytes of a call's output still zero, the middle correct.  This program takes the
// library out of the picture: plain HIP, one trivial kernel.
//
//   d2h_lab [procs] [iterations] [mode]
//     mode 0  kernel -> hipMemcpyAsync(pageable) -> sync           (what pcx_*_process did for pageable buffers)
//     mode 1  kernel -> hipMemcpyAsync(pinned bounce) -> sync -> memcpy   (the library's own bounce buffer)
//     mode 2  as 0, plus a pageable H2D of the same size in front of the kernel
//   With `churn` set (4th argument) a device buffer of random size is allocated and freed every iteration, as the test
//   suite's handles do.
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); _exit(3); } \
    } while (0)

__global__ void fill(uint32_t *out, const uint32_t *in, size_t n, uint32_t tag)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = (in ? in[i] : 0u) + (tag ^ (uint32_t)i) + 1u;     // never the 0 the host buffer starts with... unless it wraps:
}

static int child(int rank, long iters, int mode, int churn)
{
    CK(hipSetDevice(0));
    uint64_t rs = 0x9e3779b97f4a7c15ull * (uint64_t)(rank + 1);
    auto rnd = [&]() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; };
    const size_t cap = 64 * 1024;      // words
    long bad = 0;
    for (long it = 0; it < iters; it++) {
        // a new stream and new device buffers every few iterations, like a new handle
        hipStream_t st;
        CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        uint32_t *din = nullptr, *dout = nullptr, *bounce = nullptr;
        const size_t n = 1 + rnd() % cap;
        CK(hipMalloc(&dout, n * 4));
        if (mode == 2) CK(hipMalloc(&din, n * 4));
        if (mode == 1) CK(hipHostMalloc(&bounce, n * 4, hipHostMallocDefault));
        void *junk = nullptr;
        if (churn) CK(hipMalloc(&junk, 4096 + rnd() % (8u << 20)));
        for (int rep = 0; rep < 3; rep++) {
            const size_t off = rnd() % 64;                       // an unaligned window of a calloc'd block, as numpy gives
            uint32_t *block = (uint32_t *)calloc(n + 64, 4);
            uint32_t *host = block + off;
            uint32_t *src = nullptr;
            const uint32_t tag = (uint32_t)rnd();
            if (mode == 2) {
                src = (uint32_t *)malloc(n * 4);
                for (size_t i = 0; i < n; i++) src[i] = (uint32_t)i * 2654435761u;
                CK(hipMemcpyAsync(din, src, n * 4, hipMemcpyHostToDevice, st));
            }
            hipLaunchKernelGGL(fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dout, din, n, tag);
            if (mode == 1) {
                CK(hipMemcpyAsync(bounce, dout, n * 4, hipMemcpyDeviceToHost, st));
                CK(hipStreamSynchronize(st));
                memcpy(host, bounce, n * 4);
            } else {
                CK(hipMemcpyAsync(host, dout, n * 4, hipMemcpyDeviceToHost, st));
                CK(hipStreamSynchronize(st));
            }
            size_t first = n, last = 0, cnt = 0, zeros = 0;
            for (size_t i = 0; i < n; i++) {
                const uint32_t want = (src ? src[i] : 0u) + (tag ^ (uint32_t)i) + 1u;
                if (host[i] != want) { if (first == n) first = i; last = i; cnt++; zeros += host[i] == 0; }
            }
            if (cnt) {
                bad++;
                if (bad <= 5)
                    printf("rank %d it %ld rep %d: %zu of %zu words wrong (%zu zero), %zu..%zu, host %% 4096 = %zu\n", rank, it, rep, cnt, n,
                           zeros, first, last, (size_t)((uintptr_t)host % 4096));
            }
            free(block);
            free(src);
        }
        if (junk) CK(hipFree(junk));
        if (bounce) CK(hipHostFree(bounce));
        if (din) CK(hipFree(din));
        CK(hipFree(dout));
        CK(hipStreamDestroy(st));
    }
    printf("rank %d: %ld bad calls of %ld\n", rank, bad, iters * 3);
    fflush(stdout);
    return bad ? 1 : 0;
}

int main(int argc, char **argv)
{
    const int procs = argc > 1 ? atoi(argv[1]) : 8;
    const long iters = argc > 2 ? atol(argv[2]) : 5000;
    const int mode = argc > 3 ? atoi(argv[3]) : 0;
    const int churn = argc > 4 ? atoi(argv[4]) : 0;
    std::vector<pid_t> kids;
    for (int r = 0; r < procs; r++) {           // fork BEFORE anything touches the GPU
        pid_t p = fork();
        if (p == 0) _exit(child(r, iters, mode, churn));
        kids.push_back(p);
    }
    int failed = 0;
    for (pid_t p : kids) {
        int stt = 0;
        waitpid(p, &stt, 0);
        failed += !(WIFEXITED(stt) && WEXITSTATUS(stt) == 0);
    }
    printf("mode %d churn %d: %d of %d processes saw a wrong copy\n", mode, churn, failed, procs);
    return 0;
}
