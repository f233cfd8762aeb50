// Why does a pass get ~9 % longer when RCCL's send/recv kernel (one workgroup, resident ~150 us) runs beside it?  The headline FIR launch,
// queued back to back on one stream, with a ONE-workgroup companion kernel per pass on a second stream that stays resident for a set time
// and does, in turn: nothing but count (alu); relaxed polling of a host word (poll); polling with a SYSTEM-scope acquire -- the buffer
// invalidate behind it drops the non-local lines of the L2s -- (acq); polling + a system-scope release fence per turn -- L2 write-back -- (rel);
// both (acqrel).  Each on data.  Usage: spin_lab [us resident, default 150]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "pcx.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define PK(x) do { int r_ = (x); if (r_ != 0) { printf("%s: %d %s\n", #x, r_, pcx_last_error()); return 1; } } while (0)

enum { ALU = 0, POLL = 1, ACQ = 2, REL = 3, ACQREL = 4, NONE = 5 };

template <int MODE> __global__ __launch_bounds__(256) void companion(volatile unsigned *host_word, unsigned *dev_word, long long ticks)
{
    __shared__ unsigned pad[5120];                        // 20 KB, as RCCL's kernel holds
    pad[threadIdx.x] = threadIdx.x;
    const long long t0 = __builtin_readcyclecounter();    // s_memtime: the shader clock on gfx950 (tools/clk_lab.hip)
    unsigned acc = 0;
    while ((long long)__builtin_readcyclecounter() - t0 < ticks) {
        if (MODE == ALU) { acc = acc * 1664525u + 1013904223u; }
        if (MODE == POLL) { acc += *host_word; }
        if (MODE == ACQ || MODE == ACQREL) { acc += __hip_atomic_load((unsigned *)host_word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM); }
        if (MODE == REL || MODE == ACQREL) {
            if (MODE == REL) acc += *host_word;
            __hip_atomic_store(dev_word + 64 + (threadIdx.x & 63), acc, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (acc == 0x12345678u) dev_word[threadIdx.x] = acc + pad[(threadIdx.x + 1) & 255];
}

__global__ void thin_companion(unsigned *o) { if (o[0] == 0x12345678u) o[1] = 1; }
// ~136 live VGPRs: 132 accumulators kept alive across a dependent chain
__global__ __launch_bounds__(256) void fat_companion(unsigned *o)
{
    float a[132];
#pragma unroll
    for (int i = 0; i < 132; i++) a[i] = (float)(o[(threadIdx.x + i) & 63] + i);
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int i = 0; i < 132; i++) a[i] = a[i] * 1.0001f + a[(i + 1) % 132];
    float t = 0;
#pragma unroll
    for (int i = 0; i < 132; i++) t += a[i];
    if (t == 12345.f) o[2] = 1;
}

// dependent loads, one lane: what a protocol kernel's chain of look-ups (arguments -> communicator -> channel -> connection -> flags) pays per hop.
// kind 0: plain loads of device memory (L2 hits after the first round); 1: system-scope acquire loads of device memory; 2: of page-locked host memory
__global__ void chase(const unsigned *table, unsigned hops, int kind, unsigned long long *ticks_out, unsigned *sink)
{
    unsigned i = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();     // 100 MHz
    for (unsigned h = 0; h < hops; h++)
        i = kind == 0 ? __builtin_nontemporal_load(table + i) : __hip_atomic_load(table + i, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    *ticks_out = t1 - t0;
    *sink = i;
}

int main(int argc, char **argv)
{
    const double us = argc > 1 ? atof(argv[1]) : 150.0;
    const size_t C = 64u << 20, K = 255;
    float *x, *y; unsigned *dw, *hw; hipStream_t s, side;
    CK(hipMalloc(&x, 8 * (C + K - 1) + 256)); CK(hipMalloc(&y, 8 * C)); CK(hipMalloc(&dw, 4096)); CK(hipHostMalloc(&hw, 4096, hipHostMallocMapped));
    memset(hw, 0, 4096);
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    float *xin = x + 2;                                    // the samples (not the history) on a 128-byte line, as stream.ShardedFir places them
    PK(pcx_fill_uniform_f32_dev(xin, 2 * (C + K - 1), 2, 0, s));
    std::vector<double> taps(2 * K);
    for (size_t i = 0; i < K; i++) { taps[2 * i] = 0.01 * (double)((i * 37) % 17) - 0.08; taps[2 * i + 1] = 0.005 * (double)((i * 11) % 13); }
    pcx_fir *h; PK(pcx_fir_create(PCX_F32, 1, 1, &h)); PK(pcx_fir_set_taps(h, taps.data(), K)); PK(pcx_fir_set_algo(h, PCX_FIR_OLS_FFT));
    size_t c, p;
    hipEvent_t e0, e1, ek; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&ek, hipEventDisableTiming));
    const long long ticks = (long long)(us * 100.0);      // s_memtime in the companion counts the 100 MHz reference?  measured below and rescaled
    const char *names[] = {"alu", "poll", "acq", "rel", "acqrel", "none"};
    const int order[] = {NONE, ALU, POLL, ACQ, REL, ACQREL, NONE};
    // how long is the companion resident?  (alone on the device)
    double tick_scale = 1.0;
    {
        CK(hipEventRecord(e0, side)); companion<ALU><<<1, 256, 0, side>>>(hw, dw, ticks); CK(hipEventRecord(e1, side)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        tick_scale = us / (ms * 1000.0);
        printf("companion with %lld ticks alone: %.1f us -> ticks scaled by %.2f\n", ticks, ms * 1000.0, tick_scale);
    }
    const long long tk = (long long)((double)ticks * tick_scale);
    for (int rep = 0; rep < 2; rep++)
        for (int oi = 0; oi < 7; oi++) {
            const int mode = order[oi];
            for (int phase = 0; phase < 2; phase++) {      // 0: settle, 1: timed
                const int n = phase ? 1500 : 600;
                if (phase) CK(hipEventRecord(e0, s));
                for (int i = 0; i < n; i++) {
                    if (mode != NONE) {
                        CK(hipEventRecord(ek, s)); CK(hipStreamWaitEvent(side, ek, 0));    // behind the previous pass, as the exchange is
                        switch (mode) {
                        case ALU: companion<ALU><<<1, 256, 0, side>>>(hw, dw, tk); break;
                        case POLL: companion<POLL><<<1, 256, 0, side>>>(hw, dw, tk); break;
                        case ACQ: companion<ACQ><<<1, 256, 0, side>>>(hw, dw, tk); break;
                        case REL: companion<REL><<<1, 256, 0, side>>>(hw, dw, tk); break;
                        default: companion<ACQREL><<<1, 256, 0, side>>>(hw, dw, tk); break;
                        }
                    }
                    PK(pcx_fir_process_dev(h, xin, C + K - 1, y, C, &c, &p, s));
                }
                if (phase) {
                    CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(side));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    printf("%-7s %.4f ms per pass\n", names[mode], ms / n);
                }
            }
        }
    // ---- when does a kernel queued BESIDE the launch get a slot?  A one-thread kernel and one whose waves hold ~136 VGPRs (what RCCL's are
    // allocated), queued on the side stream right behind the start of a pass; events say when each was done, relative to the pass ----
    {
        hipEvent_t p0, p1, c1;
        CK(hipEventCreate(&p0)); CK(hipEventCreate(&p1)); CK(hipEventCreate(&c1));
        const unsigned slot_list[] = {1024, 896, 768, 640, 512};
        for (int big = 0; big < 2; big++)
            for (unsigned si = 0; si < 5; si++) {
                PK(pcx_fir_set_slots(h, slot_list[si]));
                for (int i = 0; i < 300; i++) PK(pcx_fir_process_dev(h, xin, C + K - 1, y, C, &c, &p, s));     // settle
                double sum_c = 0, sum_p = 0; int cnt = 0;
                for (int rep = 0; rep < 30; rep++) {
                    PK(pcx_fir_process_dev(h, xin, C + K - 1, y, C, &c, &p, s));
                    CK(hipEventRecord(p0, s));
                    CK(hipStreamWaitEvent(side, p0, 0));
                    PK(pcx_fir_process_dev(h, xin, C + K - 1, y, C, &c, &p, s));       // the pass the companion is queued beside
                    CK(hipEventRecord(p1, s));
                    if (big) fat_companion<<<1, 256, 0, side>>>(dw);
                    else thin_companion<<<1, 1, 0, side>>>(dw);
                    CK(hipEventRecord(c1, side));
                    PK(pcx_fir_process_dev(h, xin, C + K - 1, y, C, &c, &p, s));
                    CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(side));
                    float mc, mp; CK(hipEventElapsedTime(&mc, p0, c1)); CK(hipEventElapsedTime(&mp, p0, p1));
                    if (rep >= 5) { sum_c += mc; sum_p += mp; cnt++; }
                }
                printf("%s beside a pass on %4u resident workgroups: done %.0f us after the pass began; the pass took %.0f us\n",
                       big ? "a 256-lane kernel of >128 VGPRs" : "a one-thread kernel           ", slot_list[si], sum_c / cnt * 1000.0, sum_p / cnt * 1000.0);
            }
        PK(pcx_fir_set_slots(h, 1024));
        // the same beside GATED launches (gate word already open): pcx_fir_process_dev_gated on 1024 workgroups
        unsigned *gw; CK(hipMalloc(&gw, 256)); CK(hipMemset(gw, 0x7f, 4)); CK(hipMemset(gw + 1, 0, 252));      // 0x7f7f7f7f >= any small pass number
        for (int big = 0; big < 2; big++) {
            int gated = 0;
            for (int i = 0; i < 300; i++) PK(pcx_fir_process_dev_gated(h, xin, C + K - 1, y, C, &c, &p, gw, 5, s, &gated));
            double sum_c = 0, sum_p = 0; int cnt = 0;
            for (int rep = 0; rep < 30; rep++) {
                PK(pcx_fir_process_dev_gated(h, xin, C + K - 1, y, C, &c, &p, gw, 5, s, &gated));
                CK(hipEventRecord(p0, s));
                CK(hipStreamWaitEvent(side, p0, 0));
                PK(pcx_fir_process_dev_gated(h, xin, C + K - 1, y, C, &c, &p, gw, 5, s, &gated));
                CK(hipEventRecord(p1, s));
                if (big) fat_companion<<<1, 256, 0, side>>>(dw);
                else thin_companion<<<1, 1, 0, side>>>(dw);
                CK(hipEventRecord(c1, side));
                PK(pcx_fir_process_dev_gated(h, xin, C + K - 1, y, C, &c, &p, gw, 5, s, &gated));
                CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(side));
                float mc, mp; CK(hipEventElapsedTime(&mc, p0, c1)); CK(hipEventElapsedTime(&mp, p0, p1));
                if (rep >= 5) { sum_c += mc; sum_p += mp; cnt++; }
            }
            printf("%s beside a GATED pass on 1024 resident workgroups (gated = %d): done %.0f us after the pass began; the pass took %.0f us\n",
                   big ? "a 256-lane kernel of >128 VGPRs" : "a one-thread kernel           ", gated, sum_c / cnt * 1000.0, sum_p / cnt * 1000.0);
        }
    }
    // ---- latency of a chain of dependent loads, alone and beside the launch ----
    {
        const unsigned hops = 64, n = 4096;
        std::vector<unsigned> tab(n);
        for (unsigned i = 0; i < n; i++) tab[i] = (i * 1031u + 577u) % n;       // a walk through the table, 16 KB apart on average
        unsigned *dt, *ht; unsigned long long *ticks;
        CK(hipMalloc(&dt, 4 * n)); CK(hipMemcpy(dt, tab.data(), 4 * n, hipMemcpyHostToDevice));
        CK(hipHostMalloc(&ht, 4 * n, hipHostMallocMapped)); memcpy(ht, tab.data(), 4 * n);
        CK(hipHostMalloc(&ticks, 64, hipHostMallocMapped));
        const char *kn[] = {"device memory, plain loads", "device memory, system-scope acquire loads", "page-locked host memory, system-scope acquire loads"};
        for (int kind = 0; kind < 3; kind++)
            for (int loaded = 0; loaded < 2; loaded++) {
                double sum = 0; int cnt = 0;
                for (int rep = 0; rep < 40; rep++) {
                    if (loaded) for (int i = 0; i < 3; i++) PK(pcx_fir_process_dev(h, xin, C + K - 1, y, C, &c, &p, s));
                    if (loaded) { CK(hipEventRecord(ek, s)); }       // (the chase starts beside the second of three launches: no wait)
                    chase<<<1, 1, 0, side>>>(kind == 2 ? ht : dt, hops, kind, ticks, dw);
                    CK(hipStreamSynchronize(side));
                    if (rep >= 8) { sum += (double)*ticks; cnt++; }
                    CK(hipStreamSynchronize(s));
                }
                printf("%-52s %s: %.2f us per dependent load\n", kn[kind], loaded ? "beside the FIR launch" : "idle device          ", sum / cnt / hops / 100.0);
            }
    }
    return 0;
}
