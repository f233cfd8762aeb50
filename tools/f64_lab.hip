// f64_lab.hip -- issue rate of the double-precision vector instructions on gfx950 (round 6): what IS the FP64 roof of a kernel
// whose arithmetic is mostly v_add_f64, as an FFT's is?   hipcc --offload-arch=gfx950 -O3 tools/f64_lab.hip -o tools/f64_lab
// Each kernel runs ITER x 16 independent instructions of one kind per wave (16 accumulators: no dependent issue), on
// `waves` waves per SIMD of every CU; reported: shader cycles per wave-instruction per SIMD (s_memtime) and TFLOP/s-equivalent.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int ITER = 4096;

template <int KIND>
__global__ __launch_bounds__(256) void k(double *out, unsigned long long *cyc, double a, double b)
{
    double r[16];
#pragma unroll
    for (int i = 0; i < 16; i++) r[i] = a * (threadIdx.x + i);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (KIND == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(r[i]) : "v"(b));
            if (KIND == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(r[i]) : "v"(b));
            if (KIND == 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r[i]) : "v"(b), "v"(a));
            if (KIND == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(*(float *)&r[i]) : "v"((float)b));
            if (KIND == 4) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(*(unsigned *)&r[i]) : "v"((unsigned)threadIdx.x));
            if (KIND == 5) {   // the FFT's mix: 3 adds, 1 mul, 1 fma
                if (i % 5 < 3) asm volatile("v_add_f64 %0, %0, %1" : "+v"(r[i]) : "v"(b));
                else if (i % 5 == 3) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(r[i]) : "v"(b));
                else asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r[i]) : "v"(b), "v"(a));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += r[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// dependency distance: the same number of v_add_f64 / v_fma_f64, but only NACC independent chains (an instruction depends on the one
// NACC instructions before it): how much instruction-level parallelism does the FP64 pipe need from ONE wave?
template <int NACC, int FMA>
__global__ __launch_bounds__(256) void kdep(double *out, unsigned long long *cyc, double a, double b)
{
    double r[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) r[i] = a * (threadIdx.x + i);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int j = 0; j < 16 / NACC; j++)
#pragma unroll
            for (int i = 0; i < NACC; i++) {
                if (FMA) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r[i]) : "v"(b), "v"(a));
                else asm volatile("v_add_f64 %0, %0, %1" : "+v"(r[i]) : "v"(b));
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += r[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC, int FMA>
void rundep(int wgs_per_cu, double *out, unsigned long long *cyc)
{
    const int grid = 256 * wgs_per_cu;
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL((kdep<NACC, FMA>), dim3(grid), dim3(256), 0, 0, out, cyc, 1.0000001, 0.9999999);
        CHECK(hipDeviceSynchronize());
    }
    std::vector<unsigned long long> h(grid);
    CHECK(hipMemcpy(h.data(), cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double mean = 0;
    for (auto c : h) mean += (double)c;
    mean /= grid;
    printf("%s, %2d independent chain(s) per wave, %d wave(s)/SIMD: %.2f clk per wave-instruction (one wave's view), %.2f per instruction per SIMD\n",
           FMA ? "fma_f64" : "add_f64", NACC, wgs_per_cu, mean / (ITER * 16.0), mean / (ITER * 16.0) / wgs_per_cu);
}

template <int KIND>
void run(const char *name, int wgs_per_cu, double *out, unsigned long long *cyc)
{
    const int grid = 256 * wgs_per_cu;     // 4 waves per workgroup: wgs_per_cu waves per SIMD
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, out, cyc, 1.0000001, 0.9999999);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
    }
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(grid);
    CHECK(hipMemcpy(h.data(), cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double mean = 0;
    for (auto c : h) mean += (double)c;
    mean /= grid;
    const double insts = (double)ITER * 16;
    // cycles per wave-instruction as seen by ONE wave; per SIMD: divide by the waves sharing it
    printf("%-10s %d wave(s)/SIMD: %.3f ms   %.2f clk per wave-instruction (one wave's view)   %.2f clk per instruction per SIMD   %.1f G wave-inst/s\n",
           name, wgs_per_cu, ms, mean / insts, mean / insts / wgs_per_cu, insts * grid * 4 / (ms * 1e-3) / 1e9);
}

int main()
{
    double *out; unsigned long long *cyc;
    CHECK(hipMalloc(&out, 256 * 8 * 256 * sizeof(double)));
    CHECK(hipMalloc(&cyc, 256 * 8 * sizeof(unsigned long long)));
    for (int w = 1; w <= 4; w *= 2) {
        run<0>("add_f64", w, out, cyc);
        run<1>("mul_f64", w, out, cyc);
        run<2>("fma_f64", w, out, cyc);
        run<5>("fft mix", w, out, cyc);
        run<3>("add_f32", w, out, cyc);
        run<4>("xor_b32", w, out, cyc);
    }
    for (int w = 1; w <= 2; w++) {
        rundep<1, 0>(w, out, cyc); rundep<2, 0>(w, out, cyc); rundep<4, 0>(w, out, cyc); rundep<8, 0>(w, out, cyc); rundep<16, 0>(w, out, cyc);
        rundep<1, 1>(w, out, cyc); rundep<2, 1>(w, out, cyc); rundep<4, 1>(w, out, cyc); rundep<8, 1>(w, out, cyc); rundep<16, 1>(w, out, cyc);
    }
    return 0;
}
