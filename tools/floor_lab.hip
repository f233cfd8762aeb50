// floor_lab.hip -- memory and issue FLOORS of the overlap-save variants that sit furthest below the HBM roof:
// the fused chain (fmchain), the decimating FIR (decim8) and the interpolating FIR (interp4).  Diagnostic, not part of the product.
//
// For each of them the harness runs a kernel that has the product kernel's SHAPE and none of its mathematics:
//   * the same launch (256 lanes, the same workgroups per CU, the same persistent grid and block walk),
//   * the same loads (sixteen -- or four -- 2 KiB rows per block through a buffer descriptor, the interior rows non-temporal) and
//     the same stores (rows of the block's valid outputs, streaming policy),
//   * the same LDS footprint (so that no more workgroups fit than the product kernel gets),
// in three rows per kernel:
//   mem    loads and stores alone: what the memory system gives this access pattern at this occupancy;
//   dose   the same, with the product kernel's VALU instruction count per wave and block (PMC, profiles/) spent between the loads
//          and the stores as INDEPENDENT packed FMAs -- no exchange, no barrier, no dependency: the issue floor on top of memory;
//   dose+x the same, with the kernel's LDS exchanges (barrier + 16 ds_write_b64 + 16 ds_read_b64 per exchange) between the doses.
// The product kernels themselves -- on the bench's data and on all-zero input (same instructions, no switching energy) -- are timed
// next to these rows by tools/floor_table.py through the product library, on the same box in the same call.
//
// Build: make -C tools floor_lab        Run: tools/floor_lab [seconds-per-row]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float cf __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}

struct Shape {
    const char *name;
    int occ;              // workgroups per CU of the product kernel
    unsigned grid;        // its persistent grid
    int rows_in;          // 2 KiB rows loaded per block (16: a 4096-sample window; 4: the interpolator's 1024-sample window)
    size_t in_step;       // input samples a block advances by
    int out_elem_bytes;   // 8 (complex_float32) or 4 (float32)
    size_t out_per_block; // output elements per block
    int group;            // blocks per pass of the walk (the batched kernels take G blocks, then store them together)
    int valu;             // VALU instructions per wave and block of the product kernel (SQ_INSTS_VALU / waves / blocks, profiles/)
    int exchanges;        // LDS exchanges (barrier + full image write + read) per block
    size_t nblocks;
    double alg_bytes;     // algorithmic bytes per launch
};

// DOSE: N independent packed FMAs per lane on the sixteen loaded values (each is one wave-level VALU instruction)
__device__ __forceinline__ void dose(cf (&v)[16], int n, cf c)
{
    for (int i = 0; i < n; i += 16) {
#pragma unroll
        for (int q = 0; q < 16; q++) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(v[q]) : "v"(c));
    }
}

template <int OCC, int ROWS, int MODE>     // MODE 0 mem, 1 dose, 2 dose + exchanges
__global__ __launch_bounds__(256, OCC) void floor_kernel(const cf *__restrict__ in, size_t in_elems, unsigned char *__restrict__ out, Shape s, float k)
{
    __shared__ cf lds[4608 + 64];        // the product kernels' image: 36,864 bytes (+ the chain's edge slots)
    const int j = threadIdx.x;
    const cf c = {k, k};
    const size_t ngroups = (s.nblocks + s.group - 1) / s.group;
    for (size_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        cf acc[16];
#pragma unroll
        for (int q = 0; q < 16; q++) acc[q] = cf{0.f, 0.f};
        for (int g = 0; g < s.group; g++) {
            const size_t b = grp * s.group + g;
            if (b >= s.nblocks) break;
            cf v[16];
#pragma unroll
            for (int q = 0; q < 16; q++) v[q] = cf{0.f, 0.f};
            const size_t first = b * s.in_step;
            const size_t left = in_elems - first;
            const size_t want = (size_t)ROWS * 256;
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(in + first, (unsigned)((left < want ? left : want) * 8));
#pragma unroll
            for (int r = 0; r < ROWS; r++) {
                const u32x2 t = (r == 0 || r == ROWS - 1) ? __builtin_amdgcn_raw_buffer_load_b64(rs, j * 8, 2048 * r, 0)
                                                          : __builtin_amdgcn_raw_buffer_load_b64(rs, j * 8, 2048 * r, 2);
                v[r] = cf{__uint_as_float(t.x), __uint_as_float(t.y)};
            }
            if (MODE >= 1) {
                const int per = s.valu / (MODE == 2 ? s.exchanges + 1 : 1);
                dose(v, per, c);
                if (MODE == 2) {
                    for (int x = 0; x < s.exchanges; x++) {
#pragma unroll
                        for (int q = 0; q < 16; q++) lds[17 * j + q] = v[q];
                        __syncthreads();
#pragma unroll
                        for (int q = 0; q < 16; q++) v[q] = lds[j + (j >> 4) + 272 * q];
                        dose(v, per, c);
                        __syncthreads();
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 16; q++) acc[q] = acc[q] + v[q];
        }
        // the group's outputs are contiguous in the output stream: rows of 256 elements, one element per lane and row
        const size_t o0 = grp * s.group * s.out_per_block;
        const size_t total = (grp * s.group + s.group <= s.nblocks ? (size_t)s.group : s.nblocks - grp * s.group) * s.out_per_block;
        const __amdgpu_buffer_rsrc_t ws = make_rsrc(out + o0 * s.out_elem_bytes, (unsigned)(total * s.out_elem_bytes));
        const int nrows = (int)((total + 255) / 256);
        for (int r = 0; r < nrows; r++) {
            const cf y = acc[r & 15];
            if (s.out_elem_bytes == 8) __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(y.x), __float_as_uint(y.y)}, ws, j * 8, 2048 * r, 2);
            else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y.x), ws, j * 4, 1024 * r, 2);
        }
    }
}

template <int OCC, int ROWS>
static double run(const Shape &s, int mode, const cf *in, size_t in_elems, unsigned char *out, double seconds, float k)
{
    auto launch = [&]() {
        if (mode == 0) hipLaunchKernelGGL((floor_kernel<OCC, ROWS, 0>), dim3(s.grid), dim3(256), 0, 0, in, in_elems, out, s, k);
        else if (mode == 1) hipLaunchKernelGGL((floor_kernel<OCC, ROWS, 1>), dim3(s.grid), dim3(256), 0, 0, in, in_elems, out, s, k);
        else hipLaunchKernelGGL((floor_kernel<OCC, ROWS, 2>), dim3(s.grid), dim3(256), 0, 0, in, in_elems, out, s, k);
    };
    for (int i = 0; i < 200; i++) launch();
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int n = 0;
    const auto t0 = std::chrono::steady_clock::now();
    CK(hipEventRecord(e0, 0));
    do {
        for (int i = 0; i < 100; i++) launch();
        n += 100;
        CK(hipStreamSynchronize(0));
    } while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / n;
}

int main(int argc, char **argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 1.0;
    const size_t C = 64u << 20;
    cf *in;
    unsigned char *out;
    CK(hipMalloc(&in, (C + 8192) * 8));
    CK(hipMalloc(&out, (C + 8192) * 8));
    {   // real data in the input: the loads' switching energy is part of the floor
        std::vector<float> h((C + 8192) * 2);
        unsigned long long z = 88172645463325252ull;
        for (auto &x : h) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; x = (float)((int)(z >> 40) - (1 << 23)) * (1.0f / (1 << 23)); }
        CK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    }
    // shapes: block geometry from the launchers (fir_ols.hip, fir_ols_decim.hip), VALU per wave and block from profiles/r03/*_rocprofv3_summary.txt
    const Shape fm{"fmchain (127 taps, 64 Mi)", 4, 1024, 16, 3968, 4, 3968, 1, 952, 3, (C + 3967) / 3968, 12.0 * C};
    const Shape dc{"decim8 (255 taps, 64 Mi in)", 3, 729, 16, 3840, 8, 480, 4, 402, 3, (C + 3839) / 3840, 9.0 * C};
    const Shape ip{"interp4 (255 taps/phase, 16 Mi in)", 3, 729, 4, 768, 8, 3072, 4, 447, 3, (C / 4 + 767) / 768, 8.0 * (C / 4) + 8.0 * C};
    const Shape fr{"fir255 (255 taps, 64 Mi) for scale", 4, 1024, 16, 3840, 8, 3840, 1, 661, 3, (C + 3839) / 3840, 16.0 * C};
    printf("%-38s %-8s %10s %8s %8s\n", "shape", "row", "ms/launch", "GB/s", "of 8 TB/s");
    auto row = [&](const Shape &s, const char *what, double ms) {
        printf("%-38s %-8s %10.4f %8.0f %8.4f\n", s.name, what, ms, s.alg_bytes / ms / 1e6, s.alg_bytes / ms / 1e6 / 8000.0);
        fflush(stdout);
    };
    const char *names[3] = {"mem", "dose", "dose+x"};
    if (argc > 2 && std::string(argv[2]) == "sweep") {
        // what an LDS exchange costs the decimator's shape, and what a fourth workgroup per CU would buy: exchanges per block 0..3 at
        // three and at four workgroups per CU (the product kernel: three exchanges' worth at three per CU, 170 VGPRs)
        for (int occ = 3; occ <= 4; occ++)
            for (int x = 0; x <= 3; x++) {
                Shape t = dc;
                t.occ = occ; t.grid = occ == 3 ? 729 : 1024; t.exchanges = x;
                char what[32];
                snprintf(what, sizeof what, "occ%d x%d", occ, x);
                const double ms = x == 0 ? (occ == 3 ? run<3, 16>(t, 1, in, C + 4096, out, seconds, 1e-3f) : run<4, 16>(t, 1, in, C + 4096, out, seconds, 1e-3f))
                                         : (occ == 3 ? run<3, 16>(t, 2, in, C + 4096, out, seconds, 1e-3f) : run<4, 16>(t, 2, in, C + 4096, out, seconds, 1e-3f));
                row(t, what, ms);
            }
        for (int occ = 3; occ <= 4; occ++)
            for (int x = 0; x <= 3; x++) {
                Shape t = ip;
                t.occ = occ; t.grid = occ == 3 ? 729 : 1024; t.exchanges = x;
                char what[32];
                snprintf(what, sizeof what, "occ%d x%d", occ, x);
                const double ms = x == 0 ? (occ == 3 ? run<3, 4>(t, 1, in, C / 4 + 4096, out, seconds, 1e-3f) : run<4, 4>(t, 1, in, C / 4 + 4096, out, seconds, 1e-3f))
                                         : (occ == 3 ? run<3, 4>(t, 2, in, C / 4 + 4096, out, seconds, 1e-3f) : run<4, 4>(t, 2, in, C / 4 + 4096, out, seconds, 1e-3f));
                row(t, what, ms);
            }
        return 0;
    }
    for (int m = 0; m < 3; m++) row(fr, names[m], run<4, 16>(fr, m, in, C + 4096, out, seconds, 1e-3f));
    for (int m = 0; m < 3; m++) row(fm, names[m], run<4, 16>(fm, m, in, C + 4096, out, seconds, 1e-3f));
    for (int m = 0; m < 3; m++) row(dc, names[m], run<3, 16>(dc, m, in, C + 4096, out, seconds, 1e-3f));
    for (int m = 0; m < 3; m++) row(ip, names[m], run<3, 4>(ip, m, in, C / 4 + 4096, out, seconds, 1e-3f));
    return 0;
}
