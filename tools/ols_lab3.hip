// ols_lab3.hip -- round-3 diagnostic harness for the headline overlap-save FIR (NOT part of the product library).
//
// Round 2 left fir_cf32_ols4096_kernel at 0.648 of the 8 TB/s roof with three leads unexplored:
//   (a) 20 % of the LDS-active cycles are bank conflicts although fft4096.hpp called its padding conflict-free
//   (b) the 1024-workgroup persistent shape streams 5-10 % below a many-workgroup copy
//   (c) every exchange goes through LDS
// This harness runs the product's block pipeline with controlled departures, each for seconds back to back, and
// prints time per launch, algorithmic TB/s, in-kernel clock and -- for the STAMP builds -- where a workgroup's time
// goes (load wait / forward / inverse / store issue), how long the first draw takes and how ragged the launch's end is.
//
// FLAGS of the kernel template:
//   SWZ     the LDS image XOR-swizzled instead of padded: s(e) = (e & ~15) | ((e ^ (e >> 4)) & 15) -- every ds_read_b64
//           (32-lane groups over 64 banks) and ds_write_b64 (16-lane groups over 32 banks) of the three passes conflict-free,
//           image exactly 32 KiB.  (pad(i) = i + i/16 puts 33 elements under a 32-lane read: lane 31 lands on lane 0's banks,
//           two cycles per group instead of one -- the 20 % the counter shows.)
//   FIRST   the first chunk of a workgroup is its blockIdx (no atomic): 1024 simultaneous first draws on one word take
//           ~12 us to drain at 88 draws/us
//   STAMP   s_memtime stamps around the phases (diagnostic; costs a few %)
//   MEM     no transforms (the memory floor of the dealt shape)       DOSE  640 packed FMAs per lane instead of the transforms
//   NOPRIO  no s_setprio on the second half
//   LBAR    the pipeline's barriers order LDS only (s_waitcnt lgkmcnt(0); s_barrier) -- __syncthreads() also drains vmcnt, i.e.
//           the first barrier of a block waits for the dealer's atomic
//   ADRAW   the dealer's atomic stays in flight until its value is published (the product's atomicAdd is rewritten by the
//           compiler's atomic optimizer into add + s_waitcnt vmcnt(0) + v_readfirstlane: wave 0 stalls on the whole round trip,
//           behind the 16 loads, in every block that draws).  This file is built with the optimizer off; builds WITHOUT this
//           flag put the product's wait back by hand.
//   SHARD   sixteen dealer counters, a cache line apart, instead of one word: workgroup w draws from counter (w >> 3) & 15, whose
//           n-th draw is chunk first + 16 n + g -- a single word serves ~88 draws/us, exactly the rate of one draw per BLOCK at the
//           headline speed, which is why r02 had to deal pairs and the launch's end is ragged by two blocks (mean idle 12 us of 200).
//           One steal from the neighbouring counter when the own one runs dry (NOSTEAL: none).
//   DMA     the next block's input goes global -> LDS by LDS-DMA, issued BEFORE this block's stores (in-order vmcnt: the loads
//           no longer wait for the stores' acknowledgements, and their latency overlaps the last butterflies + store issue)
//   XCH     the transform pair of the product since late round 3 (fft4096.hpp dif_a_math / dif_rest / dit_back): forward decimation in
//           frequency, inverse its transpose, the second exchange of each inside sixteen lanes -- three barriers per block, no conflicts,
//           H turned across the lanes through LDS once per workgroup.  (The product also requests its first block ahead of the tables.)
//
// Build: make -C tools ols_lab3      Run: tools/ols_lab3 [seconds-per-config] [first-config] [last-config]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../pothoscomms_amd/csrc/fft4096.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

using namespace pcx::fft4k;

enum { F_SWZ = 1, F_FIRST = 2, F_STAMP = 4, F_MEM = 8, F_DOSE = 16, F_DMA = 32, F_NOPRIO = 64, F_STATIC = 128, F_SINGLES = 256, F_LBAR = 512, F_ADRAW = 1024, F_SHARD = 2048, F_NOSTEAL = 4096, F_XCH = 8192 };

struct WgStat {   // per workgroup, written once at exit
    unsigned long long t_start, t_first, t_end;      // s_memrealtime (100 MHz)
    unsigned long long c_start, c_end;               // s_memtime (shader clock)
    unsigned long long load, fwd, inv, st, iss;      // summed shader cycles per phase (STAMP builds)
    unsigned blocks, pad;
};

struct Sched { unsigned ctr[16 * 32]; unsigned finished; };   // counters 128 B apart

// ---- the three passes on either layout ----
template <bool SWZ>
__device__ __forceinline__ void p1_write(const cf (&v)[16], cf *lds, int j)
{
    if (SWZ) {
        int a = (16 * j + (j & 15)) * 8;     // byte offset; one v_xor per store, the rest of the address is the instruction's offset field
        asm volatile("" : "+v"(a));          // not loop-invariant for the compiler: the 16 XORed addresses are recomputed, not held in registers
#pragma unroll
        for (int q = 0; q < 16; q++) *reinterpret_cast<cf *>(reinterpret_cast<char *>(lds) + (a ^ (bin_of(q) * 8))) = v[q];
    } else {
#pragma unroll
        for (int q = 0; q < 16; q++) lds[17 * j + bin_of(q)] = v[q];
    }
}
template <bool SWZ>
__device__ __forceinline__ void p23_read(cf (&v)[16], const cf *lds, int j)
{
    if (SWZ) {
        const int rb = (j & 0xF0) | ((j ^ (j >> 4)) & 15);
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] = lds[rb + 256 * r];
    } else {
        const int rb = j + (j >> 4);
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] = lds[rb + 272 * r];
    }
}
template <bool SWZ>
__device__ __forceinline__ void p2_write(const cf (&v)[16], cf *lds, int j)
{
    if (SWZ) {
        int b = ((j >> 4) * 256 + (j & 15)) * 8;
        asm volatile("" : "+v"(b));
#pragma unroll
        for (int q = 0; q < 16; q++) *reinterpret_cast<cf *>(reinterpret_cast<char *>(lds) + (b ^ (bin_of(q) * 8)) + 128 * bin_of(q)) = v[q];
    } else {
        const int wb = (j >> 4) * 272 + (j & 15);
#pragma unroll
        for (int q = 0; q < 16; q++) lds[wb + 17 * bin_of(q)] = v[q];
    }
}
template <bool SWZ>
__device__ __forceinline__ void load_tw2(LaneTw &tw, const cf *lds, int j)
{
    const cf *t2 = lds + (SWZ ? N : LDS_DATA) + (j & 15);
#pragma unroll
    for (int p = 0; p < 3; p++) tw.a[p] = t2[p * 16];
#pragma unroll
    for (int p = 0; p < 12; p++) tw.c[p] = t2[(3 + p) * 16];
}
template <bool LBAR>
__device__ __forceinline__ void wg_barrier()
{
    if (LBAR) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    } else {
        __syncthreads();
    }
}
struct NoHook { __device__ __forceinline__ void operator()() const {} };
template <bool SWZ, bool LBAR, typename HOOK = NoHook>
__device__ __forceinline__ void transform(cf (&v)[16], cf *lds, int j, const LaneTw &tw3, HOOK hook = HOOK())
{
    fft16_plain(v);
    hook();                 // (the dealer's draw: every load of the block has been consumed, nothing else is in flight)
    wg_barrier<LBAR>();
    p1_write<SWZ>(v, lds, j);
    wg_barrier<LBAR>();
    p23_read<SWZ>(v, lds, j);
    LaneTw tw;
    load_tw2<SWZ>(tw, lds, j);
    fft16_tw(v, tw);
    wg_barrier<LBAR>();
    p2_write<SWZ>(v, lds, j);
    wg_barrier<LBAR>();
    p23_read<SWZ>(v, lds, j);
    fft16_tw(v, tw3);
}

#define STAMP_NOW() (__builtin_amdgcn_s_memtime())

template <int FLAGS>
__global__ __launch_bounds__(256, 4) void lab3_kernel(const float2 *__restrict__ in, float2 *__restrict__ out, size_t n_out,
                                                      const float2 *__restrict__ Hspec, int Kov, const float2 *__restrict__ twtab,
                                                      size_t nblocks, Sched *__restrict__ sched, WgStat *__restrict__ stats)
{
    constexpr bool SWZ = (FLAGS & F_SWZ) != 0, FIRST = (FLAGS & F_FIRST) != 0, STAMP = (FLAGS & F_STAMP) != 0;
    constexpr bool MEM = (FLAGS & F_MEM) != 0, DOSE = (FLAGS & F_DOSE) != 0, DMA = (FLAGS & F_DMA) != 0, PRIO = (FLAGS & F_NOPRIO) == 0;
    constexpr bool STATIC = (FLAGS & F_STATIC) != 0, SINGLES = (FLAGS & F_SINGLES) != 0;
    constexpr bool LBAR = (FLAGS & F_LBAR) != 0 || DMA, ADRAW = (FLAGS & F_ADRAW) != 0;
    constexpr unsigned NSH = (FLAGS & F_SHARD) ? 16u : 1u;
    constexpr bool STEAL = NSH > 1 && (FLAGS & F_NOSTEAL) == 0;
    constexpr bool XCH = (FLAGS & F_XCH) != 0;
    static_assert(!XCH || (!SWZ && !DMA && LBAR && ADRAW), "XCH: on the padded image, the product's barriers and draw");
    __shared__ cf lds[(SWZ ? N : LDS_DATA) + LDS_TW2];
    __shared__ unsigned slots[2];
    const int j = threadIdx.x;
    const size_t S = (size_t)(N - Kov);
    WgStat ws;
    ws.t_start = __builtin_amdgcn_s_memrealtime();
    ws.c_start = STAMP_NOW();
    ws.load = ws.fwd = ws.inv = ws.st = ws.iss = 0;
    unsigned long long p_prev = 0;
    ws.blocks = 0; ws.pad = 0;
    // ---- dealer: strided pairs (q, q + nchunks), one atomic per pair ----
    const unsigned csz = SINGLES ? 1u : 2u;
    const unsigned nchunks = (unsigned)((nblocks + csz - 1) / csz);
    unsigned chunk, sub = 0, pending = 0;
    unsigned shard = NSH > 1 ? (blockIdx.x >> 3) & (NSH - 1) : 0u;      // this workgroup's counter
    bool stolen = false;
    const unsigned first_dyn = FIRST ? gridDim.x : 0u;                     // chunks below this are the static first ones
    auto chunk_of_draw = [&](unsigned n, unsigned g) -> unsigned { return first_dyn + n * NSH + g; };
    if (STATIC) {
        chunk = blockIdx.x;
    } else if (FIRST) {
        chunk = blockIdx.x;
    } else {
        if (j == 0) slots[0] = chunk_of_draw(atomicAdd(&sched->ctr[shard * 32], 1u), shard);
        __syncthreads();
        chunk = __builtin_amdgcn_readfirstlane(slots[0]);
    }
    auto finish = [&]() {
        if (!STATIC && j == 0 && atomicAdd(&sched->finished, 1u) == gridDim.x - 1) {
            for (unsigned g = 0; g < NSH; g++) atomicExch(&sched->ctr[g * 32], 0u);
            atomicExch(&sched->finished, 0u);
        }
    };
    auto write_stats = [&]() {
        if (j == 0) {
            ws.t_end = __builtin_amdgcn_s_memrealtime();
            ws.c_end = STAMP_NOW();
            stats[blockIdx.x] = ws;
        }
    };
    if (chunk >= nchunks) { finish(); ws.t_first = ws.t_start; write_stats(); return; }
    ws.t_first = __builtin_amdgcn_s_memrealtime();
    auto block_of = [&](unsigned c, unsigned s) -> size_t { return STATIC ? (size_t)c : (size_t)c + (size_t)s * nchunks; };
    auto last_of_chunk = [&]() -> bool { return STATIC || sub + 1 >= csz || block_of(chunk, sub + 1) >= nblocks; };

    LaneTw tw3;
    load_pass3_twiddles(tw3, twtab, j);
    if (j < LDS_TW2) lds[(SWZ ? N : LDS_DATA) + j] = reinterpret_cast<const cf *>(twtab)[j];
    cf H[16];
    if (XCH) {
        load_spectrum_lanes(H, Hspec, twtab, lds, j);
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) H[k] = reinterpret_cast<const cf *>(Hspec)[j + 256 * k];
    }
    cf acc[16];
    if (DOSE) {
#pragma unroll
        for (int q = 0; q < 16; q++) acc[q] = cf{0.f, 0.f};
    }
    auto fetch = [&](cf (&dst)[16], size_t blk) {
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(in + blk * S, N * 8);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const u32x2 t = (r < 1 || r >= 15) ? __builtin_amdgcn_raw_buffer_load_b64(rs, j * 8, 2048 * r, 0)
                                               : __builtin_amdgcn_raw_buffer_load_b64(rs, j * 8, 2048 * r, 2);
            dst[r] = cf{__uint_as_float(t.x), __uint_as_float(t.y)};
        }
    };
    // LDS-DMA of one block's 32 KiB into lds[0..4095] (linear): wave w moves the 1 KiB chunks w, 4 + w, ... 28 + w, 16 B per lane.
    // The first and the last 4 KiB are shared with the neighbouring blocks' windows (cached normally), the rest is read once (nt).
    const int wv = __builtin_amdgcn_readfirstlane(j >> 6);
    auto issue_dma = [&](size_t blk) {
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(in + blk * S, N * 8);
        const int vo = (j & 63) * 16;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int c = i * 4 + wv;
            __attribute__((address_space(3))) void *dst = (__attribute__((address_space(3))) void *)(lds + c * 128);
            if (i == 0 || i == 7) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, vo, c * 1024, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, vo, c * 1024, 0, 2);
        }
    };
    size_t b = block_of(chunk, sub);
    if (DMA) issue_dma(b);
    for (;;) {
        unsigned long long p0 = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0;
        cf v[16];
        unsigned *slot = &slots[(ws.blocks + 1) & 1];          // two slots: a slow wave may still be reading the previous block's
        if (DMA) {
            if (STAMP) p0 = STAMP_NOW();
            if (ws.blocks == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // the 16 stores are younger than the DMA
            wg_barrier<true>();                               // every wave's DMA has landed
            if (STAMP) p1 = STAMP_NOW();
#pragma unroll
            for (int r = 0; r < 16; r++) v[r] = lds[j + 256 * r];
        } else {
            fetch(v, b);
            if (STAMP) { p0 = STAMP_NOW(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); p1 = STAMP_NOW(); }
        }
        auto draw = [&]() {
            if (!STATIC && j == 0 && last_of_chunk()) {
                pending = atomicAdd(&sched->ctr[shard * 32], 1u);
                if (!ADRAW) asm volatile("s_waitcnt vmcnt(0)" : "+v"(pending) :: "memory");     // what the product's optimised atomic does
            }
        };
        // r02: right behind the loads.  ADRAW: behind the first butterflies, when no load is outstanding any more -- a lane-0-only
        // atomic between the loads and their use makes the compiler wait for it with the last load (it cannot know whether the
        // branch issued it), which is the stall this flag removes
        if (!ADRAW || MEM || DOSE) draw();
        cf u[16];
        bool has_next = false;
        size_t nb = 0;
        if (MEM || DOSE) {
#pragma unroll
            for (int q = 0; q < 16; q++) u[q] = v[q];
            if (DOSE) {
#pragma unroll 1
                for (int it = 0; it < 40; it++) {
#pragma unroll
                    for (int q = 0; q < 16; q++)
                        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[q]) : "v"(v[q]), "v"(v[(q + 5) & 15]));
                }
                if (acc[0].x == 123456.789f) {
#pragma unroll
                    for (int q = 0; q < 16; q++) u[q] = u[q] + acc[q];
                }
            }
            if (!STATIC && j == 0 && last_of_chunk()) *slot = chunk_of_draw(pending, shard);
            wg_barrier<LBAR>();
        } else {
            if (XCH) { dif_a_math(v, tw3); draw(); dif_rest(v, lds, j); }
            else if (ADRAW) transform<SWZ, LBAR>(v, lds, j, tw3, draw);
            else transform<SWZ, LBAR>(v, lds, j, tw3);
#pragma unroll
            for (int q = 0; q < 16; q += 2) {
                const int k0 = bin_of(q), k1 = bin_of(q + 1);
                u[k0] = v[q];
                u[k1] = v[q + 1];
                cmul2_conj(u[k0], u[k1], H[k0], H[k1]);
            }
            if (!STATIC && j == 0 && last_of_chunk()) *slot = chunk_of_draw(pending, shard);
            if (STAMP) p2 = STAMP_NOW();
            if (PRIO) __builtin_amdgcn_s_setprio(1);
            if (XCH) {
                dit_back<false>(u, lds, j, tw3);
            } else if (!DMA) {
                transform<SWZ, LBAR>(u, lds, j, tw3);
            } else {
                // the same transform with the next block's DMA slipped in between the last gather and the last butterflies
                fft16_plain(u);
                wg_barrier<true>();
                p1_write<SWZ>(u, lds, j);
                wg_barrier<true>();
                p23_read<SWZ>(u, lds, j);
                LaneTw tw;
                load_tw2<SWZ>(tw, lds, j);
                fft16_tw(u, tw);
                wg_barrier<true>();
                p2_write<SWZ>(u, lds, j);
                wg_barrier<true>();
                p23_read<SWZ>(u, lds, j);
                // which block comes next (the draw was published in front of the barriers above)
                if (!last_of_chunk()) { has_next = true; nb = block_of(chunk, sub + 1); }
                else if (!STATIC) { const unsigned c2 = __builtin_amdgcn_readfirstlane(*slot); has_next = c2 < nchunks; nb = block_of(c2, 0); }
                else { nb = b + gridDim.x; has_next = nb < nblocks; }
                wg_barrier<true>();                               // every wave has read its last-pass operands: the image is free
                if (has_next) issue_dma(nb);
                fft16_tw(u, tw3);
            }
        }
        if (STAMP) p3 = STAMP_NOW();
        const size_t room = n_out - b * S;
        const __amdgpu_buffer_rsrc_t wsr = make_rsrc(out + b * S, (unsigned)((room < S ? room : S) * 8));
        const unsigned vbase = (unsigned)(j - Kov) * 8u;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int row = 256 * bin_of(q);
            if (!DMA && row + 255 < Kov) continue;            // DMA build: always 16 stores, so that vmcnt(16) below means "the DMA"
            store_cf<2>(wsr, vbase + (unsigned)row * 8u, cf{u[q].x, -u[q].y});
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        if (STAMP) {
            p4 = STAMP_NOW();
            ws.load += p1 - p0; ws.fwd += p2 - p1; ws.inv += p3 - p2; ws.st += p4 - p3;
            if (p_prev) ws.iss += p0 - p_prev;
            p_prev = p4;
        }
        ws.blocks++;
        if (DMA) {
            if (!has_next) break;
            if (!last_of_chunk()) sub++;
            else { chunk = __builtin_amdgcn_readfirstlane(*slot); sub = 0; }
            b = nb;
            continue;
        }
        if (STATIC) {
            b += gridDim.x;
            if (b >= nblocks) break;
            continue;
        }
        if (!last_of_chunk()) { sub++; }
        else {
            chunk = __builtin_amdgcn_readfirstlane(*slot);
            sub = 0;
            if (STEAL && chunk >= nchunks && !stolen) {
                // the own counter has run dry: one synchronous draw from the neighbour's, then stay there
                stolen = true;
                shard = (shard + 1) & (NSH - 1);
                if (j == 0) slots[0] = slots[1] = chunk_of_draw(atomicAdd(&sched->ctr[shard * 32], 1u), shard);
                __syncthreads();
                chunk = __builtin_amdgcn_readfirstlane(slots[0]);
                __syncthreads();
            }
            if (chunk >= nchunks) break;
        }
        b = block_of(chunk, sub);
    }
    finish();
    write_stats();
}

__global__ void fill_kernel(float *p, size_t n, unsigned long long seed, int zero)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = (i + seed) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        p[i] = zero ? 0.f : (float)((double)(z >> 40) / 8388608.0 - 1.0);
    }
}

static std::vector<float> make_tw4096()
{
    std::vector<float> t(2 * (15 * 16 + 15 * 256));
    const double two_pi = 6.283185307179586476925286766559;
    auto angle = [&](int p, double base) {
        if (p < 3) return base * 4.0 * (p + 1);
        const int n2 = (p - 3) / 4 + 1, k1 = (p - 3) % 4;
        return base * n2 + (double)(n2 * k1) / 16.0;
    };
    for (int p = 0; p < 15; p++) {
        for (int kk = 0; kk < 16; kk++) {
            const double a = -two_pi * angle(p, (double)kk / 256.0);
            t[2 * (p * 16 + kk)] = (float)std::cos(a);
            t[2 * (p * 16 + kk) + 1] = (float)std::sin(a);
        }
        for (int j = 0; j < 256; j++) {
            const double a = -two_pi * angle(p, (double)j / 4096.0);
            t[2 * (240 + p * 256 + j)] = (float)std::cos(a);
            t[2 * (240 + p * 256 + j) + 1] = (float)std::sin(a);
        }
    }
    return t;
}

static double median(std::vector<double> v)
{
    if (v.empty()) return 0;
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

typedef void (*KernFn)(const float2 *, float2 *, size_t, const float2 *, int, const float2 *, size_t, Sched *, WgStat *);
struct Cfg { const char *name; KernFn k; unsigned grid; bool correct; bool stamp; bool zero; };

int main(int argc, char **argv)
{
    const double secs = argc > 1 ? atof(argv[1]) : 3.0;
    const int c_first = argc > 2 ? atoi(argv[2]) : 0, c_last = argc > 3 ? atoi(argv[3]) : 1000;
    const size_t n = 64ull << 20;
    const int Kov = 256;
    const size_t S = 4096 - Kov;
    const size_t nblocks = (n + S - 1) / S;
    const size_t in_elems = nblocks * S + 4096;
    float2 *x, *y, *yref, *Hs, *tw;
    WgStat *st;
    Sched *sched;
    CK(hipMalloc(&x, in_elems * 8));
    CK(hipMalloc(&y, (nblocks * S + 64) * 8));
    CK(hipMalloc(&yref, (nblocks * S + 64) * 8));
    CK(hipMalloc(&Hs, 4096 * 8));
    CK(hipMalloc(&st, 32768 * sizeof(WgStat)));
    CK(hipMalloc(&sched, sizeof(Sched)));
    CK(hipMemset(sched, 0, sizeof(Sched)));
    std::vector<float> t = make_tw4096();
    CK(hipMalloc(&tw, t.size() * 4));
    CK(hipMemcpy(tw, t.data(), t.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, (float *)Hs, (size_t)8192, 77ull, 0);
    hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, (float *)x, in_elems * 2, 2ull, 0);
    const unsigned g971 = [&] { const size_t slots = 1024, rounds = (nblocks + slots - 1) / slots; return (unsigned)((nblocks + rounds - 1) / rounds); }();
    constexpr int B = F_FIRST;                                   // keeps the padded image: the swizzle's 64 extra VALU cost more than the conflicts
    constexpr int A = F_FIRST | F_ADRAW | F_LBAR;
    const Cfg cfgs[] = {
        /* 0 */ {"r02 product: padded, atomic first draw, prio", lab3_kernel<0>, 1024, true, false, false},
        /* 1 */ {"  + static first chunk", lab3_kernel<B>, 1024, true, false, false},
        /* 2 */ {"  + draw in flight, LDS-only barriers", lab3_kernel<A>, 1024, true, false, false},
        /* 3 */ {"  + 16 counters, still pairs", lab3_kernel<A | F_SHARD>, 1024, true, false, false},
        /* 4 */ {"  + 16 counters, SINGLE blocks", lab3_kernel<A | F_SHARD | F_SINGLES>, 1024, true, false, false},
        /* 5 */ {"    same, no stealing", lab3_kernel<A | F_SHARD | F_SINGLES | F_NOSTEAL>, 1024, true, false, false},
        /* 6 */ {"    singles, 16 counters, draw waited for at once", lab3_kernel<B | F_SHARD | F_SINGLES>, 1024, true, false, false},
        /* 7 */ {"    singles, swizzled image", lab3_kernel<A | F_SHARD | F_SINGLES | F_SWZ>, 1024, true, false, false},
        /* 8 */ {"    singles, no priority", lab3_kernel<A | F_SHARD | F_SINGLES | F_NOPRIO>, 1024, true, false, false},
        /* 9 */ {"    singles, 768 workgroups (3 per CU)", lab3_kernel<A | F_SHARD | F_SINGLES>, 768, true, false, false},
        /* 10 */ {"    singles + LDS-DMA prefetch", lab3_kernel<A | F_SHARD | F_SINGLES | F_DMA>, 1024, true, false, false},
        /* 11 */ {"dealt (16 counters, singles) loads + stores only", lab3_kernel<A | F_SHARD | F_SINGLES | F_MEM>, 1024, false, false, false},
        /* 12 */ {"  + 640 packed FMAs per lane", lab3_kernel<A | F_SHARD | F_SINGLES | F_DOSE>, 1024, false, false, false},
        /* 13 */ {"STAMPED r02 product", lab3_kernel<F_STAMP>, 1024, true, true, false},
        /* 14 */ {"STAMPED singles, 16 counters", lab3_kernel<A | F_SHARD | F_SINGLES | F_STAMP>, 1024, true, true, false},
        /* 15 */ {"r02 product (again)", lab3_kernel<0>, 1024, true, false, false},
        /* 16 */ {"static first chunk only (again)", lab3_kernel<B>, 1024, true, false, false},
        /* 17 */ {"singles, 16 counters (again)", lab3_kernel<A | F_SHARD | F_SINGLES>, 1024, true, false, false},
        /* 18 */ {"singles, 16 counters, all-zero input", lab3_kernel<A | F_SHARD | F_SINGLES>, 1024, true, false, true},
        /* 19 */ {"singles, no stealing, SIXTEEN-LANE EXCHANGE", lab3_kernel<A | F_SHARD | F_SINGLES | F_NOSTEAL | F_XCH>, 1024, true, false, false},
        /* 20 */ {"STAMPED sixteen-lane exchange", lab3_kernel<A | F_SHARD | F_SINGLES | F_NOSTEAL | F_XCH | F_STAMP>, 1024, true, true, false},
        /* 21 */ {"singles, no stealing (again)", lab3_kernel<A | F_SHARD | F_SINGLES | F_NOSTEAL>, 1024, true, false, false},
        /* 22 */ {"sixteen-lane exchange (again)", lab3_kernel<A | F_SHARD | F_SINGLES | F_NOSTEAL | F_XCH>, 1024, true, false, false},
        /* 23 */ {"sixteen-lane exchange, all-zero input", lab3_kernel<A | F_SHARD | F_SINGLES | F_NOSTEAL | F_XCH>, 1024, true, false, true},
    };
    // reference output: the r01 static-stride pipeline
    hipLaunchKernelGGL(lab3_kernel<F_STATIC>, dim3(g971), dim3(256), 0, 0, x, yref, n, Hs, Kov, tw, nblocks, sched, st);
    CK(hipDeviceSynchronize());
    const size_t cmp = 4u << 20;
    std::vector<float> ref(2 * cmp), got(2 * cmp), tailref(2 * 65536), tailgot(2 * 65536);
    CK(hipMemcpy(ref.data(), yref, cmp * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(tailref.data(), yref + (n - 65536), 65536 * 8, hipMemcpyDeviceToHost));
    printf("# 255-tap geometry: Kov=%d S=%zu blocks=%zu, %zu samples, %.1f s per configuration\n", Kov, S, nblocks, n, secs);
    printf("%-3s %-50s %9s %7s %6s %8s | per-block us: %6s %6s %6s %6s %6s | %8s %8s %8s\n", "#", "config", "ms/launch", "TB/s", "frac", "clk(GHz)",
           "issue", "load", "fwd", "inv", "store", "first_us", "tail_us", "span_us");
    bool cur_zero = false;
    int idx = -1;
    for (const Cfg &c : cfgs) {
        idx++;
        if (idx < c_first || idx > c_last) continue;
        if (c.zero != cur_zero) {
            hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, (float *)x, in_elems * 2, 2ull, c.zero ? 1 : 0);
            cur_zero = c.zero;
        }
        CK(hipDeviceSynchronize());
        CK(hipMemset(sched, 0, sizeof(Sched)));
        auto launch = [&] { hipLaunchKernelGGL(c.k, dim3(c.grid), dim3(256), 0, 0, x, y, n, Hs, Kov, tw, nblocks, sched, st); };
        // parity first (a fresh output buffer)
        double rel = -1;
        if (c.correct && !c.zero) {
            CK(hipMemset(y, 0xff, (nblocks * S + 64) * 8));
            launch();
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(got.data(), y, cmp * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(tailgot.data(), y + (n - 65536), 65536 * 8, hipMemcpyDeviceToHost));
            double mx = 0, md = 0;
            for (size_t i = 0; i < 2 * cmp; i++) { mx = std::max(mx, (double)std::fabs(ref[i])); const double d = std::fabs((double)ref[i] - got[i]); md = (d == d && d > md) ? d : (d != d ? 1e30 : md); }
            for (size_t i = 0; i < 2 * 65536; i++) { const double d = std::fabs((double)tailref[i] - tailgot[i]); md = (d == d && d > md) ? d : (d != d ? 1e30 : md); }
            rel = md / mx;
        }
        const auto w0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count() < secs * 0.5) {
            for (int i = 0; i < 200; i++) launch();
            CK(hipDeviceSynchronize());
        }
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        size_t iters = 0;
        CK(hipEventRecord(e0, 0));
        const auto w1 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - w1).count() < secs * 0.5) {
            for (int i = 0; i < 200; i++) launch();
            iters += 200;
            CK(hipStreamSynchronize(0));
        }
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double per = ms / (double)iters;
        std::vector<WgStat> hs(c.grid);
        CK(hipMemcpy(hs.data(), st, c.grid * sizeof(WgStat), hipMemcpyDeviceToHost));
        std::vector<double> clk, first;
        unsigned long long t_min = ~0ull, t_max = 0;
        double load = 0, fwd = 0, inv = 0, sto = 0, iss = 0, blocks = 0;
        for (const WgStat &s : hs) {
            if (s.t_end > s.t_start) clk.push_back((double)(s.c_end - s.c_start) / (double)(s.t_end - s.t_start) * 0.1);
            first.push_back((double)(s.t_first - s.t_start) * 0.01);
            t_min = std::min(t_min, s.t_start); t_max = std::max(t_max, s.t_end);
            iss += (double)s.iss; load += (double)s.load; fwd += (double)s.fwd; inv += (double)s.inv; sto += (double)s.st; blocks += s.blocks;
        }
        double tail = 0;
        for (const WgStat &s : hs) tail += (double)(t_max - s.t_end) * 0.01;
        tail /= (double)hs.size();
        const double ghz = median(clk);
        const double cyc2us = ghz > 0 ? 1.0 / (ghz * 1e3) : 0;
        printf("%-3d %-50s %9.4f %7.3f %6.4f %8.3f | ", idx, c.name, per, 16.0 * (double)n / (per * 1e-3) / 1e12, 16.0 * (double)n / (per * 1e-3) / 8e12, ghz);
        if (c.stamp) printf("%6.2f %6.2f %6.2f %6.2f %6.2f | ", iss / blocks * cyc2us, load / blocks * cyc2us, fwd / blocks * cyc2us, inv / blocks * cyc2us, sto / blocks * cyc2us);
        else printf("%6s %6s %6s %6s %6s | ", "-", "-", "-", "-", "-");
        printf("%8.2f %8.2f %8.2f", median(first), tail, (double)(t_max - t_min) * 0.01);
        if (rel >= 0) printf("  parity %.2g%s", rel, rel < 1e-6 ? "" : "  <-- MISMATCH");
        printf("\n");
        fflush(stdout);
        CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    }
    return 0;
}
