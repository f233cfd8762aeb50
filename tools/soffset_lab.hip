// soffset_lab.hip -- is the SCALAR offset of a raw buffer store part of the range check on gfx950?  (diagnostic; make -C tools soffset_lab)
// A descriptor of 1024 bytes over a 16 KiB buffer of sentinels; lane l stores at voffset 4 l with soffset 0 / 512 / 2048 / 8192.
// If the range check covers voffset + imm only, the stores with soffset >= 1024 land outside the descriptor's records.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
__global__ void probe(unsigned *buf, int soff, unsigned tag)
{
    const __amdgpu_buffer_rsrc_t ws = make_rsrc(buf, 1024);
    __builtin_amdgcn_raw_buffer_store_b32(tag + threadIdx.x, ws, (int)(threadIdx.x * 4u), soff, 0);
}
__global__ void probe_neg(unsigned *buf, int soff, unsigned tag)     // voffset wraps (negative lane offset), positive sum
{
    const __amdgpu_buffer_rsrc_t ws = make_rsrc(buf, 1024);
    __builtin_amdgcn_raw_buffer_store_b32(tag + threadIdx.x, ws, (int)((threadIdx.x - 32u) * 4u), soff, 0);
}
int main()
{
    unsigned *d;
    std::vector<unsigned> h(4096);
    hipMalloc(&d, 16384);
    for (int soff : {0, 512, 900, 1020, 2048, 8192}) {   // 900: lanes 0..30 fit (900 + 4*30 + 4 = 1024), 31.. do not; 1020: lane 0 alone
        hipMemset(d, 0xEE, 16384);
        probe<<<1, 64>>>(d, soff, 0x1000u);
        hipMemcpy(h.data(), d, 16384, hipMemcpyDeviceToHost);
        int n = 0, first = -1;
        for (int i = 0; i < 4096; i++) if (h[i] != 0xEEEEEEEEu) { n++; if (first < 0) first = i; }
        printf("voffset 4*lane, soffset %5d, 1024 records: %2d words written, first at byte %d\n", soff, n, first * 4);
    }
    for (int soff : {0, 256, 2048}) {
        hipMemset(d, 0xEE, 16384);
        probe_neg<<<1, 64>>>(d, soff, 0x2000u);
        hipMemcpy(h.data(), d, 16384, hipMemcpyDeviceToHost);
        int n = 0, first = -1;
        for (int i = 0; i < 4096; i++) if (h[i] != 0xEEEEEEEEu) { n++; if (first < 0) first = i; }
        printf("voffset 4*(lane-32) (wraps for lanes < 32), soffset %5d: %2d words written, first at byte %d\n", soff, n, first * 4);
    }
    return 0;
}
