// Raw buffer loads whose lane offsets start NEGATIVE (wrapped) inside a wave: which valid lanes come back?
// Found through fir_ols_part.hip's real-stream prologue (round 6): `buffer_load_dword v, voff, rsrc, 0 offen offset:IMM` with IMM != 0 and a
// wrapped register offset drops the VALID lanes that share an aligned four-lane group with lanes that are still out of range after
// the addition (1-3 samples at the front of a buffer lost); IMM = 0 and dwordx2 loads are not affected; nt makes no difference.
// The compiler produces that form by itself from `(lane + 256 * row - shift) * 4` (row constant -> instruction offset).
//   make -C tools bufload_quad_lab && ./tools/bufload_quad_lab        (profiles/r06/bufload_quad_lab.txt)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(const float *in, float *out, int first_valid, unsigned records)
{
    const int j = threadIdx.x;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)records, 0x00020000);
    // MODE 0: dword, imm 0; 1: dword, imm 3072; 2: dwordx2 imm 0; 3: dwordx2 imm 2048; 4: dword imm 3072 nt; 5: dwordx2 imm 2048 nt
    constexpr int EB = (MODE == 2 || MODE == 3 || MODE == 5) ? 8 : 4;
    constexpr int IMM = (MODE == 1 || MODE == 4) ? 3072 : (MODE == 3 || MODE == 5) ? 2048 : 0;
    int v = (j - first_valid) * EB - IMM;
    asm volatile("" : "+v"(v));
    unsigned a = 0xdead; unsigned long long c = 0xdead;
    if (MODE == 0) asm volatile("buffer_load_dword %0, %1, %2, 0 offen\n\ts_waitcnt vmcnt(0)" : "=&v"(a) : "v"(v), "s"(rs));
    if (MODE == 1) asm volatile("buffer_load_dword %0, %1, %2, 0 offen offset:3072\n\ts_waitcnt vmcnt(0)" : "=&v"(a) : "v"(v), "s"(rs));
    if (MODE == 4) asm volatile("buffer_load_dword %0, %1, %2, 0 offen offset:3072 nt\n\ts_waitcnt vmcnt(0)" : "=&v"(a) : "v"(v), "s"(rs));
    if (MODE == 2) asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen\n\ts_waitcnt vmcnt(0)" : "=&v"(c) : "v"(v), "s"(rs));
    if (MODE == 3) asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen offset:2048\n\ts_waitcnt vmcnt(0)" : "=&v"(c) : "v"(v), "s"(rs));
    if (MODE == 5) asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen offset:2048 nt\n\ts_waitcnt vmcnt(0)" : "=&v"(c) : "v"(v), "s"(rs));
    out[j] = EB == 4 ? __uint_as_float(a) : __uint_as_float((unsigned)c);
}
int main()
{
    float *in, *out;
    (void)hipMalloc(&in, 1 << 20); (void)hipMalloc(&out, 64 * 4);
    static float h[8192]; for (int i = 0; i < 8192; i++) h[i] = 1000.f + i;
    (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    for (int mode = 0; mode < 6; mode++)
        for (int fv : {16, 17, 18, 19, 29, 30, 31, 47}) {
            const unsigned rec = 4096 * 4;
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, in, out, fv, rec);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, in, out, fv, rec);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, in, out, fv, rec);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, in, out, fv, rec);
            if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, in, out, fv, rec);
            if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(1), dim3(64), 0, 0, in, out, fv, rec);
            float r[64]; (void)hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost);
            const int eb = (mode == 2 || mode == 3 || mode == 5) ? 2 : 1;
            int lost = 0, leak = 0;
            for (int j = 0; j < 64; j++) {
                if (j >= fv && r[j] != 1000.f + (j - fv) * eb) lost++;
                if (j < fv && r[j] != 0.f) leak++;
            }
            printf("mode %d first valid lane %2d: valid lanes lost %d, invalid lanes not zero %d\n", mode, fv, lost, leak);
        }
    return 0;
}
