// Does a hipMemsetAsync captured into a hipGraph run on EVERY replay?  (tools/graph_probe.py: a captured reset + call of a stateful
// handle replays correctly once and then sees non-zero state.)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void use(float *state, float *out, int slot)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = state[slot * 8]; state[(slot ^ 1) * 8] = 123.0f; state[slot * 8 + 1] = 77.0f; }
}
int main()
{
    float *state, *out; hipStream_t s;
    CK(hipMalloc(&state, 64)); CK(hipMalloc(&out, 4)); CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int bytes : {64, 4096}) {
        float *st2; CK(hipMalloc(&st2, bytes));
        CK(hipMemset(st2, 0xff, bytes)); CK(hipDeviceSynchronize());
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        CK(hipMemsetAsync(st2, 0, bytes, s));
        hipLaunchKernelGGL(use, dim3(1), dim3(64), 0, s, st2, out, 0);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 4; rep++) {
            CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
            float h[16], o; CK(hipMemcpy(&o, out, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h, st2, 64, hipMemcpyDeviceToHost));
            printf("memset of %d bytes, replay %d: kernel read %g (want 0); state after: [0]=%g [1]=%g [8]=%g\n", bytes, rep, o, h[0], h[1], h[8]);
        }
    }
    return 0;
}
