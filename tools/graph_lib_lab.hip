// pcx_freqdemod_reset + pcx_freqdemod_process_dev captured into a hipGraph by hand (no torch): does every replay start from zero state?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "pcx.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define PK(x) do { int r_ = (x); if (r_ != 0) { printf("%s: %d %s\n", #x, r_, pcx_last_error()); return 1; } } while (0)
int main()
{
    const size_t n = 1 << 16;
    std::vector<float> hx(2 * n);
    for (size_t i = 0; i < 2 * n; i++) hx[i] = (float)((i * 2654435761u) % 1000) / 500.0f - 1.0f;
    float *x, *y; hipStream_t s;
    CK(hipMalloc(&x, 8 * n)); CK(hipMalloc(&y, 4 * n)); CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipMemcpy(x, hx.data(), 8 * n, hipMemcpyHostToDevice));
    pcx_freqdemod *h; PK(pcx_freqdemod_create(PCX_F32, &h));
    PK(pcx_freqdemod_reset(h)); PK(pcx_freqdemod_process_dev(h, x, y, n, s)); CK(hipStreamSynchronize(s));
    float want[2]; CK(hipMemcpy(want, y, 8, hipMemcpyDeviceToHost));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    PK(pcx_freqdemod_reset(h));
    PK(pcx_freqdemod_process_dev(h, x, y, n, s));
    CK(hipStreamEndCapture(s, &g));
    size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn));
    printf("graph has %zu nodes\n", nn);
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int where = 0; where < 2; where++) {
        hipStream_t ls = where ? nullptr : s;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipMemset(y, 0xff, 4 * n));
            CK(hipDeviceSynchronize());
            CK(hipGraphLaunch(ge, ls)); CK(hipDeviceSynchronize());
            float got[2]; CK(hipMemcpy(got, y, 8, hipMemcpyDeviceToHost));
            printf("%s, replay %d: y[0] = %g (want %g), y[1] = %g (want %g)\n", where ? "launched on the null stream" : "launched on the capture stream", rep, got[0],
                   want[0], got[1], want[1]);
        }
    }
    return 0;
}
