/* c_block_path.c -- /comms/fir_filter driven work() by work() from a plain C process (the system's HIP runtime, as a Pothos
 * process would load it; no Python, no torch): the PCIe-inclusive rate of the block on its own page-locked port buffers, single
 * device against setDevices([0,0]) / ([0,0,0,0]) (the call's samples split over shards -- here all on device 0, halos by peer
 * copies: one link, what this shows is scatter + pass + gather against the in-place call), and on the framework's pageable
 * double-mapped circular input (page-locked where it lies by the block).  Build: make -C examples */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pcx.h"
#include "pcx_blocks.h"

#define CK(x) do { if ((x) != 0) { fprintf(stderr, "%s: %s / %s\n", #x, pcxb_last_error(), pcx_last_error()); exit(1); } } while (0)
enum { K = 255 };

static double run(size_t n, int nshards, int circular)
{
    pcxb_block *b = NULL;
    CK(pcxb_make("/comms/fir_filter", "complex_float32", 1, "COMPLEX", 0, 0, &b));
    static double taps[2 * K];
    for (int k = 0; k < K; k++) {                       /* windowed-sinc band-pass: any 255 complex taps will do for a rate */
        const double t = k - (K - 1) / 2.0, w = 0.5 - 0.5 * cos(2 * M_PI * k / (K - 1)), s = t == 0 ? 0.1 : sin(0.1 * M_PI * t) / (M_PI * t);
        taps[2 * k] = w * s * cos(0.1 * M_PI * t);
        taps[2 * k + 1] = w * s * sin(0.1 * M_PI * t);
    }
    CK(pcxb_call_taps(b, "setTaps", taps, K, 1));
    if (nshards > 1) {
        size_t devs[8] = {0};
        CK(pcxb_call_sizes(b, "setDevices", devs, (size_t)nshards));
    }
    CK(pcxb_activate(b));
    void *in = NULL, *out = NULL, *circ = NULL;
    size_t got = 0, clen = 0;
    int pinned = 0;
    CK(pcxb_acquire_buffer(b, 1, n * 8, &out, &got, &pinned));
    if (circular) {
        CK(pcxb_circular_create(2 * (n + K) * 8, &circ, &clen));
        in = (char *)circ + (clen - n / 2 * 8);        /* the window runs across the wrap */
    } else {
        CK(pcxb_acquire_buffer(b, 0, (n + K - 1) * 8, &in, &got, &pinned));
    }
    float *x = (float *)in;
    unsigned long long sd = 88172645463325252ull;
    for (size_t i = 0; i < 2 * (n + K - 1); i++) { sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; x[i] = (float)((double)(sd >> 40) / 8388608.0 - 1.0); }
    double t = 0, best = 1e9;
    size_t c = 0, p = 0;
    CK(pcxb_work_loop(b, in, n + K - 1, out, n, 3, &t, &c, &p));
    const size_t reps = n >= ((size_t)1 << 24) ? 8 : (((size_t)1 << 25) / n);
    for (int r = 0; r < 3; r++) {
        CK(pcxb_work_loop(b, in, n + K - 1, out, n, reps, &t, &c, &p));
        if (t / reps < best) best = t / reps;
    }
    if (c != p || c + (size_t)nshards < n) { fprintf(stderr, "consumed %zu produced %zu of %zu\n", c, p, n); exit(1); }
    CK(pcxb_destroy(b));
    if (circ) CK(pcxb_circular_destroy(circ));
    return best;
}

int main(int argc, char **argv)
{
    if (argc > 1 && !strcmp(argv[1], "brief")) {      /* machine-readable, for bench.py: samples per call, seconds per call on pinned slabs / on the circular input */
        const size_t ns[] = {(size_t)1 << 20, (size_t)1 << 22, (size_t)1 << 23, (size_t)1 << 24};     /* 8 Mi samples = the default 64 MiB port slab */
        for (int i = 0; i < 2; i++) printf("%zu %.9f %.9f\n", ns[i], run(ns[i], 1, 0), run(ns[i], 1, 1));
        return 0;
    }
    double up, down, both;
    CK(pcx_pcie_probe((size_t)128 << 20, 3, &up, &down, &both));
    printf("plain C process, system HIP runtime.  PCIe: H2D alone %.1f GB/s, D2H alone %.1f, both at once %.1f per direction\n", up, down, both);
    const size_t sizes[] = {(size_t)1 << 20, (size_t)1 << 22, (size_t)1 << 24};
    for (int i = 0; i < 3; i++) {
        const size_t n = sizes[i];
        const double a = run(n, 1, 0), c = run(n, 1, 1), s2 = run(n, 2, 0), s4 = run(n, 4, 0);
        printf("n=%9zu  single device %7.3f ms %5.2f Gs/s | circular input, page-locked where it lies %7.3f ms %5.2f Gs/s | 2 shards %7.3f ms %5.2f Gs/s | 4 shards %7.3f ms %5.2f Gs/s\n",
               n, a * 1e3, n / a / 1e9, c * 1e3, n / c / 1e9, s2 * 1e3, n / s2 / 1e9, s4 * 1e3, n / s4 / 1e9);
    }
    return 0;
}
