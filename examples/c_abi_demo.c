/*
 * c_abi_demo.c -- the C ABI (include/pcx.h) from plain C, no Python and no PyTorch in the process:
 * what a Pothos block's work() does with host buffers.  Filters a complex_float32 stream with a
 * 63-tap complex FIR in three work()-sized pieces (history carried the way the reference's circular
 * buffer does), checks the result against the textbook sum on the host, then an FFT round trip.
 *
 *   make -C pothoscomms_amd/csrc && make -C examples && examples/c_abi_demo
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pcx.h"

#define CHECK(call)                                                               \
    do {                                                                          \
        int rc__ = (call);                                                        \
        if (rc__ != PCX_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc__, pcx_last_error()); return 1; } \
    } while (0)

int main(void)
{
    enum { K = 63, N = 200000 };
    int ndev = 0;
    CHECK(pcx_device_count(&ndev));
    printf("%s, %d device(s)\n", pcx_version(), ndev);

    /* taps: Hann-windowed sinc shifted to 0.05 cycles/sample (complex) */
    double taps[2 * K];
    for (int k = 0; k < K; k++) {
        const double m = k - (K - 1) / 2.0;
        const double sinc = m == 0 ? 0.2 : sin(M_PI * 0.2 * m) / (M_PI * m);
        const double w = 0.5 - 0.5 * cos(2 * M_PI * k / (K - 1));
        taps[2 * k] = sinc * w * cos(2 * M_PI * 0.05 * k);
        taps[2 * k + 1] = sinc * w * sin(2 * M_PI * 0.05 * k);
    }
    float *x = malloc(sizeof(float) * 2 * N), *y = malloc(sizeof(float) * 2 * N);
    unsigned s = 12345;
    for (int i = 0; i < 2 * N; i++) { s = s * 1664525u + 1013904223u; x[i] = (float)((int)(s >> 8) - (1 << 23)) / (float)(1 << 23); }

    pcx_fir *fir = NULL;
    CHECK(pcx_fir_create(PCX_F32, 1, 1, &fir));
    CHECK(pcx_fir_set_taps(fir, taps, K));
    size_t Kg = 0, need = 0;
    CHECK(pcx_fir_get_geometry(fir, &Kg, &need));
    printf("K = %zu, input requirement = %zu\n", Kg, need);

    /* three work() calls: whatever a call does not consume stays at the front of the next buffer */
    size_t pos = 0, made = 0;
    const size_t avail[3] = {50000, 120000, N};
    for (int c = 0; c < 3; c++) {
        size_t consumed = 0, produced = 0;
        CHECK(pcx_fir_process(fir, x + 2 * pos, avail[c] - pos, y + 2 * made, N - made, &consumed, &produced));
        printf("work %d: %zu in -> consumed %zu, produced %zu\n", c, avail[c] - pos, consumed, produced);
        pos += consumed; made += produced;
    }
    if (made != N - K + 1) { fprintf(stderr, "produced %zu, expected %d\n", made, N - K + 1); return 1; }

    /* host check on a few hundred outputs: y[n] = sum_k h[k] x[n + K-1 - k]  (FIRFilter.cpp:294-300) */
    double worst = 0, scale = 0;
    for (size_t n = 0; n < made; n += 997) {
        double re = 0, im = 0;
        for (int k = 0; k < K; k++) {
            const double xr = x[2 * (n + K - 1 - k)], xi = x[2 * (n + K - 1 - k) + 1];
            re += taps[2 * k] * xr - taps[2 * k + 1] * xi;
            im += taps[2 * k] * xi + taps[2 * k + 1] * xr;
        }
        const double e = hypot(re - y[2 * n], im - y[2 * n + 1]);
        if (e > worst) worst = e;
        if (hypot(re, im) > scale) scale = hypot(re, im);
    }
    printf("FIR max error / max|y| = %.2e\n", worst / scale);
    if (!(worst / scale <= 1e-5)) return 1;
    CHECK(pcx_fir_destroy(fir));

    /* FFT round trip, 1024 bins x 64 frames: ifft(fft(x)) = N x */
    enum { NB = 1024, NF = 64 };
    pcx_fft *fwd = NULL, *inv = NULL;
    CHECK(pcx_fft_create(PCX_F32, NB, 0, &fwd));
    CHECK(pcx_fft_create(PCX_F32, NB, 1, &inv));
    float *X = malloc(sizeof(float) * 2 * NB * NF), *xb = malloc(sizeof(float) * 2 * NB * NF);
    CHECK(pcx_fft_transform(fwd, x, X, NF));
    CHECK(pcx_fft_transform(inv, X, xb, NF));
    worst = 0;
    for (int i = 0; i < 2 * NB * NF; i++) { const double e = fabs(xb[i] / NB - x[i]); if (e > worst) worst = e; }
    printf("FFT round-trip max error = %.2e\n", worst);
    if (!(worst <= 1e-5)) return 1;
    CHECK(pcx_fft_destroy(fwd));
    CHECK(pcx_fft_destroy(inv));
    free(x); free(y); free(X); free(xb);

    /* the Q-format reading of the integer element types is a parameter (pcx_qformat): int16 Scale, 0.5 x (5, -5) = +-2.5 under the three
     * fromQ roundings the reference's own tests cannot tell apart -- floor 2 / -3, toward zero 2 / -2, nearest 3 / -2 */
    {
        const short in16[2] = {5, -5};
        const short want[3][2] = {{2, -3}, {2, -2}, {3, -2}};
        for (int m = 0; m < 3; m++) {
            const pcx_qformat q = {PCX_Q_FRAC_HALF_Q, PCX_Q_TRUNCATE, m};
            short out16[2] = {0, 0};
            CHECK(pcx_scale_q(PCX_I16, 0, 0.5, &q, in16, out16, 2));
            printf("scale int16 0.5 x (5, -5), fromQ mode %d: (%d, %d)\n", m, out16[0], out16[1]);
            if (out16[0] != want[m][0] || out16[1] != want[m][1]) return 1;
        }
        pcx_qformat cur;
        CHECK(pcx_get_qformat(&cur));
        if (cur.frac != 0 || cur.float_to_q != 0 || cur.from_q != 0) return 1;      /* the process-wide reading is the built-in default */
    }
    printf("ok\n");
    return 0;
}
