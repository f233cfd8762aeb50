/* c_shard_demo.c -- the native multi-device entry points of include/pcx.h from a plain C process (no Python, no
 * PyTorch, the system HIP runtime and the system RCCL): what a Pothos block that owns a pcx_shard does inside work().
 *
 *   1. one shard per visible device over RCCL (on a one-GPU box: a communicator of one) -- with a single device the
 *      pass must be bit-identical to pcx_fir_process on the whole stream;
 *   2. two shards on device 0 with the peer-copy transport, halos poisoned first -- the concatenated outputs must match
 *      the single-device result to 1e-5 of its largest sample (the seam is invisible).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pcx.h"

#define CK(x) do { int rc_ = (x); if (rc_ != PCX_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, pcx_last_error()); return 1; } } while (0)

static float frand(unsigned long long *s)
{
    *s = *s * 6364136223846793005ull + 1442695040888963407ull;
    return (float)((double)(*s >> 40) / 8388608.0 - 1.0);
}

int main(void)
{
    enum { K = 255 };
    const size_t C = 50000;
    int ndev = 0;
    CK(pcx_device_count(&ndev));
    if (ndev < 1) { fprintf(stderr, "no device\n"); return 1; }
    if (ndev > 8) ndev = 8;
    double taps[2 * K];
    for (int k = 0; k < K; k++) {   /* windowed complex band-pass, values irrelevant here */
        const double w = 0.5 - 0.5 * cos(2.0 * M_PI * (k + 1) / (K + 1)), t = k - (K - 1) / 2.0;
        const double sinc = t == 0 ? 0.1 : sin(M_PI * 0.1 * t) / (M_PI * t);
        taps[2 * k] = w * sinc * cos(2.0 * M_PI * 0.05 * k);
        taps[2 * k + 1] = w * sinc * sin(2.0 * M_PI * 0.05 * k);
    }
    /* ---- 1. RCCL, one shard per device ---- */
    {
        const size_t total = (size_t)ndev * C, n_in = K - 1 + total;
        float *x = malloc(n_in * 8), *y = malloc(total * 8), *ref = malloc(total * 8);
        unsigned long long seed = 7;
        for (size_t i = 0; i < 2 * n_in; i++) x[i] = frand(&seed);
        pcx_fir *f;
        size_t c = 0, p = 0;
        CK(pcx_fir_create(PCX_F32, 1, 1, &f));
        CK(pcx_fir_set_taps(f, taps, K));
        CK(pcx_fir_process(f, x, n_in, ref, total, &c, &p));
        if (p != total) { fprintf(stderr, "reference pass produced %zu of %zu\n", p, total); return 1; }
        pcx_shard *s;
        CK(pcx_shard_create(ndev, NULL, PCX_SHARD_RCCL, &s));
        CK(pcx_shard_set_taps(s, taps, K, 1));
        CK(pcx_shard_configure(s, C));
        CK(pcx_shard_scatter(s, x, n_in));
        CK(pcx_shard_step(s));
        CK(pcx_shard_step(s));          /* a second pass re-receives into halo slots the first pass read */
        CK(pcx_shard_gather(s, y, total));
        double mx = 0, md = 0;
        for (size_t i = 0; i < 2 * total; i++) { mx = fmax(mx, fabs(ref[i])); md = fmax(md, fabs((double)y[i] - ref[i])); }
        printf("rccl: %d device(s) x %zu samples, max|diff| / max|ref| = %.3g%s\n", ndev, C, md / mx, ndev == 1 ? " (must be 0)" : "");
        if (ndev == 1 ? md != 0 : md / mx > 1e-5) { fprintf(stderr, "FAIL\n"); return 1; }
        CK(pcx_shard_destroy(s));
        CK(pcx_fir_destroy(f));
        free(x); free(y); free(ref);
    }
    /* ---- 2. two shards on device 0, peer copies, poisoned halos ---- */
    {
        const int devs[2] = {0, 0};
        const size_t total = 2 * C, n_in = K - 1 + total;
        float *x = malloc(n_in * 8), *y = malloc(total * 8), *ref = malloc(total * 8), *nan = malloc((K - 1) * 8);
        unsigned long long seed = 9;
        for (size_t i = 0; i < 2 * n_in; i++) x[i] = frand(&seed);
        for (size_t i = 0; i < 2 * (K - 1); i++) nan[i] = NAN;
        pcx_fir *f;
        size_t c = 0, p = 0;
        CK(pcx_fir_create(PCX_F32, 1, 1, &f));
        CK(pcx_fir_set_taps(f, taps, K));
        CK(pcx_fir_process(f, x, n_in, ref, total, &c, &p));
        pcx_shard *s;
        CK(pcx_shard_create(2, devs, PCX_SHARD_PEER_COPY, &s));
        CK(pcx_shard_set_taps(s, taps, K, 1));
        CK(pcx_shard_configure(s, C));
        CK(pcx_shard_scatter(s, x, n_in));
        void *in1, *st1;
        CK(pcx_shard_buffers(s, 1, &in1, NULL, &st1, NULL));
        CK(pcx_memcpy_h2d(in1, nan, (K - 1) * 8, st1));
        CK(pcx_shard_step(s));
        CK(pcx_shard_gather(s, y, total));
        double mx = 0, md = 0;
        for (size_t i = 0; i < 2 * total; i++) { mx = fmax(mx, fabs(ref[i])); md = fmax(md, fabs((double)y[i] - ref[i])); }
        printf("peer copy: 2 shards on device 0, max|diff| / max|ref| = %.3g\n", md / mx);
        if (!(md / mx <= 1e-5)) { fprintf(stderr, "FAIL\n"); return 1; }
        CK(pcx_shard_destroy(s));
        CK(pcx_fir_destroy(f));
        free(x); free(y); free(ref); free(nan);
    }
    printf("ok\n");
    return 0;
}
