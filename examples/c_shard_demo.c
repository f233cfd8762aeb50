/* c_shard_demo.c -- the native multi-device entry points of include/pcx.h from a plain C process (no Python, no
 * PyTorch, the system HIP runtime and the system RCCL): what a Pothos block that owns a pcx_shard does inside work().
 *
 *   1. one shard per visible device over RCCL (on a one-GPU box: a communicator of one) -- with a single device the
 *      pass must be bit-identical to pcx_fir_process on the whole stream;
 *   2. two shards on device 0 with the peer-copy transport, halos poisoned first -- the concatenated outputs must match
 *      the single-device result to 1e-5 of its largest sample (the seam is invisible);
 *   3. the fused chain Rotate -> FIR -> FreqDemod (pcx_shard_set_chain, BASELINE configs[4]) over two shards on device 0, once with
 *      short shards and once with shards of more than 2048 blocks -- ONE gated launch per shard and pass
 *      (pcx_fmchain_process_dev_gated) -- against pcx_fmchain_process on the whole stream.
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
    /* ---- 3. the fused chain over two shards on device 0 ---- */
    for (int big = 0; big < 2; big++) {
        enum { KC = 127 };
        const int devs[2] = {0, 0};
        const size_t Cc = big ? (size_t)2080 * 3968 : 30000, total = 2 * Cc, n_in = KC - 1 + total;
        const double phase = 0.7;
        double rtaps[KC];
        for (int k = 0; k < KC; k++) {
            const double w = 0.5 - 0.5 * cos(2.0 * M_PI * (k + 1) / (KC + 1)), t = k - (KC - 1) / 2.0;
            rtaps[k] = w * (t == 0 ? 0.2 : sin(M_PI * 0.2 * t) / (M_PI * t));
        }
        float *x = malloc(n_in * 8), *y = malloc(total * 4), *ref = malloc(total * 4);
        /* an FM signal with a little noise: the envelope never vanishes, so every angle is well conditioned */
        unsigned long long seed = 11;
        double ph = 0;
        for (size_t i = 0; i < n_in; i++) {
            ph += 2.0 * M_PI * (0.02 + 0.01 * sin(2.0 * M_PI * (double)i / 1000.0));
            x[2 * i] = (float)(cos(ph) + 1e-3 * frand(&seed));
            x[2 * i + 1] = (float)(sin(ph) + 1e-3 * frand(&seed));
        }
        pcx_fmchain *ch;
        size_t c = 0, p = 0;
        CK(pcx_fmchain_create(&ch));
        CK(pcx_fmchain_set_phase(ch, phase));
        CK(pcx_fmchain_set_taps(ch, rtaps, KC, 0));
        CK(pcx_fmchain_process(ch, x, n_in, ref, total, &c, &p));
        if (p != total) { fprintf(stderr, "reference chain produced %zu of %zu\n", p, total); return 1; }
        pcx_shard *s;
        CK(pcx_shard_create(2, devs, PCX_SHARD_PEER_COPY, &s));
        CK(pcx_shard_set_chain(s, 1, phase));
        CK(pcx_shard_set_taps(s, rtaps, KC, 0));
        CK(pcx_shard_configure(s, Cc));
        CK(pcx_shard_scatter(s, x, n_in));
        CK(pcx_shard_step(s));
        CK(pcx_shard_step(s));
        CK(pcx_shard_gather(s, y, total));
        CK(pcx_shard_sync(s));          /* also reports a gate that timed out */
        double md = 0;
        for (size_t i = 0; i < total; i++) {
            double d = fabs((double)y[i] - ref[i]);
            if (d > M_PI) d = 2.0 * M_PI - d;
            md = fmax(md, d);
        }
        printf("fused chain: 2 shards x %zu samples on device 0 (%s), max|wrap(diff)| / pi = %.3g\n", Cc,
               big ? "one gated launch per shard" : "short shards", md / M_PI);
        if (!(md / M_PI <= 2e-5)) { fprintf(stderr, "FAIL\n"); return 1; }
        CK(pcx_shard_destroy(s));
        CK(pcx_fmchain_destroy(ch));
        free(x); free(y); free(ref);
    }
    /* ---- 4. double-buffered: two handles, three batches; the halos of batch k+1 travel while batch k is filtered ---- */
    {
        const int devs[2] = {0, 0};
        const size_t total = 2 * C, n_in = K - 1 + total;
        float *x = malloc(n_in * 8), *y = malloc(total * 8), *ref = malloc(total * 8);
        pcx_fir *f;
        CK(pcx_fir_create(PCX_F32, 1, 1, &f));
        CK(pcx_fir_set_taps(f, taps, K));
        pcx_shard *ab[2];
        for (int i = 0; i < 2; i++) {
            CK(pcx_shard_create(2, devs, PCX_SHARD_PEER_COPY, &ab[i]));
            CK(pcx_shard_set_taps(ab[i], taps, K, 1));
            CK(pcx_shard_configure(ab[i], C));
        }
        if (pcx_shard_compute(ab[0]) != PCX_ERR_STATE) { fprintf(stderr, "compute without a posted exchange must be refused\n"); return 1; }
        unsigned long long seed = 21;
        double worst = 0;
        for (size_t i = 0; i < 2 * n_in; i++) x[i] = frand(&seed);
        CK(pcx_shard_scatter(ab[0], x, n_in));                 /* batch 0 */
        CK(pcx_shard_post_exchange(ab[0]));
        for (int k = 0; k < 3; k++) {
            pcx_shard *cur = ab[k & 1], *nxt = ab[(k + 1) & 1];
            size_t c = 0, p = 0;
            CK(pcx_fir_process(f, x, n_in, ref, total, &c, &p));      /* what batch k must come out as */
            CK(pcx_shard_compute(cur));                                /* batch k: its exchange was posted a turn ago */
            for (size_t i = 0; i < 2 * n_in; i++) x[i] = frand(&seed); /* batch k+1 */
            CK(pcx_shard_scatter(nxt, x, n_in));
            CK(pcx_shard_post_exchange(nxt));
            CK(pcx_shard_gather(cur, y, total));
            double mx = 0, md = 0;
            for (size_t i = 0; i < 2 * total; i++) { mx = fmax(mx, fabs(ref[i])); md = fmax(md, fabs((double)y[i] - ref[i])); }
            if (!(md / mx <= 1e-5)) { fprintf(stderr, "FAIL: batch %d, %.3g\n", k, md / mx); return 1; }
            worst = fmax(worst, md / mx);
        }
        CK(pcx_shard_compute(ab[1]));                          /* the exchange still posted for batch 3 */
        CK(pcx_shard_sync(ab[1]));
        printf("double-buffered: 3 batches through two handles of 2 shards, max|diff| / max|ref| = %.3g\n", worst);
        for (int i = 0; i < 2; i++) CK(pcx_shard_destroy(ab[i]));
        CK(pcx_fir_destroy(f));
        free(x); free(y); free(ref);
    }
    printf("ok\n");
    return 0;
}
