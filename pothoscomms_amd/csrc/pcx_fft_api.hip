// pcx_fft_api.hip -- the extern "C" boundary (include/pcx.h), part 3: /comms/fft (pcx_fft_*), /comms/freq_demod
// (pcx_freqdemod_*) and the stateless maps (rotate, scale, abs, conjugate, angle, arithmetic, split / combine complex).
// Host-side only.
#include "pcx_host.hpp"
#include "pcx_tables.hpp"

using namespace pcx;

/* ===================================================================== *
 *  FFT
 * ===================================================================== */
struct pcx_fft {
    ExecCtx cx;
    int scalar = PCX_F32;
    size_t nbins = 0;
    int inverse = 0;
    enum Kind { IDENTITY, R16_4096, R16, POW2, Q15_POW2, Q15_GLOBAL, MIXED, SMOOTH, FOURSTEP, FOURSTEP_SHORT, BLUESTEIN } kind = MIXED;
    int log2n = 0;
    DevBuf tw, perm;
    StageBuf wsIn, wsOut;
    DevBuf sched;            // dynamic frame assignment of fft4096_kernel (pcx_sched.hpp; diagnostic A/B only since the family kernel took over), zeroed at create
    std::vector<int> radix;  // kf_factor order (kissfft.hh:38-55 / kiss_fft.c:309-328 give the same list)
    // FOURSTEP (fft_large.hip): numBins = n1 * n2, both within the single-workgroup plans
    size_t n1 = 0, n2 = 0;
    pcx_fft *sub1 = nullptr, *sub2 = nullptr;
    DevBuf ws1, ws2;
    // BLUESTEIN (fft_bluestein.hip): n2 = M, the power-of-two convolution size; sub1 / sub2 = forward / inverse M-point plans;
    // tw = the chirp w[N], tw1 = B[M] = FFT_M of the wrapped conjugate chirp; ws1 = M-point work rows
    // FOURSTEP_SHORT (complex_float32, numBins <= 4 Mi): n1 = 256 columns pass with strided I/O (fft_large.hip),
    // then rows of n2 -- with the final transpose on their store when n2 <= 256, else sub2 + one transpose
    DevBuf tw1, tw2;
    ~pcx_fft() { delete sub1; delete sub2; }
};
// longest power-of-two transform one workgroup handles
static bool fft_is_5_smooth(size_t n)
{
    for (size_t r : {2, 3, 5})
        while (n % r == 0) n /= r;
    return n == 1;
}
static size_t fft_single_wg_limit(int scalar) { return scalar == PCX_F32 ? 16384 : scalar == PCX_F64 ? 8192 : 4096; }

int pcx_fft_create(int scalar, size_t num_bins, int inverse, pcx_fft **out)
{
    PCX_CHECK_ARG(out, "null out");
    // FFTFactory, FFT.cpp:83-93: complex<double>, complex<float>, complex<int16> only
    PCX_CHECK_ARG(scalar == PCX_F64 || scalar == PCX_F32 || scalar == PCX_I16, "FFTFactory: unsupported type (scalar %d)", scalar);
    PCX_CHECK_ARG(num_bins >= 1, "FFT: numBins must be >= 1");
    const size_t esz = 2 * (size_t)scalar_bytes(scalar);
    const bool pow2 = (num_bins & (num_bins - 1)) == 0;
    // single-workgroup LDS plans: the frame (x2 for ping-pong) must fit 160 KB
    const bool r16_f64 = scalar == PCX_F64 && pow2 && num_bins >= 16 && num_bins <= 8192 && !(PCX_ENV_SET("PCX_FFT_F64_POW2") && num_bins <= 4096);
    const bool r16 = (scalar == PCX_F32 && pow2 && num_bins >= 16 && num_bins <= 16384) || r16_f64;
    // float power-of-two sizes beyond one workgroup: four-step around the short kernels (fft_large.hip)
    const size_t wg_limit = fft_single_wg_limit(scalar);
    const bool four_step = scalar != PCX_I16 && pow2 && num_bins > wg_limit && num_bins <= wg_limit * wg_limit;
    // float sizes with other factors that do not fit one workgroup's LDS (ping-pong image): the same four-step
    // decomposition N = n1 * n2 around two mixed-radix (or power-of-two) plans, n1 the largest divisor <= sqrt(N)
    // whose cofactor still fits.  (kissfft recurses over the factor list instead, kissfft.hh:81-161: same DFT.)
    size_t mixed_n1 = 0;
    const size_t lds_limit = 160 * 1024 / (2 * esz);
    if (scalar != PCX_I16 && !pow2 && num_bins > lds_limit) {
        for (size_t d = (size_t)std::floor(std::sqrt((double)num_bins)); d >= 2; d--)
            if (num_bins % d == 0) { if (num_bins / d <= lds_limit) mixed_n1 = d; break; }
    }
    bool bluestein = false, q15_global = false;
    if (num_bins > 1 && !r16 && !four_step && !mixed_n1 && num_bins * esz * ((scalar == PCX_I16 && pow2) ? 1 : 2) > 160 * 1024) {
        if (num_bins > ((size_t)1 << 26)) {
            set_error("FFT: numBins=%zu is beyond every device plan (2^26 bins)", num_bins);
            return PCX_ERR_UNSUPPORTED;
        }
        // complex_int16 frames that no workgroup's LDS holds: kf_work's stages one launch each over global memory -- the Q15
        // rounding sequence of kiss_fft is kept whatever the size (fft_mixed.hip launch_fft_q15_global); a four-step split
        // would not keep it
        if (scalar == PCX_I16) q15_global = true;
        else bluestein = true;   // float sizes with no other plan (e.g. 2 x a prime beyond one workgroup): chirp-z on the power-of-two plans
    }
    pcx_fft *h = new (std::nothrow) pcx_fft();
    if (!h) { set_error("out of memory"); return PCX_ERR_STATE; }
    h->scalar = scalar; h->nbins = num_bins; h->inverse = inverse ? 1 : 0;
    DeviceScope bind(h->cx.device);   // tables are uploaded below: the handle belongs to the creating thread's current device
    {   // kf_factor: 4s, then 2s, then 3, 5, 7, ... (kiss_fft.c:309-328)
        int n = (int)num_bins, p = 4;
        const double floor_sqrt = std::floor(std::sqrt((double)n));
        if (n > 1) do {
            while (n % p) {
                switch (p) { case 4: p = 2; break; case 2: p = 3; break; default: p += 2; break; }
                if (p > floor_sqrt) p = n;
            }
            n /= p;
            h->radix.push_back(p);
        } while (n > 1);
    }
    const double two_pi = 6.283185307179586476925286766559;
    int rc = PCX_OK;
    if (num_bins == 1) {
        h->kind = pcx_fft::IDENTITY;
    } else if (bluestein) {
        h->kind = pcx_fft::BLUESTEIN;
        size_t M = 1;
        while (M < 2 * num_bins - 1) M <<= 1;
        h->n1 = num_bins; h->n2 = M;
        // w[n] = exp(-j pi n^2 / N), n^2 reduced modulo 2N so the phase is exact for any N
        std::vector<double> w(2 * num_bins), b(2 * M, 0.0);
        for (size_t n = 0; n < num_bins; n++) {
            const unsigned long long r = ((unsigned long long)n * (unsigned long long)n) % (2ull * num_bins);
            const double ph = -3.141592653589793238462643383279502884 * (double)r / (double)num_bins;
            w[2 * n] = std::cos(ph); w[2 * n + 1] = std::sin(ph);
            // conjugate chirp, wrapped: b[n] = b[M - n] = conj(w[n])
            b[2 * n] = w[2 * n]; b[2 * n + 1] = -w[2 * n + 1];
            if (n) { b[2 * (M - n)] = w[2 * n]; b[2 * (M - n) + 1] = -w[2 * n + 1]; }
        }
        rc = pcx_fft_create(scalar, M, 0, &h->sub1);
        if (rc == PCX_OK) rc = pcx_fft_create(scalar, M, 1, &h->sub2);
        if (rc == PCX_OK) {
            if (scalar == PCX_F32) {
                std::vector<float> wf(w.begin(), w.end()), bf(b.begin(), b.end());
                rc = upload(h->tw, wf);
                if (rc == PCX_OK) rc = upload(h->ws1, bf);
            } else {
                rc = upload(h->tw, w);
                if (rc == PCX_OK) rc = upload(h->ws1, b);
            }
        }
        // B = FFT_M(b), once, on the device (the same plan the frames use)
        if (rc == PCX_OK) rc = h->tw1.ensure(M * esz);
        if (rc == PCX_OK) rc = pcx_fft_transform_dev(h->sub1, h->ws1.p, h->tw1.p, 1, nullptr);
        if (rc == PCX_OK && hipStreamSynchronize(nullptr) != hipSuccess) { set_error("hipStreamSynchronize failed"); rc = PCX_ERR_HIP; }
    } else if (four_step && ((scalar == PCX_F32 && num_bins <= ((size_t)4 << 20)) || (scalar == PCX_F64 && num_bins <= ((size_t)2 << 20))) &&
               !PCX_ENV_SET("PCX_FFT_FIVE_PASS")) {
        h->kind = pcx_fft::FOURSTEP_SHORT;
        const size_t sub_limit = fft_single_wg_limit(scalar);   // longest row transform: 16384 (float) / 8192 (double) bins
        const size_t n1_forced = (size_t)PCX_ENV_INT("PCX_FFT_N1", 0);
        // measured (tools/sweep_fft.py): 128 columns per tile (256-byte runs) beat 256 except where only n1 = 256
        // leaves n2 <= 256 (65,536 bins: two passes instead of three) or n2 would exceed the 16384-bin plans
        h->n1 = num_bins == 65536 ? 256 : 128;
        if (n1_forced == 128 || n1_forced == 256) h->n1 = n1_forced;
        if (num_bins / h->n1 > sub_limit) h->n1 = 256;
        h->n2 = num_bins / h->n1;                // 128 ... 16384
        const bool f64 = scalar == PCX_F64;
        rc = f64 ? upload(h->tw1, make_tw_r16<double>(h->n1 == 128 ? 7 : 8)) : upload(h->tw1, make_tw_r16(h->n1 == 128 ? 7 : 8));
        if (rc == PCX_OK && h->n2 <= 256) {
            int l2 = 0;
            while (((size_t)1 << l2) < h->n2) l2++;
            rc = f64 ? upload(h->tw2, make_tw_r16<double>(l2)) : upload(h->tw2, make_tw_r16(l2));
        } else if (rc == PCX_OK) {
            rc = pcx_fft_create(scalar, h->n2, inverse, &h->sub2);
        }
    } else if (four_step || mixed_n1) {
        h->kind = pcx_fft::FOURSTEP;
        int l2 = 0;
        while (((size_t)1 << l2) < num_bins) l2++;
        h->n1 = mixed_n1 ? mixed_n1 : (size_t)1 << ((l2 + 1) / 2);
        h->n2 = num_bins / h->n1;
        rc = pcx_fft_create(scalar, h->n1, inverse, &h->sub1);
        if (rc == PCX_OK) rc = pcx_fft_create(scalar, h->n2, inverse, &h->sub2);
    } else if (scalar == PCX_F32 && num_bins == 4096) {
        h->kind = pcx_fft::R16_4096;
        rc = upload(h->tw, make_tw4096());
        if (rc == PCX_OK) rc = h->sched.ensure_zeroed(kSchedBytes);
    } else if (scalar == PCX_F32 && pow2 && num_bins >= 16 && num_bins <= 16384) {
        h->kind = pcx_fft::R16;
        while (((size_t)1 << h->log2n) < num_bins) h->log2n++;
        rc = upload(h->tw, make_tw_r16(h->log2n));
    } else if (r16_f64) {
        // the same radix-16 plan in double precision (fft_r16_f64.hip); PCX_FFT_F64_POW2 (A/B) keeps the radix-2/4 LDS kernel
        h->kind = pcx_fft::R16;
        while (((size_t)1 << h->log2n) < num_bins) h->log2n++;
        rc = upload(h->tw, make_tw_r16<double>(h->log2n));
    } else if (!pow2 && fft_is_5_smooth(num_bins) && !PCX_ENV_SET("PCX_FFT_KISS_ORDER") &&
               ((scalar == PCX_F32 && num_bins < 8192) || (scalar == PCX_F64 && num_bins >= 256 && num_bins < 2048))) {
        // complex_float32 / complex_float64, 2^a 3^b 5^c bins: a float transform may take its radices in any order -- 16s first, then
        // 8 / 4 / 2, 6 / 15, 5s, 3s (fft_smooth_f32_kernel); kissfft's own order stays with the bit-exact Q15 path.
        // PCX_FFT_KISS_ORDER (A/B) keeps the kissfft plan, as do the sizes where it measured faster (tools/sweep_fft_mixed.py):
        // float from 8192 bins up (10000: 102 vs 94 Gsamples/s), double below 256 and from 2048 up (60: 118 vs 87, 3000: 84 vs 65).
        // Forward table; the kernel conjugates around it for the inverse.
        h->kind = pcx_fft::SMOOTH;
        h->radix.clear();
        // 16s, one of 8 / 4 / 2 for the remaining twos, then pairs of odd factors as single passes (2 x 3 = 6 and 3 x 5 = 15:
        // prime-factor butterflies without inner twiddles; 3 x 3 = 9 with them), then the 5s and a 3 left over.
        // PCX_FFT_SMOOTH_PRIMES (A/B): no pairs
        int e2 = 0, e3 = 0, e5 = 0;
        for (size_t n = num_bins; n % 2 == 0; n /= 2) e2++;
        for (size_t n = num_bins; n % 3 == 0; n /= 3) e3++;
        for (size_t n = num_bins; n % 5 == 0; n /= 5) e5++;
        const bool pairs = !PCX_ENV_SET("PCX_FFT_SMOOTH_PRIMES");
        for (; e2 >= 4; e2 -= 4) h->radix.push_back(16);
        if (e2 == 1 && e3 > 0 && pairs) { h->radix.push_back(6); e3--; }
        else if (e2 > 0) h->radix.push_back(1 << e2);
        // how many 3 x 5 pairs leave the fewest passes once the remaining 3s go out two at a time (3 x 3 = 9, inner twiddles)
        int n15 = 0, best = 1 << 30;
        for (int c = 0; pairs && c <= std::min(e3, e5); c++) {
            const int passes = c + (e5 - c) + (e3 - c + 1) / 2;
            if (passes <= best) { best = passes; n15 = c; }
        }
        for (int c = 0; c < n15; c++, e3--, e5--) h->radix.push_back(15);
        for (; e5 > 0; e5--) h->radix.push_back(5);
        for (; pairs && e3 >= 2; e3 -= 2) h->radix.push_back(9);
        for (; e3 > 0; e3--) h->radix.push_back(3);
        if (scalar == PCX_F32) {
            std::vector<float> t(2 * num_bins);
            for (size_t i = 0; i < num_bins; i++) { t[2 * i] = (float)std::cos(two_pi * i / num_bins); t[2 * i + 1] = (float)(-std::sin(two_pi * i / num_bins)); }
            rc = upload(h->tw, t);
        } else {
            std::vector<double> t(2 * num_bins);
            for (size_t i = 0; i < num_bins; i++) { t[2 * i] = std::cos(two_pi * i / num_bins); t[2 * i + 1] = -std::sin(two_pi * i / num_bins); }
            rc = upload(h->tw, t);
        }
    } else if (scalar != PCX_I16) {
        // forward table exp(-j 2 pi i / N); the power-of-two kernels conjugate it for the inverse,
        // the mixed-radix kernel gets the direction baked in like kissfft's fill_twiddles (kissfft.hh:21-26)
        h->kind = pow2 ? pcx_fft::POW2 : pcx_fft::MIXED;
        const double sgn = (!pow2 && h->inverse) ? 1.0 : -1.0;
        if (scalar == PCX_F32) {
            std::vector<float> t(2 * num_bins);
            for (size_t i = 0; i < num_bins; i++) { t[2 * i] = (float)std::cos(two_pi * i / num_bins); t[2 * i + 1] = (float)(sgn * std::sin(two_pi * i / num_bins)); }
            rc = upload(h->tw, t);
        } else {
            std::vector<double> t(2 * num_bins);
            for (size_t i = 0; i < num_bins; i++) { t[2 * i] = std::cos(two_pi * i / num_bins); t[2 * i + 1] = sgn * std::sin(two_pi * i / num_bins); }
            rc = upload(h->tw, t);
        }
    } else {
        // kiss_fft_alloc, kiss_fft.c:339-368: Q15 twiddles floor(.5 + 32767*cos/sin(phase))
        h->kind = q15_global ? pcx_fft::Q15_GLOBAL : pow2 ? pcx_fft::Q15_POW2 : pcx_fft::MIXED;
        std::vector<int16_t> t(2 * num_bins);
        for (size_t i = 0; i < num_bins; i++) {
            const double pi = 3.141592653589793238462643383279502884197169399375105820974944;
            double phase = -2 * pi * (double)i / (double)num_bins;
            if (h->inverse) phase *= -1;
            t[2 * i] = (int16_t)std::floor(.5 + 32767 * std::cos(phase));
            t[2 * i + 1] = (int16_t)std::floor(.5 + 32767 * std::sin(phase));
        }
        rc = upload(h->tw, t);
        if (rc == PCX_OK && pow2 && num_bins <= 65536 && !q15_global) {
            // the leaf gather of kf_work (kiss_fft.c:276-280): position sum q_s*m_s <- input index sum q_s*fstride_s
            std::vector<uint16_t> perm(num_bins);
            for (size_t pos = 0; pos < num_bins; pos++) {
                size_t rem = pos, m = num_bins, fstride = 1, idx = 0;
                for (size_t si = 0; si < h->radix.size(); si++) {
                    const size_t p = (size_t)h->radix[si];
                    m /= p;
                    const size_t q = rem / m;
                    rem -= q * m;
                    idx += q * fstride;
                    fstride *= p;
                }
                perm[pos] = (uint16_t)idx;
            }
            rc = upload(h->perm, perm);
        }
    }
    if (rc == PCX_OK && (h->kind == pcx_fft::MIXED || h->kind == pcx_fft::SMOOTH)) {
        // inverse of kf_work's leaf gather (kiss_fft.c:276-280, kissfft.hh:94-98): input index sum q_s*fstride_s lands at
        // position sum q_s*m_s; the mixed-radix kernel reads a frame contiguously and scatters it into LDS with this table
        std::vector<uint16_t> iperm(num_bins);
        for (size_t pos = 0; pos < num_bins; pos++) {
            size_t rem = pos, m = num_bins, fstride = 1, idx = 0;
            for (size_t si = 0; si < h->radix.size(); si++) {
                const size_t p = (size_t)h->radix[si];
                m /= p;
                const size_t q = rem / m;
                rem -= q * m;
                idx += q * fstride;
                fstride *= p;
            }
            iperm[idx] = (uint16_t)pos;
        }
        rc = upload(h->perm, iperm);
    }
    if (rc != PCX_OK) { delete h; return rc; }
    *out = h;
    return PCX_OK;
}
int pcx_fft_destroy(pcx_fft *h) { delete h; return PCX_OK; }

// The plans that go through workspaces (four-step, chirp-z) take a long call in batches of frames, so that the workspaces stay
// at kFftWorkspaceCap bytes each whatever the call: a 34 GB call of 20486-bin frames would otherwise ask for 2 x 110 GB
// (tests/test_huge_gpu.py).  A batch of that size is still tens of thousands of workgroups per launch.
constexpr size_t kFftWorkspaceCap = (size_t)1 << 30;
static int fft_transform_batch(pcx_fft *h, const void *in_dev, void *out_dev, size_t nframes, void *stream);

int pcx_fft_transform_dev(pcx_fft *h, const void *in_dev, void *out_dev, size_t nframes, void *stream)
{
    PCX_TRACE();
    PCX_CHECK_ARG(h, "null handle");
    DeviceScope dev_scope(h->cx.device);
    if (nframes == 0) return PCX_OK;
    PCX_CHECK_ARG(in_dev && out_dev, "null buffer");
    PCX_TRY(ctx_enter(h->cx, as_stream(stream)));
    if (h->kind == pcx_fft::FOURSTEP_SHORT || h->kind == pcx_fft::FOURSTEP || h->kind == pcx_fft::BLUESTEIN || h->kind == pcx_fft::Q15_GLOBAL) {
        const size_t esz = 2 * (size_t)scalar_bytes(h->scalar);
        const size_t ws_frame = (h->kind == pcx_fft::BLUESTEIN ? h->n2 : h->nbins) * esz;   // workspace bytes per frame
        size_t batch = kFftWorkspaceCap / ws_frame;
        if (batch < 1) batch = 1;
        for (size_t f = 0; f < nframes; f += batch) {
            const size_t nf = nframes - f < batch ? nframes - f : batch;
            PCX_TRY(fft_transform_batch(h, static_cast<const char *>(in_dev) + f * h->nbins * esz, static_cast<char *>(out_dev) + f * h->nbins * esz, nf, stream));
        }
        return PCX_OK;
    }
    return fft_transform_batch(h, in_dev, out_dev, nframes, stream);
}

static int fft_transform_batch(pcx_fft *h, const void *in_dev, void *out_dev, size_t nframes, void *stream)
{
    hipStream_t st = as_stream(stream);
    switch (h->kind) {
    case pcx_fft::IDENTITY:  // DFT of one point is the identity (kissfft leaf copy, kissfft.hh:94-98) -- except in Q15
        if (h->scalar == PCX_I16) return launch_fft_q15_one(in_dev, out_dev, nframes, st);
        PCX_HIP(hipMemcpyAsync(out_dev, in_dev, nframes * 2 * (size_t)scalar_bytes(h->scalar), hipMemcpyDeviceToDevice, st));
        return PCX_OK;
    case pcx_fft::R16_4096:
        // the radix-16 family's kernel at 12 bits: no register prefetch, no dealer, four frames per workgroup and the hardware
        // dispatcher doing the balancing -- 0.74 -> 0.78 of the HBM peak on 65,536 frames against the dedicated persistent kernel
        // (tools/ab_fft4096_family.sh, profiles/r02/ab_fft4096_family.txt), which stays in the diagnostic library for that A/B
        if (PCX_ENV_SET("PCX_FFT4096_DEDICATED")) return launch_fft4096_cf32(in_dev, out_dev, nframes, h->inverse != 0, h->tw.p, h->sched.p, st);
        return launch_fft_r16_cf32(in_dev, out_dev, 12, nframes, h->inverse != 0, h->tw.p, st);
    case pcx_fft::R16:
        return h->scalar == PCX_F64 ? launch_fft_r16_cf64(in_dev, out_dev, h->log2n, nframes, h->inverse != 0, h->tw.p, st)
                                    : launch_fft_r16_cf32(in_dev, out_dev, h->log2n, nframes, h->inverse != 0, h->tw.p, st);
    case pcx_fft::POW2:
        return h->scalar == PCX_F32 ? launch_fft_pow2_cf32(in_dev, out_dev, h->nbins, nframes, h->inverse != 0, h->tw.p, st)
                                    : launch_fft_pow2_cf64(in_dev, out_dev, h->nbins, nframes, h->inverse != 0, h->tw.p, st);
    case pcx_fft::Q15_GLOBAL:
        PCX_TRY(h->ws1.ensure(nframes * h->nbins * 4));
        return launch_fft_q15_global(in_dev, out_dev, h->ws1.p, h->nbins, nframes, h->inverse != 0, h->tw.p, h->radix.data(), (int)h->radix.size(), st);
    case pcx_fft::Q15_POW2:
        return launch_fft_q15(in_dev, out_dev, h->nbins, nframes, h->inverse != 0, h->tw.p, h->perm.p, h->radix.data(), (int)h->radix.size(), st);
    case pcx_fft::FOURSTEP_SHORT: {
        const bool f64 = h->scalar == PCX_F64;
        const size_t bytes = nframes * h->nbins * (f64 ? 16 : 8);
        PCX_TRY(h->ws1.ensure(bytes));
        // columns of the n1 x n2 view (transform along n1, twiddle), then rows of n2 into natural order
        PCX_TRY((f64 ? launch_fft_columns_f64 : launch_fft_columns)(in_dev, h->ws1.p, h->n1 == 128 ? 7 : 8, h->n2, nframes, h->inverse != 0, h->tw1.p, st));
        if (h->n2 <= 256) {
            int l2 = 0;
            while (((size_t)1 << l2) < h->n2) l2++;
            return (f64 ? launch_fft_rows_transposed_f64 : launch_fft_rows_transposed)(h->ws1.p, out_dev, h->n1, l2, nframes, h->inverse != 0, h->tw2.p, st);
        }
        PCX_TRY(h->ws2.ensure(bytes));
        PCX_TRY(pcx_fft_transform_dev(h->sub2, h->ws1.p, h->ws2.p, nframes * h->n1, stream));
        return launch_transpose(h->scalar, h->ws2.p, out_dev, h->n1, h->n2, nframes, 0, st);
    }
    case pcx_fft::FOURSTEP: {
        const size_t bytes = nframes * h->nbins * 2 * (size_t)scalar_bytes(h->scalar);
        PCX_TRY(h->ws1.ensure(bytes));
        PCX_TRY(h->ws2.ensure(bytes));
        // [F][n1][n2] -> [F][n2][n1]; n2*F transforms of n1; twiddle + back to [F][n1][n2]; n1*F transforms of n2; -> [F][n2][n1] = natural order
        PCX_TRY(launch_transpose(h->scalar, in_dev, h->ws1.p, h->n1, h->n2, nframes, 0, st));
        PCX_TRY(pcx_fft_transform_dev(h->sub1, h->ws1.p, h->ws2.p, nframes * h->n2, stream));
        PCX_TRY(launch_transpose(h->scalar, h->ws2.p, h->ws1.p, h->n2, h->n1, nframes, h->inverse ? 2 : 1, st));
        PCX_TRY(pcx_fft_transform_dev(h->sub2, h->ws1.p, h->ws2.p, nframes * h->n1, stream));
        return launch_transpose(h->scalar, h->ws2.p, out_dev, h->n1, h->n2, nframes, 0, st);
    }
    case pcx_fft::BLUESTEIN: {
        const size_t N = h->nbins, M = h->n2, esz = 2 * (size_t)scalar_bytes(h->scalar);
        PCX_TRY(h->ws1.ensure(nframes * M * esz));
        PCX_TRY(h->ws2.ensure(nframes * M * esz));
        PCX_TRY(launch_bluestein_pre(h->scalar, in_dev, h->ws1.p, h->tw.p, N, M, nframes, h->inverse != 0, st));
        PCX_TRY(pcx_fft_transform_dev(h->sub1, h->ws1.p, h->ws2.p, nframes, stream));
        PCX_TRY(launch_bluestein_mul(h->scalar, h->ws2.p, h->tw1.p, M, nframes, st));
        PCX_TRY(pcx_fft_transform_dev(h->sub2, h->ws2.p, h->ws1.p, nframes, stream));
        return launch_bluestein_post(h->scalar, h->ws1.p, out_dev, h->tw.p, N, M, nframes, h->inverse != 0, st);
    }
    case pcx_fft::SMOOTH:
        return launch_fft_smooth(h->scalar, in_dev, out_dev, h->nbins, nframes, h->inverse != 0, h->tw.p, h->perm.p, h->radix.data(), (int)h->radix.size(), st);
    case pcx_fft::MIXED:
        return launch_fft_mixed(h->scalar, in_dev, out_dev, h->nbins, nframes, h->inverse != 0, h->tw.p, h->perm.p, h->radix.data(), (int)h->radix.size(), st);
    }
    return PCX_ERR_STATE;
}
int pcx_fft_transform(pcx_fft *h, const void *in, void *out, size_t nframes)
{
    PCX_TRACE();
    PCX_CHECK_ARG(h, "null handle");
    DeviceScope dev_scope(h->cx.device);
    if (nframes == 0) return PCX_OK;
    PCX_CHECK_ARG(in && out, "null buffer");
    const size_t bytes = nframes * h->nbins * 2 * (size_t)scalar_bytes(h->scalar);
    hipStream_t st;
    PCX_TRY(ctx_own_stream(h->cx, &st));
    const void *din; void *dout; bool staged;
    PCX_TRY(stage_reserve(out, bytes, h->wsOut));
    PCX_TRY(stage_in(in, bytes, h->wsIn, st, &din));
    PCX_TRY(stage_out_begin(out, bytes, h->wsOut, &dout, &staged));
    PCX_TRY(pcx_fft_transform_dev(h, din, dout, nframes, st));
    return stage_out_end(out, bytes, h->wsOut, staged, st);
}

/* ===================================================================== *
 *  FreqDemod
 * ===================================================================== */
struct pcx_freqdemod {
    ExecCtx cx;
    int scalar = PCX_F32;
    DevBuf prev;  // two complex slots (ping-pong), holds _prev = conj(last input)
    int cur = 0;
    StageBuf wsIn, wsOut;
};
int pcx_freqdemod_create(int scalar, pcx_freqdemod **out)
{
    PCX_CHECK_ARG(out, "null out");
    PCX_CHECK_ARG(valid_scalar(scalar), "FreqDemodFactory: unsupported types (scalar %d)", scalar);
    pcx_freqdemod *h = new (std::nothrow) pcx_freqdemod();
    if (!h) { set_error("out of memory"); return PCX_ERR_STATE; }
    h->scalar = scalar;
    DeviceScope dev_scope(h->cx.device);
    int rc = h->prev.ensure_zeroed(64);
    if (rc != PCX_OK) { delete h; return rc; }
    *out = h;
    return PCX_OK;
}
int pcx_freqdemod_destroy(pcx_freqdemod *h) { delete h; return PCX_OK; }
int pcx_freqdemod_reset(pcx_freqdemod *h)
{
    PCX_CHECK_ARG(h, "null handle");
    DeviceScope dev_scope(h->cx.device);
    // _prev = 0, FreqDemod.cpp:46 -- enqueued behind the handle's previous call (its kernel still reads/writes prev) and
    // ahead of the next one, whatever stream that arrives on (ctx_enter)
    hipStream_t st = h->cx.have_last ? h->cx.last : nullptr;
    if (!h->cx.have_last) PCX_TRY(ctx_own_stream(h->cx, &st));
    PCX_TRY(ctx_enter(h->cx, st));
    PCX_TRY(launch_zero_words(h->prev.p, 16, st));   // (a kernel, not hipMemsetAsync: see launch_zero_words)
    h->cur = 0;
    return PCX_OK;
}
#ifdef PCX_DIAG
// (diagnostic library only) the 64 bytes of carried state and the slot the next call reads, after a device synchronise
extern "C" __attribute__((visibility("default"))) int pcx_diag_freqdemod_state(pcx_freqdemod *h, void *out64, int *cur)
{
    if (!h || !out64 || !cur) return PCX_ERR_ARG;
    PCX_HIP(hipDeviceSynchronize());
    PCX_HIP(hipMemcpy(out64, h->prev.p, 64, hipMemcpyDeviceToHost));
    *cur = h->cur;
    return PCX_OK;
}
#endif
int pcx_freqdemod_process_dev(pcx_freqdemod *h, const void *in_dev, void *out_dev, size_t n, void *stream)
{
    PCX_TRACE();
    PCX_CHECK_ARG(h, "null handle");
    DeviceScope dev_scope(h->cx.device);
    if (n == 0) return PCX_OK;
    PCX_CHECK_ARG(in_dev && out_dev, "null buffer");
    char *base = static_cast<char *>(h->prev.p);
    const void *pin = base + 32 * h->cur;
    void *pout = base + 32 * (h->cur ^ 1);
    PCX_TRY(ctx_enter(h->cx, as_stream(stream)));
    PCX_TRY(launch_freqdemod(h->scalar, in_dev, out_dev, n, pin, pout, as_stream(stream)));
    h->cur ^= 1;
    return PCX_OK;
}
int pcx_freqdemod_process(pcx_freqdemod *h, const void *in, void *out, size_t n)
{
    PCX_TRACE();
    PCX_CHECK_ARG(h, "null handle");
    DeviceScope dev_scope(h->cx.device);
    if (n == 0) return PCX_OK;
    PCX_CHECK_ARG(in && out, "null buffer");
    const size_t sb = (size_t)scalar_bytes(h->scalar);
    hipStream_t st;
    PCX_TRY(ctx_own_stream(h->cx, &st));
    const void *din; void *dout; bool staged;
    PCX_TRY(stage_reserve(out, n * sb, h->wsOut));
    PCX_TRY(stage_in(in, n * 2 * sb, h->wsIn, st, &din));
    PCX_TRY(stage_out_begin(out, n * sb, h->wsOut, &dout, &staged));
    {
        LinkBound shape(in, out, nullptr, 64);
        PCX_TRY(pcx_freqdemod_process_dev(h, din, dout, n, st));
    }
    return stage_out_end(out, n * sb, h->wsOut, staged, st);
}

/* ===================================================================== *
 *  stateless maps
 * ===================================================================== */
// host-buffer wrapper of the stateless maps: page-locked buffers are processed in place (device_alias), pageable ones
// staged through a per-THREAD workspace -- the maps have no handle, and a Pothos block calls them from its own actor
// thread -- that belongs to the thread's CURRENT device and owns a non-blocking stream.  When the thread's device changes
// (pcx_set_device) the workspace is released and rebuilt on the new device.
struct MapWs {
    int device = -1;
    hipStream_t st = nullptr;
    StageBuf in, out, in2, out2;
    void drop()
    {
        in.release(); out.release(); in2.release(); out2.release();
        if (st) (void)hipStreamDestroy(st);
        st = nullptr;
        device = -1;
    }
    ~MapWs() { drop(); }
};
static thread_local MapWs g_mapws;
static int map_ws(MapWs **out)
{
    int cur = -1;
    PCX_HIP(hipGetDevice(&cur));
    if (g_mapws.device != cur) {
        if (g_mapws.device >= 0) {   // buffers and stream of the previous device: free them there
            (void)hipSetDevice(g_mapws.device);
            g_mapws.drop();
            PCX_HIP(hipSetDevice(cur));
        }
        g_mapws.device = cur;
    }
    if (!g_mapws.st) PCX_HIP(hipStreamCreateWithFlags(&g_mapws.st, hipStreamNonBlocking));
    *out = &g_mapws;
    return PCX_OK;
}

template <typename F>
static int run_host_map(const void *in, void *out, size_t in_bytes, size_t out_bytes, F &&launch)
{
    if (in_bytes == 0) return PCX_OK;
    PCX_CHECK_ARG(in && out, "null buffer");
    MapWs *ws;
    PCX_TRY(map_ws(&ws));
    const void *din; void *dout; bool staged;
    PCX_TRY(stage_reserve(out, out_bytes, ws->out));
    PCX_TRY(stage_in(in, in_bytes, ws->in, ws->st, &din));
    PCX_TRY(stage_out_begin(out, out_bytes, ws->out, &dout, &staged));
    {
        LinkBound shape(in, out);
        PCX_TRY(launch(din, dout, ws->st));
    }
    return stage_out_end(out, out_bytes, ws->out, staged, ws->st);
}

int pcx_rotate_q_dev(int scalar, double pr, double pi, const pcx_qformat *q, const void *in_dev, void *out_dev, size_t n, void *stream)
{
    PCX_TRACE();
    PCX_CHECK_ARG(valid_scalar(scalar), "rotateFactory: unsupported type (scalar %d)", scalar);
    QFormat qf;
    PCX_TRY(qformat_from_api(q, &qf));
    if (n == 0) return PCX_OK;
    PCX_CHECK_ARG(in_dev && out_dev, "null buffer");
    return launch_rotate(scalar, pr, pi, qf, in_dev, out_dev, n, as_stream(stream));
}
int pcx_rotate_q(int scalar, double pr, double pi, const pcx_qformat *q, const void *in, void *out, size_t n)
{
    PCX_TRACE();
    PCX_CHECK_ARG(valid_scalar(scalar), "rotateFactory: unsupported type (scalar %d)", scalar);
    QFormat qf;
    PCX_TRY(qformat_from_api(q, &qf));
    const size_t b = n * 2 * (size_t)scalar_bytes(scalar);
    return run_host_map(in, out, b, b, [&](const void *di, void *dout, hipStream_t st) { return launch_rotate(scalar, pr, pi, qf, di, dout, n, st); });
}
int pcx_rotate_dev(int scalar, double pr, double pi, const void *in_dev, void *out_dev, size_t n, void *stream)
{
    return pcx_rotate_q_dev(scalar, pr, pi, nullptr, in_dev, out_dev, n, stream);
}
int pcx_rotate(int scalar, double pr, double pi, const void *in, void *out, size_t n) { return pcx_rotate_q(scalar, pr, pi, nullptr, in, out, n); }
int pcx_scale_q_dev(int scalar, int is_complex, double factor, const pcx_qformat *q, const void *in_dev, void *out_dev, size_t n, void *stream)
{
    PCX_TRACE();
    PCX_CHECK_ARG(valid_scalar(scalar), "scaleFactory: unsupported type (scalar %d)", scalar);
    QFormat qf;
    PCX_TRY(qformat_from_api(q, &qf));
    if (n == 0) return PCX_OK;
    PCX_CHECK_ARG(in_dev && out_dev, "null buffer");
    return launch_scale(scalar, is_complex, factor, qf, in_dev, out_dev, n, as_stream(stream));
}
int pcx_scale_q(int scalar, int is_complex, double factor, const pcx_qformat *q, const void *in, void *out, size_t n)
{
    PCX_TRACE();
    PCX_CHECK_ARG(valid_scalar(scalar), "scaleFactory: unsupported type (scalar %d)", scalar);
    QFormat qf;
    PCX_TRY(qformat_from_api(q, &qf));
    const size_t b = n * (is_complex ? 2 : 1) * (size_t)scalar_bytes(scalar);
    return run_host_map(in, out, b, b, [&](const void *di, void *dout, hipStream_t st) { return launch_scale(scalar, is_complex, factor, qf, di, dout, n, st); });
}
int pcx_scale_dev(int scalar, int is_complex, double factor, const void *in_dev, void *out_dev, size_t n, void *stream)
{
    return pcx_scale_q_dev(scalar, is_complex, factor, nullptr, in_dev, out_dev, n, stream);
}
int pcx_scale(int scalar, int is_complex, double factor, const void *in, void *out, size_t n) { return pcx_scale_q(scalar, is_complex, factor, nullptr, in, out, n); }
int pcx_abs_dev(int scalar, int is_complex, const void *in_dev, void *out_dev, size_t n, void *stream)
{
    PCX_TRACE();
    PCX_CHECK_ARG(valid_scalar(scalar), "absFactory: unsupported type (scalar %d)", scalar);
    if (n == 0) return PCX_OK;
    PCX_CHECK_ARG(in_dev && out_dev, "null buffer");
    return launch_abs(scalar, is_complex, in_dev, out_dev, n, as_stream(stream));
}
int pcx_abs(int scalar, int is_complex, const void *in, void *out, size_t n)
{
    PCX_TRACE();
    PCX_CHECK_ARG(valid_scalar(scalar), "absFactory: unsupported type (scalar %d)", scalar);
    const size_t sb = (size_t)scalar_bytes(scalar);
    return run_host_map(in, out, n * (is_complex ? 2 : 1) * sb, n * sb,
                        [&](const void *di, void *dout, hipStream_t st) { return launch_abs(scalar, is_complex, di, dout, n, st); });
}
int pcx_conj_dev(int scalar, const void *in_dev, void *out_dev, size_t n, void *stream)
{
    PCX_TRACE();
    PCX_CHECK_ARG(valid_scalar(scalar), "conjugateFactory: unsupported type (scalar %d)", scalar);
    if (n == 0) return PCX_OK;
    PCX_CHECK_ARG(in_dev && out_dev, "null buffer");
    return launch_conj(scalar, in_dev, out_dev, n, as_stream(stream));
}
int pcx_conj(int scalar, const void *in, void *out, size_t n)
{
    PCX_TRACE();
    PCX_CHECK_ARG(valid_scalar(scalar), "conjugateFactory: unsupported type (scalar %d)", scalar);
    const size_t b = n * 2 * (size_t)scalar_bytes(scalar);
    return run_host_map(in, out, b, b, [&](const void *di, void *dout, hipStream_t st) { return launch_conj(scalar, di, dout, n, st); });
}

int pcx_angle_dev(int scalar, const void *in_dev, void *out_dev, size_t n, void *stream)
{
    PCX_TRACE();
    PCX_CHECK_ARG(valid_scalar(scalar), "angleFactory: unsupported type (scalar %d)", scalar);
    if (n == 0) return PCX_OK;
    PCX_CHECK_ARG(in_dev && out_dev, "null buffer");
    return launch_angle(scalar, in_dev, out_dev, n, as_stream(stream));
}
int pcx_angle(int scalar, const void *in, void *out, size_t n)
{
    PCX_TRACE();
    PCX_CHECK_ARG(valid_scalar(scalar), "angleFactory: unsupported type (scalar %d)", scalar);
    const size_t sb = (size_t)scalar_bytes(scalar);
    return run_host_map(in, out, n * 2 * sb, n * sb, [&](const void *di, void *dout, hipStream_t st) { return launch_angle(scalar, di, dout, n, st); });
}

int pcx_arith_dev(int scalar, int is_complex, int op, const void *in0_dev, const void *in1_dev, void *out_dev, size_t n, void *stream)
{
    PCX_TRACE();
    PCX_CHECK_ARG(valid_arith_scalar(scalar) && op >= PCX_ARITH_ADD && op <= PCX_ARITH_DIV,
                  "arithmeticFactory: unsupported args (scalar %d, op %d)", scalar, op);
    if (n == 0) return PCX_OK;
    PCX_CHECK_ARG(in0_dev && in1_dev && out_dev, "null buffer");
    return launch_arith(scalar, is_complex, op, in0_dev, in1_dev, out_dev, n, as_stream(stream));
}
int pcx_arith(int scalar, int is_complex, int op, const void *in0, const void *in1, void *out, size_t n)
{
    PCX_TRACE();
    PCX_CHECK_ARG(valid_arith_scalar(scalar) && op >= PCX_ARITH_ADD && op <= PCX_ARITH_DIV,
                  "arithmeticFactory: unsupported args (scalar %d, op %d)", scalar, op);
    if (n == 0) return PCX_OK;
    PCX_CHECK_ARG(in0 && in1 && out, "null buffer");
    const size_t b = n * (is_complex ? 2 : 1) * (size_t)scalar_bytes(scalar);
    MapWs *ws;
    PCX_TRY(map_ws(&ws));
    const void *d0, *d1; void *dout; bool staged;
    PCX_TRY(stage_reserve(in1, b, ws->in2));
    PCX_TRY(stage_reserve(out, b, ws->out));
    PCX_TRY(stage_in(in0, b, ws->in, ws->st, &d0));
    PCX_TRY(stage_in(in1, b, ws->in2, ws->st, &d1));
    PCX_TRY(stage_out_begin(out, b, ws->out, &dout, &staged));
    {
        LinkBound shape(in0, in1, out);
        PCX_TRY(launch_arith(scalar, is_complex, op, d0, d1, dout, n, ws->st));
    }
    return stage_out_end(out, b, ws->out, staged, ws->st);
}
int pcx_split_complex_dev(int scalar, const void *in_dev, void *re_dev, void *im_dev, size_t n, void *stream)
{
    PCX_TRACE();
    PCX_CHECK_ARG(valid_scalar(scalar), "splitComplexFactory: unsupported type (scalar %d)", scalar);
    if (n == 0) return PCX_OK;
    PCX_CHECK_ARG(in_dev && re_dev && im_dev, "null buffer");
    return launch_split_complex(scalar, in_dev, re_dev, im_dev, n, as_stream(stream));
}
int pcx_split_complex(int scalar, const void *in, void *re, void *im, size_t n)
{
    PCX_TRACE();
    PCX_CHECK_ARG(valid_scalar(scalar), "splitComplexFactory: unsupported type (scalar %d)", scalar);
    if (n == 0) return PCX_OK;
    PCX_CHECK_ARG(in && re && im, "null buffer");
    const size_t b = n * (size_t)scalar_bytes(scalar);
    MapWs *ws;
    PCX_TRY(map_ws(&ws));
    const void *din; void *dre, *dim; bool sre, sim;
    PCX_TRY(stage_reserve(re, b, ws->out));
    PCX_TRY(stage_reserve(im, b, ws->out2));
    PCX_TRY(stage_in(in, 2 * b, ws->in, ws->st, &din));
    PCX_TRY(stage_out_begin(re, b, ws->out, &dre, &sre));
    PCX_TRY(stage_out_begin(im, b, ws->out2, &dim, &sim));
    {
        LinkBound shape(in, re, im);
        PCX_TRY(launch_split_complex(scalar, din, dre, dim, n, ws->st));
    }
    PCX_TRY(stage_out_end(re, b, ws->out, sre, ws->st));
    return stage_out_end(im, b, ws->out2, sim, ws->st);
}
int pcx_combine_complex_dev(int scalar, const void *re_dev, const void *im_dev, void *out_dev, size_t n, void *stream)
{
    PCX_TRACE();
    PCX_CHECK_ARG(valid_scalar(scalar), "combineComplexFactory: unsupported type (scalar %d)", scalar);
    if (n == 0) return PCX_OK;
    PCX_CHECK_ARG(re_dev && im_dev && out_dev, "null buffer");
    return launch_combine_complex(scalar, re_dev, im_dev, out_dev, n, as_stream(stream));
}
int pcx_combine_complex(int scalar, const void *re, const void *im, void *out, size_t n)
{
    PCX_TRACE();
    PCX_CHECK_ARG(valid_scalar(scalar), "combineComplexFactory: unsupported type (scalar %d)", scalar);
    if (n == 0) return PCX_OK;
    PCX_CHECK_ARG(re && im && out, "null buffer");
    const size_t b = n * (size_t)scalar_bytes(scalar);
    MapWs *ws;
    PCX_TRY(map_ws(&ws));
    const void *dre, *dim; void *dout; bool staged;
    PCX_TRY(stage_reserve(im, b, ws->in2));
    PCX_TRY(stage_reserve(out, 2 * b, ws->out));
    PCX_TRY(stage_in(re, b, ws->in, ws->st, &dre));
    PCX_TRY(stage_in(im, b, ws->in2, ws->st, &dim));
    PCX_TRY(stage_out_begin(out, 2 * b, ws->out, &dout, &staged));
    {
        LinkBound shape(re, im, out);
        PCX_TRY(launch_combine_complex(scalar, dre, dim, dout, n, ws->st));
    }
    return stage_out_end(out, 2 * b, ws->out, staged, ws->st);
}
