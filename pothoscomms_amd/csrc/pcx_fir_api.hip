// pcx_fir_api.hip -- the extern "C" boundary (include/pcx.h), part 2: /comms/fir_filter (pcx_fir_*) and the fused
// Rotate -> FIR -> FreqDemod chain (pcx_fmchain_*): tap bookkeeping (FIRFilter.cpp:138-173,327-354), the coefficient tables of
// every plan, the choice of kernel family per call, host-buffer calls.  Host-side only.
#include "pcx_host.hpp"
#include "pcx_tables.hpp"

using namespace pcx;

/* ===================================================================== *
 *  FIR
 * ===================================================================== */
struct pcx_fir {
    ExecCtx cx;
    int scalar = PCX_F32, cplx = 1, ctaps = 1;
    std::vector<double> taps;  // ntaps * (ctaps ? 2 : 1)
    size_t ntaps = 1, M = 1, L = 1, K = 1, inputRequire = 1;
    int algo = PCX_FIR_AUTO, last_algo = 0;
    QFormat qf = kDefaultQFormat;   // integer element types: the floatToQ / fromQ reading (pcx_fir_set_qformat; the process-wide one at creation)
    bool dirty = true;
    DevBuf rowLen, rowTaps, tapsRev, Hspec, tw4096;
    StageBuf wsIn, wsOut;
    DevBuf sched;             // SchedState: dynamic block assignment of the overlap-save kernels (pcx_sched.hpp), zeroed once
    unsigned slots = 1024;    // resident workgroups a persistent launch may take (pcx_shard: several shards on one device share it)
    size_t lead_valid = 0;    // set around the chunks of a drained host call: samples of the same stream in front of the chunk's first
    size_t Kp = 8;
    bool have_ols = false;
    bool have_poly = false;   // frequency-domain rows for L > 1 or M > 1
    DevBuf wsRows;            // interpolation by other factors: one contiguous output row per polyphase row, interleaved afterwards
    bool have_decim = false;  // L = 1, M in {2,4,8,16}: decimation folded into the spectrum (Hdecim)
    bool have_interp = false; // M = 1, L in {2,4,8,16}: replicated spectrum of the short forward transform (Hdecim holds H of all taps)
    DevBuf Hdecim;
    bool have_real_ols = false;   // real float32 stream, real taps, M=L=1
    bool have_interp_f32 = false;  // REAL float32, L > 1, rows of up to 2049 taps: the real float kernel row by row + interleave
    bool have_upols_rows = false;  // complex_float32 / float32, L > 1, polyphase rows of 2050 .. 8193 taps: the partitioned kernel row by row + interleave
    bool have_upols_decim = false; // complex_float32 / float32, L = 1, M > 1, 2049 < K <= 8193: the partitioned kernel with a decimating store
    bool have_ols64 = false;      // complex_float64 stream, M=L=1 (Hspec / tw4096 then hold doubles)
    bool have_ols_int = false;    // complex_int16 / complex_int8 stream, M=L=1: exact integer convolution on the double transform
    bool have_ols_real64 = false; // REAL float64 / int16 / int8 stream (real taps), M=L=1: two real blocks per double transform
    bool have_interp64 = false;   // complex_float64 / int16 / int8, M = 1, L > 1: polyphase rows on the double pipeline (HrowsD) + interleave
    bool have_interp_real = false; // REAL float64 / float32 / int16 / int8, L > 1: the same with the two-real-blocks kernel
    DevBuf HrowsD;
    DevBuf HspecRows;
    int ols_parts = 0;        // complex_float32 / float32, M = L = 1: 0 = fir_ols.hip's 4096 kernels alone, 2 .. 4 = that many tap partitions (fir_ols_part.hip)
    int ols_log2n = 0;        // the double-precision plans: log2 of the block (12, 13)
    bool taps24 = false;      // integer Q taps all fit 24 signed bits (v_mul_i32_i24 path)
    bool taps16 = false;      // complex_int16 / complex_int8 stream, complex taps within +-32767 after floatToQ (v_dot2_i32_i16 path)
    DevBuf tapsP;             // packed (a, -b), (b, a) pairs for that path
};

// Tap partitions of the overlap-save plan for K taps (complex_float32, M = L = 1): 0 = the dedicated 4096-sample kernel alone
// (fir_ols.hip, K <= 2049); P = 2 .. 4 = the same blocks with the taps in P partitions of 2048 (fir_ols_part.hip, 2049 < K <= 8193).
// (Until round 6 longer filters took 8192- / 16384-sample blocks on radix-16 family passes -- 143 / 88 Gsamples/s at 4097 / 8193
// taps against 206 / 167 now, profiles/r06/ab_upols.txt; blocks SHORTER than 4096 never paid either: 0.2413 ms at 2048, 0.2723 at
// 1024 against 0.2246 at 255 taps, round 2.)
static int fir_ols_partitions(size_t K) { return K <= 2049 ? 0 : (int)((K - 1 + 2047) / 2048); }
constexpr size_t kOlsMaxTaps = 8193;
constexpr size_t kRowsWorkspaceCap = (size_t)1 << 30;   // polyphase-row workspace of the interpolating paths (pcx_fir_process_dev)
// complex_float64 (fir_ols_f64.hip): 4096-sample blocks to K = 2049, 8192 to K = 4097; PCX_OLS64_N forces a plan (A/B)
constexpr size_t kOls64MaxTaps = 4097;
// below this many taps the sliding-window kernel is the faster complex_float64 form (tools/sweep_fir_f64.py: 128 vs 112 Gsamples/s at K = 2)
constexpr size_t kOls64MinTaps = 4;
// complex_int16 / complex_int8 on the same pipeline (bit-exact): 166-170 / 128-131 Gsamples/s whatever the tap count, so it
// takes over where the packed dot-product kernel falls below that (tools/sweep_fir_int.py: 164 Gsamples/s at 63 taps, 91 at
// 127, 48 at 255, 12 at 1023); PCX_OLS_INT_MIN overrides (A/B)
static size_t ols_int_min_taps(int scalar)
{
    const size_t forced = (size_t)PCX_ENV_INT("PCX_OLS_INT_MIN", 0);
    return forced ? forced : scalar == PCX_I16 ? 64 : 96;
}

// REAL float64 / int16 / int8 streams on the double pipeline, two real blocks per transform: 234 / 290 / 296 Gsamples/s
// whatever the tap count; the sliding-window kernel is faster below about 24 / 48 / 48 taps (tools/sweep_fir_int.py real:
// float64 278 vs 228 at 16 taps, 202 vs 234 at 32; int16 356 vs 280 at 32, 230 vs 286 at 63); PCX_OLS_REAL_MIN overrides (A/B)
static size_t ols_real64_min_taps(int scalar)
{
    const size_t forced = (size_t)PCX_ENV_INT("PCX_OLS_REAL_MIN", 0);
    return forced ? forced : scalar == PCX_F64 ? 24 : 48;
}
static int fir_ols64_block_log2(size_t K)
{
    const int forced = (int)PCX_ENV_INT("PCX_OLS64_N", 0);
    int l2 = K <= 2049 ? 12 : 13;
    if (forced == 8192) l2 = 13;
    return l2;
}

// FIRFilter::updateInternals, FIRFilter.cpp:327-354 (host mirror; tables uploaded lazily)
static void fir_update_internals(pcx_fir *h)
{
    h->K = h->ntaps / h->L + ((h->ntaps % h->L) == 0 ? 0 : 1);
    h->inputRequire = h->M + (h->K - 1);
    h->dirty = true;
}

template <typename TT>
static int fir_upload_rows(pcx_fir *h, bool integer)
{
    const size_t L = h->L, K = h->K, w = h->ctaps ? 2 : 1;
    std::vector<uint32_t> rowLen(L, 0);
    std::vector<TT> rows(L * K * w, TT(0));
    for (size_t j = 0; j < L; j++) {
        size_t len = 0;
        for (size_t k = 0; k < K; k++) {
            const size_t i = j + k * L;
            if (i >= h->ntaps) continue;
            for (size_t c = 0; c < w; c++) {
                const double t = h->taps[i * w + c];
                rows[(j * K + len) * w + c] = integer ? (TT)float_to_q(t, h->scalar, h->qf) : (TT)t;  // floatToQ<QTapsType>, :348
            }
            len++;
        }
        rowLen[j] = (uint32_t)len;
    }
    PCX_TRY(upload(h->rowLen, rowLen));
    PCX_TRY(upload(h->rowTaps, rows));
    h->taps24 = integer;
    if (integer)
        for (const TT &t : rows)
            if ((long long)t < -(1ll << 23) || (long long)t >= (1ll << 23)) { h->taps24 = false; break; }
    h->taps16 = false;
    if (integer && (h->scalar == PCX_I16 || h->scalar == PCX_I8) && h->cplx && h->ctaps && L == 1 && h->M == 1) {
        bool ok = true;
        for (const TT &t : rows)
            if ((long long)t < -32767 || (long long)t > 32767) { ok = false; break; }
        if (ok) {
            std::vector<uint32_t> packed(2 * K);
            for (size_t k = 0; k < K; k++) {
                const uint32_t a = (uint16_t)(int16_t)rows[2 * k], b = (uint16_t)(int16_t)rows[2 * k + 1];
                const uint32_t nb = (uint16_t)(int16_t)(-(long long)rows[2 * k + 1]);
                packed[2 * k] = a | (nb << 16);        // (a, -b): real part
                packed[2 * k + 1] = b | (a << 16);     // (b,  a): imaginary part
            }
            PCX_TRY(upload(h->tapsP, packed));
            h->taps16 = true;
        }
    }
    return PCX_OK;
}

static bool fir_fast_applicable(const pcx_fir *h) { return h->scalar == PCX_F32 && h->cplx && h->M == 1 && h->L == 1; }

static int fir_sync_tables(pcx_fir *h)
{
    if (!h->dirty) return PCX_OK;
    PCX_TRY(ctx_quiesce(h->cx));   // a kernel of an earlier call may still be reading the tables rewritten below
    switch (h->scalar) {
    case PCX_F32: PCX_TRY(fir_upload_rows<float>(h, false)); break;
    case PCX_F64: PCX_TRY(fir_upload_rows<double>(h, false)); break;
    case PCX_I64: case PCX_I32: PCX_TRY(fir_upload_rows<int64_t>(h, true)); break;
    case PCX_I16: PCX_TRY(fir_upload_rows<int32_t>(h, true)); break;
    case PCX_I8: PCX_TRY(fir_upload_rows<int16_t>(h, true)); break;
    }
    h->have_ols = false;
    if (!h->sched.p) {
        PCX_TRY(h->sched.ensure_zeroed(kSchedBytes));
    }
    if (fir_fast_applicable(h)) {
        const size_t K = h->K;
        // reversed, zero-padded complex taps for the LDS-tiled direct kernel
        h->Kp = (K + 7) / 8 * 8;
        std::vector<float> rev(2 * h->Kp, 0.f);
        for (size_t m = 0; m < K; m++) {
            const size_t k = K - 1 - m;
            rev[2 * m] = (float)(h->ctaps ? h->taps[2 * k] : h->taps[k]);
            rev[2 * m + 1] = h->ctaps ? (float)h->taps[2 * k + 1] : 0.f;
        }
        PCX_TRY(upload(h->tapsRev, rev));
        if (K <= kOlsMaxTaps) {
            std::vector<std::complex<double>> hq(K);
            for (size_t k = 0; k < K; k++)   // floatToQ<QTapsType>: narrowed to float first (FIRFilter.cpp:348)
                hq[k] = std::complex<double>((double)(float)(h->ctaps ? h->taps[2 * k] : h->taps[k]),
                                             h->ctaps ? (double)(float)h->taps[2 * k + 1] : 0.0);
            h->ols_parts = fir_ols_partitions(K);
            if (h->ols_parts == 0) {   // the dedicated 4096-sample kernel (fir_ols.hip)
                PCX_TRY(upload(h->Hspec, make_hspec4096(hq)));
            } else {                   // the same blocks, the taps in partitions (fir_ols_part.hip)
                PCX_TRY(upload(h->Hspec, make_hparts(hq, h->ols_parts)));
            }
            PCX_TRY(upload(h->tw4096, make_tw4096()));
            h->have_ols = true;
        }
    }
    h->have_ols64 = false;
    if (h->scalar == PCX_F64 && h->cplx && h->M <= 65535 && h->L == 1 && h->K >= 2 && h->K <= kOls64MaxTaps) {   // M > 1: decimate on store
        // complex_float64: the same frequency-domain evaluation in double (fir_ols_f64.hip)
        std::vector<std::complex<double>> hq(h->K);
        for (size_t k = 0; k < h->K; k++) hq[k] = std::complex<double>(h->ctaps ? h->taps[2 * k] : h->taps[k], h->ctaps ? h->taps[2 * k + 1] : 0.0);
        h->ols_log2n = fir_ols64_block_log2(h->K);
        PCX_TRY(upload(h->Hspec, make_hspec<double>(hq, (size_t)1 << h->ols_log2n)));
        PCX_TRY(upload(h->tw4096, make_tw_ols64(h->ols_log2n)));
        h->have_ols64 = true;
    }
    h->have_ols_int = false;
    if ((h->scalar == PCX_I16 || h->scalar == PCX_I8) && h->cplx && h->M <= 65535 && h->L == 1 && h->K >= 2 && h->K <= kOls64MaxTaps) {
        // the Q-format taps exactly as the time-domain kernels use them (floatToQ<QTapsType>, FIRFilter.cpp:348), as doubles;
        // the double transform reproduces the integer convolution bit for bit while ||h_q||_2 < 2^22 (fir_ols_f64.hip)
            auto tq = [&](double t) { return h->scalar == PCX_I16 ? (double)(int32_t)float_to_q(t, h->scalar, h->qf) : (double)(int16_t)float_to_q(t, h->scalar, h->qf); };
        std::vector<std::complex<double>> hq(h->K);
        double norm2 = 0;
        for (size_t k = 0; k < h->K; k++) {
            hq[k] = std::complex<double>(tq(h->ctaps ? h->taps[2 * k] : h->taps[k]), h->ctaps ? tq(h->taps[2 * k + 1]) : 0.0);
            norm2 += std::norm(hq[k]);
        }
        if (norm2 < 17592186044416.0) {   // 2^44
            h->ols_log2n = fir_ols64_block_log2(h->K);
            PCX_TRY(upload(h->Hspec, make_hspec<double>(hq, (size_t)1 << h->ols_log2n)));
            PCX_TRY(upload(h->tw4096, make_tw_ols64(h->ols_log2n)));
            h->have_ols_int = true;
        }
    }
    h->have_interp64 = false;
    if ((h->scalar == PCX_F64 || h->scalar == PCX_I16 || h->scalar == PCX_I8) && h->cplx && h->M <= 65535 && h->L > 1 && h->L <= 64 && h->K >= 2 &&
        h->K <= 2049) {   // M > 1: rational resampling, the interleaving pass keeps one position in M
        // interpolating filters of these types: every polyphase row h_j[k] = taps[j + k L] (FIRFilter.cpp:341-350) through the
        // double-precision pipeline into a contiguous workspace row, then one interleaving pass; integers stay exact row by row
            auto tq = [&](double t) {
            return h->scalar == PCX_F64 ? t : h->scalar == PCX_I16 ? (double)(int32_t)float_to_q(t, h->scalar, h->qf) : (double)(int16_t)float_to_q(t, h->scalar, h->qf);
        };
        std::vector<double> rows(h->L * 2 * 4096);
        bool ok = true;
        for (size_t jr = 0; jr < h->L && ok; jr++) {
            std::vector<std::complex<double>> hq;
            double norm2 = 0;
            for (size_t k = 0; k < h->K; k++) {
                const size_t i = jr + k * h->L;
                if (i >= h->ntaps) continue;
                hq.push_back(std::complex<double>(tq(h->ctaps ? h->taps[2 * i] : h->taps[i]), h->ctaps ? tq(h->taps[2 * i + 1]) : 0.0));
                norm2 += std::norm(hq.back());
            }
            if (h->scalar != PCX_F64 && norm2 >= 17592186044416.0) ok = false;
            if (hq.empty()) hq.push_back(0.0);
            const std::vector<double> H = make_hspec<double>(hq, 4096);
            std::copy(H.begin(), H.end(), rows.begin() + jr * 2 * 4096);
        }
        if (ok) {
            PCX_TRY(upload(h->HrowsD, rows));
            PCX_TRY(upload(h->tw4096, make_tw_ols64(12)));
            h->ols_log2n = 12;
            h->have_interp64 = true;
        }
    }
    h->have_interp_real = false;
    // (real float32 has the float kernel for its rows: have_interp_f32 below)
    if ((h->scalar == PCX_F64 || h->scalar == PCX_I16 || h->scalar == PCX_I8) && !h->cplx && h->M <= 65535 && h->L > 1 &&
        h->L <= 64 && h->K >= 2 && h->K <= 2049) {
            auto tq = [&](double t) {
            return h->scalar == PCX_F64 ? t : h->scalar == PCX_F32 ? (double)(float)t
                 : h->scalar == PCX_I16 ? (double)(int32_t)float_to_q(t, h->scalar, h->qf) : (double)(int16_t)float_to_q(t, h->scalar, h->qf);
        };
        const bool integer = h->scalar == PCX_I16 || h->scalar == PCX_I8;
        std::vector<double> rows(h->L * 2 * 4096);
        bool ok = true;
        for (size_t jr = 0; jr < h->L && ok; jr++) {
            std::vector<std::complex<double>> hq;
            double norm2 = 0;
            for (size_t k = 0; k < h->K; k++) {
                const size_t i = jr + k * h->L;
                if (i >= h->ntaps) continue;
                hq.push_back(std::complex<double>(tq(h->taps[i]), 0.0));
                norm2 += std::norm(hq.back());
            }
            if (integer && norm2 >= 17592186044416.0) ok = false;
            if (hq.empty()) hq.push_back(0.0);
            const std::vector<double> H = make_hspec<double>(hq, 4096);
            std::copy(H.begin(), H.end(), rows.begin() + jr * 2 * 4096);
        }
        if (ok) {
            PCX_TRY(upload(h->HrowsD, rows));
            PCX_TRY(upload(h->tw4096, make_tw_ols64(12)));
            h->ols_log2n = 12;
            h->have_interp_real = true;
        }
    }
    h->have_interp_f32 = false;
    if (h->scalar == PCX_F32 && !h->cplx && h->M <= 65535 && h->L > 1 && h->L <= 64 && h->K >= 2 && h->K <= 2049) {
        // interpolating REAL float32 filters: every polyphase row through the real float kernel (two real blocks per complex transform,
        // fir_ols.hip) into a workspace row, then the interleaving pass -- instead of the double-precision rows they had shared with
        // the integer types
        std::vector<float> rows(h->L * 2 * 4096);
        for (size_t jr = 0; jr < h->L; jr++) {
            std::vector<std::complex<double>> hq;
            for (size_t k = 0; k < h->K; k++) {
                const size_t i = jr + k * h->L;
                if (i >= h->ntaps) continue;
                hq.push_back(std::complex<double>((double)(float)h->taps[i], 0.0));
            }
            if (hq.empty()) hq.push_back(0.0);
            const std::vector<float> H = make_hspec4096(hq);
            std::copy(H.begin(), H.end(), rows.begin() + jr * 2 * 4096);
        }
        PCX_TRY(upload(h->HspecRows, rows));
        PCX_TRY(upload(h->tw4096, make_tw4096()));
        h->have_interp_f32 = true;
    }
    h->have_ols_real64 = false;
    // (real float32 streams have the float kernels: undecimated fir_ols.hip, decimating the partitioned kernel, have_upols_decim below)
    if ((h->scalar == PCX_F64 || h->scalar == PCX_I16 || h->scalar == PCX_I8) && !h->cplx && h->M <= 65535 &&
        h->L == 1 && h->K >= 2 && h->K <= kOls64MaxTaps) {
            std::vector<std::complex<double>> hq(h->K);
        double norm2 = 0;
        for (size_t k = 0; k < h->K; k++) {
            const double t = h->taps[k];
            hq[k] = h->scalar == PCX_F64 ? t : h->scalar == PCX_F32 ? (double)(float)t
                    : h->scalar == PCX_I16 ? (double)(int32_t)float_to_q(t, h->scalar, h->qf) : (double)(int16_t)float_to_q(t, h->scalar, h->qf);
            norm2 += std::norm(hq[k]);
        }
        if (h->scalar == PCX_F64 || h->scalar == PCX_F32 || norm2 < 17592186044416.0) {   // integers: ||h_q||_2 < 2^22 keeps the rounded sums exact
            h->ols_log2n = h->K <= 2049 ? 12 : 13;
            PCX_TRY(upload(h->Hspec, make_hspec<double>(hq, (size_t)1 << h->ols_log2n)));
            PCX_TRY(upload(h->tw4096, make_tw_ols64(h->ols_log2n)));
            h->have_ols_real64 = true;
        }
    }
    h->have_real_ols = false;
    if (h->scalar == PCX_F32 && !h->cplx && h->M == 1 && h->L == 1 && h->K <= kOlsMaxTaps) {
        // two real blocks per complex transform (fir_ols.hip); beyond 2049 taps the two halves of the call side by side through the
        // partitioned kernel (fir_ols_part.hip)
        std::vector<std::complex<double>> hq(h->K);
        for (size_t k = 0; k < h->K; k++) hq[k] = std::complex<double>((double)(float)h->taps[k], 0.0);
        h->ols_parts = fir_ols_partitions(h->K);
        if (h->ols_parts == 0) PCX_TRY(upload(h->Hspec, make_hspec4096(hq)));
        else PCX_TRY(upload(h->Hspec, make_hparts(hq, h->ols_parts)));
        PCX_TRY(upload(h->tw4096, make_tw4096()));
        h->have_real_ols = true;
    }
    h->have_upols_decim = false;
    if (h->scalar == PCX_F32 && h->L == 1 && h->M > 1 && h->M <= 65535 && h->K <= kOlsMaxTaps && (h->K > 2049 || (!h->cplx && h->K >= 2))) {
        // long DECIMATING filters, complex_float32 or float32: the partitioned kernel at the full rate, one output in M stored
        // (fir_ols_part.hip) -- the time-domain tile they fell to runs at 1-2 Gsamples/s of input at these tap counts.  REAL float32
        // decimators of any length take it as well (one partition up to 2049 taps): 300 against the 155 Gsamples/s of the
        // double-precision pipeline they shared with the integer types
        std::vector<std::complex<double>> hq(h->K);
        for (size_t k = 0; k < h->K; k++)
            hq[k] = std::complex<double>((double)(float)(h->ctaps ? h->taps[2 * k] : h->taps[k]), h->ctaps ? (double)(float)h->taps[2 * k + 1] : 0.0);
        h->ols_parts = std::max(1, fir_ols_partitions(h->K));
        PCX_TRY(upload(h->Hspec, make_hparts(hq, h->ols_parts)));
        PCX_TRY(upload(h->tw4096, make_tw4096()));
        h->have_upols_decim = true;
    }
    h->have_upols_rows = false;
    if (h->scalar == PCX_F32 && h->L > 1 && h->L <= 64 && h->M <= 65535 && h->K > 2049 && h->K <= kOlsMaxTaps) {
        // INTERPOLATING filters whose polyphase rows h_j[k] = taps[j + k L] (FIRFilter.cpp:341-350) are longer than 2049 taps: every row
        // through the partitioned kernel at the input rate into a workspace row, then the interleaving pass (which also keeps one
        // position in M) -- as the shorter rows go through fir_ols.hip.  A row the tap vector leaves short ends in zeros.
        const int parts = fir_ols_partitions(h->K);
        const size_t tb = fir_upols_table_bytes(parts) / sizeof(float);
        std::vector<float> rows(h->L * tb);
        for (size_t jr = 0; jr < h->L; jr++) {
            std::vector<std::complex<double>> hq(h->K, 0.0);
            for (size_t k = 0; k < h->K; k++) {
                const size_t i = jr + k * h->L;
                if (i >= h->ntaps) break;
                hq[k] = std::complex<double>((double)(float)(h->ctaps ? h->taps[2 * i] : h->taps[i]), h->ctaps ? (double)(float)h->taps[2 * i + 1] : 0.0);
            }
            const std::vector<float> T = make_hparts(hq, parts);
            std::copy(T.begin(), T.end(), rows.begin() + jr * tb);
        }
        h->ols_parts = parts;
        PCX_TRY(upload(h->HspecRows, rows));
        PCX_TRY(upload(h->tw4096, make_tw4096()));
        h->have_upols_rows = true;
    }
    h->have_poly = false;
    if (h->scalar == PCX_F32 && h->cplx && (h->L > 1 || h->M > 1) && h->K <= 2049 && h->L <= 64 && h->M < (1u << 17)) {
        // one spectrum per polyphase row: h_j[k] = taps[j + k*L] (FIRFilter.cpp:341-350)
        std::vector<float> rows(h->L * 2 * 4096);
        for (size_t j = 0; j < h->L; j++) {
            std::vector<std::complex<double>> hq;
            for (size_t k = 0; k < h->K; k++) {
                const size_t i = j + k * h->L;
                if (i >= h->ntaps) continue;
                hq.push_back(std::complex<double>((double)(float)(h->ctaps ? h->taps[2 * i] : h->taps[i]),
                                                  h->ctaps ? (double)(float)h->taps[2 * i + 1] : 0.0));
            }
            const std::vector<float> H = make_hspec4096(hq);
            std::copy(H.begin(), H.end(), rows.begin() + j * 2 * 4096);
        }
        PCX_TRY(upload(h->HspecRows, rows));
        PCX_TRY(upload(h->tw4096, make_tw4096()));
        h->have_poly = true;
    }
    h->have_decim = false;
    // folding pays from 4-fold on (and for M = 2 itself); 2-fold plus a cofactor measured slower than the full-rate kernel
    // (M = 10: 244 vs 281, M = 50: 251 vs 284 Gsamples/s in; M = 160 = 16 * 10: 373 vs 287)
    if (h->have_poly && h->L == 1 && (h->M == 2 || fir_decim_fold_factor(h->M) >= 4) && h->M / fir_decim_fold_factor(h->M) <= 65535 &&
        !PCX_ENV_SET("PCX_FIR_DECIM_FULLRATE")) {
        // decimating filter: one forward transform, the spectrum folded M-fold, a 4096/M-point inverse (fir_ols_decim.hip).
        // PCX_FIR_DECIM_FULLRATE (A/B) keeps the full-rate evaluation of the polyphase kernel.
        std::vector<std::complex<double>> hq(h->K);
        for (size_t k = 0; k < h->K; k++)
            hq[k] = std::complex<double>((double)(float)(h->ctaps ? h->taps[2 * k] : h->taps[k]), h->ctaps ? (double)(float)h->taps[2 * k + 1] : 0.0);
        // even M = M1 * M2: M1 = 16 / 8 / 4 / 2 folded into the spectrum, the cofactor kept one in M2 on the store
        PCX_TRY(upload(h->Hdecim, turn_spectrum_lanes(make_hspec(hq, 4096, fir_decim_fold_factor(h->M) - 1))));
        h->have_decim = true;
    }
    h->have_interp = false;
    if (h->have_poly && h->M == 1 && (h->L == 2 || h->L == 4 || h->L == 8 || h->L == 16) && !PCX_ENV_SET("PCX_FIR_DECIM_FULLRATE")) {
        // interpolating filter: a 4096/L-point forward transform, its spectrum replicated against H of the WHOLE tap vector,
        // the ordinary 4096-point inverse writing the interleaved output stream (fir_ols_decim.hip)
        const size_t A = 16 / h->L, kov_in = (h->K - 1 + A - 1) / A * A;
        if (kov_in <= 4096 / h->L / 2 && h->ntaps <= 2049) {
            std::vector<std::complex<double>> hq(h->ntaps);
            for (size_t k = 0; k < h->ntaps; k++)
                hq[k] = std::complex<double>((double)(float)(h->ctaps ? h->taps[2 * k] : h->taps[k]), h->ctaps ? (double)(float)h->taps[2 * k + 1] : 0.0);
            PCX_TRY(upload(h->Hdecim, turn_spectrum_lanes(make_hspec(hq, 4096))));
            h->have_interp = true;
        }
    }
    h->dirty = false;
    return PCX_OK;
}

// (internal, pcx_shard.hip) upload the handle's tables now -- every allocation and transfer of the control plane -- instead of at its next call
namespace pcx {
int fir_prepare(pcx_fir *h)
{
    DeviceScope dev_scope(h->cx.device);
    return fir_sync_tables(h);
}
void fir_set_slots(pcx_fir *h, unsigned slots) { h->slots = slots; }
}  // namespace pcx

int pcx_fir_create(int scalar, int is_complex, int complex_taps, pcx_fir **out)
{
    PCX_CHECK_ARG(out, "null out");
    // FIRFilterFactory's if-chain, FIRFilter.cpp:371-383
    PCX_CHECK_ARG(valid_scalar(scalar), "FIRFilterFactory: unsupported types (scalar %d)", scalar);
    PCX_CHECK_ARG(!(complex_taps && !is_complex), "FIRFilterFactory: unsupported types (COMPLEX taps on a real stream)");
    pcx_fir *h = new (std::nothrow) pcx_fir();
    if (!h) { set_error("out of memory"); return PCX_ERR_STATE; }
    h->scalar = scalar; h->cplx = is_complex ? 1 : 0; h->ctaps = complex_taps ? 1 : 0;
    h->taps.assign(h->ctaps ? 2 : 1, 0.0);
    h->taps[0] = 1.0;  // ctor: setTaps({1}), FIRFilter.cpp:125
    h->ntaps = 1;
    h->qf = process_qformat();
    fir_update_internals(h);
    { DeviceScope bind(h->cx.device); }   // the handle belongs to the device current on the creating thread
    *out = h;
    return PCX_OK;
}
int pcx_fir_destroy(pcx_fir *h) { delete h; return PCX_OK; }
int pcx_fir_set_taps(pcx_fir *h, const double *taps, size_t ntaps)
{
    PCX_CHECK_ARG(h, "null handle");
    PCX_CHECK_ARG(ntaps > 0 && taps, "FIRFilter::setTaps(): taps cannot be empty");
    h->taps.assign(taps, taps + ntaps * (h->ctaps ? 2 : 1));
    h->ntaps = ntaps;
    fir_update_internals(h);
    return PCX_OK;
}
int pcx_fir_set_decimation(pcx_fir *h, size_t decim)
{
    PCX_CHECK_ARG(h, "null handle");
    PCX_CHECK_ARG(decim != 0, "FIRFilter::setDecimation(): decimation cannot be 0");
    h->M = decim;
    fir_update_internals(h);
    return PCX_OK;
}
int pcx_fir_set_interpolation(pcx_fir *h, size_t interp)
{
    PCX_CHECK_ARG(h, "null handle");
    PCX_CHECK_ARG(interp != 0, "FIRFilter::setInterpolation(): interpolation cannot be 0");
    h->L = interp;
    fir_update_internals(h);
    return PCX_OK;
}
int pcx_fir_set_algo(pcx_fir *h, int algo)
{
    PCX_CHECK_ARG(h, "null handle");
    PCX_CHECK_ARG(algo >= PCX_FIR_AUTO && algo <= PCX_FIR_EXACT, "unknown FIR algorithm %d", algo);
    h->algo = algo;
    return PCX_OK;
}
int pcx_fir_set_qformat(pcx_fir *h, const pcx_qformat *q)
{
    PCX_CHECK_ARG(h, "null handle");
    QFormat f;
    PCX_TRY(qformat_from_api(q, &f));
    h->qf = f;
    h->dirty = true;      // the Q-format taps are quantised again before the next call
    return PCX_OK;
}
int pcx_fir_get_geometry(const pcx_fir *h, size_t *K, size_t *input_require)
{
    PCX_CHECK_ARG(h, "null handle");
    if (K) *K = h->K;
    if (input_require) *input_require = h->inputRequire;
    return PCX_OK;
}
int pcx_fir_last_algo(const pcx_fir *h) { return h ? h->last_algo : PCX_ERR_ARG; }
int pcx_fir_set_slots(pcx_fir *h, unsigned slots)
{
    PCX_CHECK_ARG(h, "null handle");
    PCX_CHECK_ARG(slots >= 128 && slots <= 1024 && slots % 128 == 0, "pcx_fir_set_slots: %u is not a multiple of 128 in 128..1024", slots);
    h->slots = slots;
    return PCX_OK;
}

static size_t fir_elem_bytes(const pcx_fir *h) { return (size_t)scalar_bytes(h->scalar) * (h->cplx ? 2 : 1); }

// N of FIRFilter.cpp:278
static size_t fir_iterations(const pcx_fir *h, size_t in_elems, size_t out_cap)
{
    if (in_elems < h->K - 1) return 0;
    const size_t a = (in_elems - (h->K - 1)) / h->M, b = out_cap / h->L;
    return std::min(a, b) * h->M;
}

static int fir_process_dev_impl(pcx_fir *h, const void *in_dev, size_t in_elems, void *out_dev, size_t out_cap,
                                size_t *consumed, size_t *produced, void *stream, const void *gate_word, unsigned gate_value, int *gated);

// Iterations a call may be cut at without changing any output: the block payload of the plain overlap-save plan (complex_float32,
// M = L = 1, 4096-sample blocks: block b of a call computes outputs [b S, (b + 1) S), S = 4096 - (K - 1 rounded up to 16)) where that
// plan serves the handle; the time-domain kernels and the exact integer pipelines compute every output by itself, any multiple of
// M will do (a generous one: chunks stay whole tiles).  Other float plans (long taps, resamplers) are cut at multiples of M * 4096:
// their outputs stay within the 1e-5 of the oracle either way, but are not bit-identical to an uncut call's.  (The partitioned
// long-tap plan computes block b from windows b, b - 1, ... alone -- whichever workgroup's run it falls into, so a call returns the
// same bits on any grid -- but the first blocks of a CUT call see zeros where the uncut call's windows hold samples that meet no
// tap: equal in exact arithmetic, not in the transform's rounding.)
static size_t fir_chunk_quantum(const pcx_fir *h)
{
    const bool plain = h->scalar == PCX_F32 && h->cplx && h->M == 1 && h->L == 1 && h->have_ols && h->ols_parts == 0 && h->K > 1;
    if (plain) return 4096 - (h->K - 1 + 15) / 16 * 16;
    return h->M * 4096;
}

int pcx_fir_process_dev(pcx_fir *h, const void *in_dev, size_t in_elems, void *out_dev, size_t out_cap,
                        size_t *consumed, size_t *produced, void *stream)
{
    PCX_TRACE();
    return fir_process_dev_impl(h, in_dev, in_elems, out_dev, out_cap, consumed, produced, stream, nullptr, 0, nullptr);
}
int pcx_fir_process_dev_gated(pcx_fir *h, const void *in_dev, size_t in_elems, void *out_dev, size_t out_cap,
                              size_t *consumed, size_t *produced, const void *gate_dev, unsigned gate_value, void *stream, int *gated)
{
    PCX_TRACE();
    PCX_CHECK_ARG(gate_dev && gated, "null gate");
    return fir_process_dev_impl(h, in_dev, in_elems, out_dev, out_cap, consumed, produced, stream, gate_dev, gate_value, gated);
}
int pcx_gate_signal_dev(void *gate_dev, unsigned value, void *stream)
{
    PCX_TRACE();
    PCX_CHECK_ARG(gate_dev, "null gate");
    // (diagnostic library only: the word written by the command processor instead of a one-thread kernel -- a kernel needs a slot, and beside
    // a launch that fills the device it gets one only when a workgroup of that launch exits; profiles/r04/gate_write_value.txt)
    if (PCX_ENV_SET("PCX_GATE_WRITE_VALUE")) {
        PCX_HIP(hipStreamWriteValue32(as_stream(stream), gate_dev, value, 0));
        return PCX_OK;
    }
    return launch_gate_signal(gate_dev, value, as_stream(stream));
}

static int fir_process_dev_impl(pcx_fir *h, const void *in_dev, size_t in_elems, void *out_dev, size_t out_cap,
                                size_t *consumed, size_t *produced, void *stream, const void *gate_word, unsigned gate_value, int *gated)
{
    PCX_CHECK_ARG(h && consumed && produced, "null argument");
    if (gated) *gated = 0;
    DeviceScope dev_scope(h->cx.device);
    *consumed = 0; *produced = 0;
    const size_t N = fir_iterations(h, in_elems, out_cap);
    if (N == 0) return PCX_OK;
    PCX_CHECK_ARG(in_dev && out_dev, "null buffer");
    PCX_TRY(fir_sync_tables(h));
    const size_t n_out = (N / h->M) * h->L;
    hipStream_t st = as_stream(stream);
    PCX_TRY(ctx_enter(h->cx, st));
    int algo = h->algo;
    const bool fast = fir_fast_applicable(h);
    if (algo == PCX_FIR_AUTO) {
        // measured sweep (tools/sweep_fir.py, 16 Mi samples): the frequency-domain kernel runs at
        // 285-325 Gsamples/s for every K <= 1023 (197 at K = 2049) while the time-domain tile
        // peaks at 240-256 and falls as 1/K beyond K ~ 48 -- so it is the choice whenever it applies
        // K == 1 (the block's default unit tap) stays on the time-domain tile: a pass-through
        // filter must return its input bit for bit, as the reference does
        if (fast && h->K == 1) algo = PCX_FIR_DIRECT;
        // decimating complex_float64 / complex_int16 / complex_int8 filters: the full-rate double pipeline with one output in M
        // stored runs at 130-170 Gsamples/s of input whatever K; the one-output-per-lane kernel it replaces measured 45-129
        // (int16) / 27-31 (float64) at 63 taps and 12-33 / 6-8 at 255 (tools/decim_int_probe.py)
        else if ((fast && h->have_ols) || h->have_poly || (h->have_real_ols && h->K > 1) || (h->have_upols_decim && h->K >= 16) || h->have_upols_rows ||
                 (h->have_ols64 && h->K >= (h->M > 1 ? 16 : kOls64MinTaps)) ||
                 (h->have_ols_int && h->K >= (h->M > 1 ? 32 : ols_int_min_taps(h->scalar))) ||
                 (h->have_ols_real64 && h->K >= (h->M > 1 ? 16 : ols_real64_min_taps(h->scalar))) ||
                 ((h->have_interp64 || h->have_interp_real || h->have_interp_f32) && h->K >= 16)) algo = PCX_FIR_OLS_FFT;
        // longer than every frequency-domain plan (K > 8193): the sliding-window kernel in the reference's own
        // operation order -- 8k-term float sums accumulate enough rounding that a reordered sum would sit on the 1e-5 bar
        else if (fast) algo = h->K > kOlsMaxTaps ? PCX_FIR_EXACT : PCX_FIR_DIRECT;
        else algo = is_float_scalar(h->scalar) ? PCX_FIR_DIRECT : PCX_FIR_EXACT;
    }
    if (algo == PCX_FIR_OLS_FFT && !((fast && h->have_ols) || h->have_poly || h->have_real_ols || h->have_upols_decim || h->have_upols_rows || h->have_ols64 || h->have_ols_int || h->have_ols_real64 ||
                                     h->have_interp64 || h->have_interp_real || h->have_interp_f32)) {
        set_error("fir: OLS_FFT needs complex_float32 or float32 and K<=8193 (interpolating: L<=64 rows) or complex_float64 / complex_int16 / complex_int8 with M=L=1, 2<=K<=4097");
        return PCX_ERR_UNSUPPORTED;
    }
    int rc;
    // only the samples the N iterations touch: N + K-1
    const size_t used_in = N + h->K - 1;
    const QShift qs = q_shift(h->qf, h->scalar);      // integer element types: fromQ<OutType> of FIRFilter.cpp:300 under the handle's reading
    if (gate_word) {
        // a gated call: only the plain complex_float32 M = L = 1 plan on 4096-sample blocks has the gate (and only its dealt launch,
        // launch_fir_cf32_ols4096 decides).  Anything else: *gated stays 0, nothing has been queued, the caller orders the halo itself.
        const bool plain = algo == PCX_FIR_OLS_FFT && !h->have_interp_real && !h->have_interp64 && !h->have_ols_real64 && !h->have_ols64 &&
                           !h->have_ols_int && !h->have_real_ols && !h->have_interp && !h->have_decim && !h->have_poly && h->ols_parts == 0;
        if (!plain) return PCX_OK;
        rc = launch_fir_cf32_ols4096(in_dev, used_in, out_dev, n_out, h->Hspec.p, h->K, h->tw4096.p, h->sched.p, st, gate_word, gate_value, gated, h->slots);
        if (rc != PCX_OK || !*gated) return rc;
        h->last_algo = algo;
        *consumed = N;
        *produced = n_out;
        return PCX_OK;
    }
    // interpolation through polyphase ROWS: every row filtered at the input rate into a contiguous workspace row, then one
    // interleaving pass.  Long calls go in batches of iterations so that the workspace stays at kRowsWorkspaceCap bytes
    // whatever the call (it would be a second copy of the output otherwise).
    auto rows_path = [&](size_t eb, size_t Mdec, auto &&row) -> int {
        size_t nb_max = kRowsWorkspaceCap / (h->L * eb) / Mdec * Mdec;     // whole output samples per batch
        if (nb_max < Mdec) nb_max = Mdec;
        PCX_TRY(h->wsRows.ensure((N < nb_max ? N : nb_max) * h->L * eb));
        for (size_t i0 = 0; i0 < N; i0 += nb_max) {
            const size_t nb = N - i0 < nb_max ? N - i0 : nb_max;
            const char *in_b = static_cast<const char *>(in_dev) + i0 * eb;   // the rows run at M = 1: one input sample per iteration
            for (size_t jr = 0; jr < h->L; jr++) PCX_TRY(row(in_b, nb, static_cast<char *>(h->wsRows.p) + jr * nb * eb, jr));
            PCX_TRY(launch_interleave_rows(h->wsRows.p, static_cast<char *>(out_dev) + i0 * h->L / Mdec * eb, nb, h->L, eb, Mdec, st));
        }
        return PCX_OK;
    };
    if (algo == PCX_FIR_OLS_FFT && h->have_interp_f32) {
        rc = rows_path(4, h->M, [&](const void *in_b, size_t nb, void *dst, size_t jr) {
            return launch_fir_f32_ols4096(in_b, nb + h->K - 1, dst, nb, static_cast<const char *>(h->HspecRows.p) + jr * 2 * 4096 * sizeof(float), h->K,
                                          h->tw4096.p, h->sched.p, st);
        });
    } else if (algo == PCX_FIR_OLS_FFT && h->have_interp_real) {
        rc = rows_path(fir_elem_bytes(h), h->M, [&](const void *in_b, size_t nb, void *dst, size_t jr) {
            return launch_fir_real_ols(in_b, nb + h->K - 1, dst, nb, static_cast<const char *>(h->HrowsD.p) + jr * 2 * 4096 * sizeof(double), h->K, 12,
                                       h->tw4096.p, h->scalar == PCX_F64 ? 0 : h->scalar == PCX_I16 ? 1 : h->scalar == PCX_I8 ? 2 : 3, 1, qs, st);
        });
    } else if (algo == PCX_FIR_OLS_FFT && h->have_interp64) {
        rc = rows_path(fir_elem_bytes(h), h->M, [&](const void *in_b, size_t nb, void *dst, size_t jr) {
            return launch_fir_cf64_ols(in_b, nb + h->K - 1, dst, nb, static_cast<const char *>(h->HrowsD.p) + jr * 2 * 4096 * sizeof(double), h->K, 12,
                                       h->tw4096.p, h->scalar == PCX_F64 ? 0 : h->scalar == PCX_I16 ? 1 : 2, 1, qs, st);
        });
    } else if (algo == PCX_FIR_OLS_FFT && h->have_ols_real64) {
        rc = launch_fir_real_ols(in_dev, used_in, out_dev, N, h->Hspec.p, h->K, h->ols_log2n, h->tw4096.p,
                                 h->scalar == PCX_F64 ? 0 : h->scalar == PCX_I16 ? 1 : h->scalar == PCX_I8 ? 2 : 3, h->M, qs, st, h->sched.p);
    } else if (algo == PCX_FIR_OLS_FFT && (h->have_ols64 || h->have_ols_int)) {
        rc = launch_fir_cf64_ols(in_dev, used_in, out_dev, N, h->Hspec.p, h->K, h->ols_log2n, h->tw4096.p,
                                 h->scalar == PCX_F64 ? 0 : h->scalar == PCX_I16 ? 1 : 2, h->M, qs, st, h->sched.p);
    } else if (algo == PCX_FIR_OLS_FFT && h->have_upols_rows) {
        const size_t tb = fir_upols_table_bytes(h->ols_parts);
        rc = rows_path(fir_elem_bytes(h), h->M, [&](const void *in_b, size_t nb, void *dst, size_t jr) {
            return launch_fir_cf32_upols(in_b, nb + h->K - 1, dst, nb, static_cast<const char *>(h->HspecRows.p) + jr * tb, h->K, h->ols_parts, h->tw4096.p, st,
                                         !h->cplx, 1);
        });
    } else if (algo == PCX_FIR_OLS_FFT && h->have_upols_decim) {
        rc = launch_fir_cf32_upols(in_dev, used_in, out_dev, N, h->Hspec.p, h->K, h->ols_parts, h->tw4096.p, st, !h->cplx, h->M);
    } else if (algo == PCX_FIR_OLS_FFT && h->have_real_ols && h->ols_parts != 0) {
        rc = launch_fir_cf32_upols(in_dev, used_in, out_dev, n_out, h->Hspec.p, h->K, h->ols_parts, h->tw4096.p, st, true);
    } else if (algo == PCX_FIR_OLS_FFT && h->have_real_ols) {
        rc = launch_fir_f32_ols4096(in_dev, used_in, out_dev, n_out, h->Hspec.p, h->K, h->tw4096.p, h->sched.p, st);
    } else if (algo == PCX_FIR_OLS_FFT && h->have_interp) {
        rc = launch_fir_cf32_ols4096_interp(in_dev, used_in, out_dev, N, h->Hdecim.p, h->K, h->L, h->tw4096.p, h->sched.p, st);
    } else if (algo == PCX_FIR_OLS_FFT && h->have_decim) {
        rc = launch_fir_cf32_ols4096_decim(in_dev, used_in, out_dev, N, h->Hdecim.p, h->K, h->M, h->tw4096.p, h->sched.p, st);
    } else if (algo == PCX_FIR_OLS_FFT && h->have_poly && h->M == 1 && h->K <= 2049 && !PCX_ENV_SET("PCX_FIR_POLY_STRIDED")) {
        // interpolation by other factors: each polyphase row through the undecimated kernel into a contiguous workspace row,
        // then one interleaving pass (PCX_FIR_POLY_STRIDED (A/B) keeps the polyphase kernel's stride-L stores)
        rc = rows_path(8, 1, [&](const void *in_b, size_t nb, void *dst, size_t jr) {
            return launch_fir_cf32_ols4096(in_b, nb + h->K - 1, dst, nb, static_cast<const char *>(h->HspecRows.p) + jr * 2 * 4096 * sizeof(float), h->K,
                                           h->tw4096.p, h->sched.p, st);
        });
    } else if (algo == PCX_FIR_OLS_FFT && h->have_poly) {
        rc = launch_fir_cf32_ols4096_poly(in_dev, used_in, out_dev, N, h->HspecRows.p, h->K, h->L, h->M, h->tw4096.p, st);
    } else if (algo == PCX_FIR_OLS_FFT && h->ols_parts != 0) {
        rc = launch_fir_cf32_upols(in_dev, used_in, out_dev, n_out, h->Hspec.p, h->K, h->ols_parts, h->tw4096.p, st);
    } else if (algo == PCX_FIR_OLS_FFT) {
        rc = launch_fir_cf32_ols4096(in_dev, used_in, out_dev, n_out, h->Hspec.p, h->K, h->tw4096.p, h->sched.p, st, nullptr, 0, nullptr, h->slots, h->lead_valid);
    } else if (algo == PCX_FIR_DIRECT && fast && (2048 + h->Kp + 8) * 9 / 8 * 8 + 64 <= 64 * 1024) {
        // the LDS-tiled time-domain kernel while its tile (2048 outputs + taps) fits; longer filters than every
        // fast plan (K > 8193) take the sliding-window kernel below
        rc = launch_fir_cf32_direct(in_dev, used_in, out_dev, n_out, h->tapsRev.p, h->K, h->Kp, st);
    } else {
        FirGeom g{h->L, h->M, h->K, static_cast<const uint32_t *>(h->rowLen.p), h->rowTaps.p};
        // PCX_FIR_SLIDE=0 keeps the one-output-per-lane kernel for M = L = 1 too (A/B)
        const int slide = (int)PCX_ENV_INT("PCX_FIR_SLIDE", 1);
        // PCX_FIR_DOT2=0 keeps complex_int16 on the 24-bit multiply path (A/B)
        const int dot2 = (int)PCX_ENV_INT("PCX_FIR_DOT2", 1);
        if (slide && dot2 && h->taps16 && h->L == 1 && h->M == 1 && h->K <= 12000)
            rc = launch_fir_ci16_dot2(in_dev, out_dev, n_out, h->K, h->tapsP.p, h->scalar == PCX_I8, qs, st);
        else if (slide && h->L == 1 && h->M == 1)
            rc = launch_fir_slide(h->scalar, h->cplx, h->ctaps, algo == PCX_FIR_EXACT, h->taps24, g, in_dev, out_dev, n_out, qs, st);
        else
            rc = launch_fir_generic(h->scalar, h->cplx, h->ctaps, algo == PCX_FIR_EXACT, g, in_dev, out_dev, n_out, qs, st);
    }
    if (rc != PCX_OK) return rc;
    h->last_algo = algo;
    *consumed = N;
    *produced = n_out;
    return PCX_OK;
}

int pcx_fir_process(pcx_fir *h, const void *in, size_t in_elems, void *out, size_t out_cap, size_t *consumed, size_t *produced)
{
    PCX_TRACE();
    PCX_CHECK_ARG(h && consumed && produced, "null argument");
    DeviceScope dev_scope(h->cx.device);
    *consumed = 0; *produced = 0;
    const size_t N = fir_iterations(h, in_elems, out_cap);
    if (N == 0) return PCX_OK;
    PCX_CHECK_ARG(in && out, "null buffer");
    const size_t esz = fir_elem_bytes(h), used_in = N + h->K - 1, n_out = (N / h->M) * h->L;
    // page-locked buffers (a pinned BufferManager's slabs): the kernels run on them in place; pageable ones are staged.
    // Everything goes through the handle's own stream.
    hipStream_t st;
    PCX_TRY(ctx_own_stream(h->cx, &st));
    PCX_TRY(fir_sync_tables(h));        // (tables first: nothing of the control plane between the transfers queued below)
    const void *din; void *dout; bool staged;
    if (n_out * esz >= drain_from() && host_page_locked(out)) {
        // the drained output (above): chunk by chunk into a device workspace, the copy engine behind.  A chunk is a whole number of
        // the plan's blocks where the plan has blocks (so that every output is computed exactly as by one call over everything --
        // the overlap-save kernels round differently at other block boundaries; lead_valid makes a chunk's first block a full
        // one), and of M iterations always
        const int nch = drain_chunks(n_out * esz);
        const size_t q = fir_chunk_quantum(h);
        size_t Nc = ((N + nch - 1) / nch + q - 1) / q * q;
        PCX_TRY(h->wsOut.dev.ensure(n_out * esz));
        PCX_TRY(drain_setup(h->cx, nch));
        if (!device_alias(in)) { PCX_TRY(h->wsIn.dev.ensure(used_in * esz)); PCX_TRY(h->wsIn.pin.ensure(used_in * esz)); }
        PCX_TRY(stage_in(in, used_in * esz, h->wsIn, st, &din));
        size_t done = 0;
        for (int c = 0; c < nch && done < N; c++) {
            const size_t n = N - done < Nc ? N - done : Nc, o0 = done / h->M * h->L, no = n / h->M * h->L;
            size_t cc = 0, pp = 0;
            h->lead_valid = done;
            const int rc = pcx_fir_process_dev(h, static_cast<const char *>(din) + done * esz, n + h->K - 1, static_cast<char *>(h->wsOut.dev.p) + o0 * esz, no,
                                               &cc, &pp, st);
            h->lead_valid = 0;
            PCX_TRY(rc);
            if (cc != n || pp != no) { set_error("fir: a chunk of the drained call came back short (%zu of %zu iterations)", cc, n); return PCX_ERR_STATE; }
            PCX_TRY(drain_chunk(h->cx, c, st, static_cast<char *>(out) + o0 * esz, static_cast<const char *>(h->wsOut.dev.p) + o0 * esz, no * esz));
            done += n;
        }
        PCX_TRY(drain_finish(h->cx, st));
        *consumed = N;
        *produced = n_out;
        return PCX_OK;
    }
    PCX_TRY(stage_reserve(out, n_out * esz, h->wsOut));
    PCX_TRY(stage_in(in, used_in * esz, h->wsIn, st, &din));
    PCX_TRY(stage_out_begin(out, n_out * esz, h->wsOut, &dout, &staged));
    // the kernel reads or writes the caller's page-locked memory in place: the launch shape of a link-bound call (host_grid above)
    const unsigned keep_slots = h->slots;
    int rc;
    {
        LinkBound shape(in, out);                      // (every static plan: persistent_grid / stream_grid look at it)
        if (g_link_grid) h->slots = g_link_grid;       // (the dealt plain plan: slots < 128 = that many workgroups, no dealer)
        rc = pcx_fir_process_dev(h, din, used_in, dout, n_out, consumed, produced, st);
    }
    h->slots = keep_slots;
    PCX_TRY(rc);
    return stage_out_end(out, *produced * esz, h->wsOut, staged, st);
}

/* ===================================================================== *
 *  fused Rotate -> FIR -> FreqDemod
 * ===================================================================== */
struct pcx_fmchain {
    ExecCtx cx;
    double phase = 0.0;
    bool phase_set = false;  // Rotate before setPhase: zero phasor (Rotate.cpp:60-62)
    std::vector<double> taps;
    size_t ntaps = 1;
    int ctaps = 0;
    bool dirty = true;
    size_t K = 1, Kp = 8;
    DevBuf tapsRev, Hspec, tw4096, prev;
    StageBuf wsIn, wsOut;
    DevBuf sched;   // dynamic block assignment of the fused kernel (pcx_sched.hpp), zeroed at create
    unsigned slots = 1024;
    int cur = 0;
    int algo = PCX_FIR_AUTO, last_algo = 0;
    bool have_ols = false;
    // filters longer than the fused kernels' plans (K > 2048): the FIR stage as its own launch (any K),
    // FreqDemod behind it on the same carried state
    pcx_fir *long_fir = nullptr;
    DevBuf long_y;
    ~pcx_fmchain() { delete long_fir; }
};
int pcx_fmchain_create(pcx_fmchain **out)
{
    PCX_CHECK_ARG(out, "null out");
    pcx_fmchain *h = new (std::nothrow) pcx_fmchain();
    if (!h) { set_error("out of memory"); return PCX_ERR_STATE; }
    h->taps.assign(1, 1.0);
    DeviceScope dev_scope(h->cx.device);
    int rc = h->prev.ensure_zeroed(64);
    if (rc == PCX_OK) rc = h->sched.ensure_zeroed(kSchedBytes);
    if (rc != PCX_OK) { delete h; return rc; }
    *out = h;
    return PCX_OK;
}
int pcx_fmchain_destroy(pcx_fmchain *h) { delete h; return PCX_OK; }
int pcx_fmchain_set_phase(pcx_fmchain *h, double phase)
{
    PCX_CHECK_ARG(h, "null handle");
    h->phase = phase; h->phase_set = true; h->dirty = true;
    return PCX_OK;
}
int pcx_fmchain_set_taps(pcx_fmchain *h, const double *taps, size_t ntaps, int complex_taps)
{
    PCX_CHECK_ARG(h, "null handle");
    PCX_CHECK_ARG(ntaps > 0 && taps, "FIRFilter::setTaps(): taps cannot be empty");
    h->taps.assign(taps, taps + ntaps * (complex_taps ? 2 : 1));
    h->ntaps = ntaps; h->ctaps = complex_taps ? 1 : 0; h->dirty = true;
    return PCX_OK;
}
int pcx_fmchain_reset(pcx_fmchain *h)
{
    PCX_CHECK_ARG(h, "null handle");
    DeviceScope dev_scope(h->cx.device);
    // as pcx_freqdemod_reset: ordered behind the previous call and ahead of the next one
    hipStream_t st = h->cx.have_last ? h->cx.last : nullptr;
    if (!h->cx.have_last) PCX_TRY(ctx_own_stream(h->cx, &st));
    PCX_TRY(ctx_enter(h->cx, st));
    PCX_TRY(launch_zero_words(h->prev.p, 16, st));   // (a kernel, not hipMemsetAsync: see launch_zero_words)
    h->cur = 0;
    return PCX_OK;
}
static int fmchain_sync(pcx_fmchain *h)
{
    if (!h->dirty) return PCX_OK;
    PCX_TRY(ctx_quiesce(h->cx));   // an earlier call's kernel may still be reading the tables rewritten below
    const size_t K = h->ntaps;
    h->K = K;
    h->Kp = (K + 7) / 8 * 8;
    // Rotate's phasor folded into the taps: FIR(p*x) = (p*h) (*) x.  p is first narrowed
    // to float as floatToQ<complex<float>> does (Rotate.cpp:74), h as FIRFilter.cpp:348.
    const std::complex<double> pd = std::polar(1.0, h->phase);   // the expression of Rotate::setPhase (Rotate.cpp:74)
    const std::complex<double> p = h->phase_set ? std::complex<double>((double)(float)pd.real(), (double)(float)pd.imag())
                                                : std::complex<double>(0.0, 0.0);
    std::vector<float> rev(2 * h->Kp, 0.f);
    for (size_t m = 0; m < K; m++) {
        const size_t k = K - 1 - m;
        const std::complex<double> t = h->ctaps ? std::complex<double>((double)(float)h->taps[2 * k], (double)(float)h->taps[2 * k + 1])
                                                : std::complex<double>((double)(float)h->taps[k], 0.0);
        const std::complex<double> g = p * t;
        rev[2 * m] = (float)g.real();
        rev[2 * m + 1] = (float)g.imag();
    }
    PCX_TRY(upload(h->tapsRev, rev));
    h->have_ols = false;
    if (K <= 2048) {   // frequency-domain variant: H' = FFT(p * h) / 4096
        std::vector<std::complex<double>> g(K);
        for (size_t m = 0; m < K; m++) g[K - 1 - m] = std::complex<double>((double)rev[2 * m], (double)rev[2 * m + 1]);
        PCX_TRY(upload(h->Hspec, make_hspec4096(g)));
        PCX_TRY(upload(h->tw4096, make_tw4096()));
        h->have_ols = true;
    } else {
        // unfused long-filter path: complex taps g = p * h through the FIR handle (frequency-domain plans to
        // 8193 taps, the reference-order kernel beyond)
        if (!h->long_fir) PCX_TRY(pcx_fir_create(PCX_F32, 1, 1, &h->long_fir));
        std::vector<double> g(2 * K);
        for (size_t m = 0; m < K; m++) { g[2 * (K - 1 - m)] = (double)rev[2 * m]; g[2 * (K - 1 - m) + 1] = (double)rev[2 * m + 1]; }
        PCX_TRY(pcx_fir_set_taps(h->long_fir, g.data(), K));
    }
    h->dirty = false;
    return PCX_OK;
}
// (internal, pcx_shard.hip) upload the chain's tables now instead of at its next call
namespace pcx {
int fmchain_prepare(pcx_fmchain *h)
{
    DeviceScope dev_scope(h->cx.device);
    return fmchain_sync(h);
}
void fmchain_set_slots(pcx_fmchain *h, unsigned slots) { h->slots = slots; }
}  // namespace pcx
int pcx_fmchain_set_algo(pcx_fmchain *h, int algo)
{
    PCX_CHECK_ARG(h, "null handle");
    PCX_CHECK_ARG(algo == PCX_FIR_AUTO || algo == PCX_FIR_DIRECT || algo == PCX_FIR_OLS_FFT, "fm chain: algorithm %d not available", algo);
    h->algo = algo;
    return PCX_OK;
}
int pcx_fmchain_last_algo(const pcx_fmchain *h) { return h ? h->last_algo : PCX_ERR_ARG; }
int pcx_fmchain_set_slots(pcx_fmchain *h, unsigned slots)
{
    PCX_CHECK_ARG(h, "null handle");
    PCX_CHECK_ARG(slots >= 128 && slots <= 1024 && slots % 128 == 0, "pcx_fmchain_set_slots: %u is not a multiple of 128 in 128..1024", slots);
    h->slots = slots;
    return PCX_OK;
}
static int fmchain_process_dev_impl(pcx_fmchain *h, const void *in_dev, size_t in_elems, void *out_dev, size_t out_cap,
                                    size_t *consumed, size_t *produced, void *stream, const void *gate_word, unsigned gate_value, int *gated);
int pcx_fmchain_process_dev(pcx_fmchain *h, const void *in_dev, size_t in_elems, void *out_dev, size_t out_cap,
                            size_t *consumed, size_t *produced, void *stream)
{
    PCX_TRACE();
    return fmchain_process_dev_impl(h, in_dev, in_elems, out_dev, out_cap, consumed, produced, stream, nullptr, 0, nullptr);
}
int pcx_fmchain_process_dev_gated(pcx_fmchain *h, const void *in_dev, size_t in_elems, void *out_dev, size_t out_cap,
                                  size_t *consumed, size_t *produced, const void *gate_dev, unsigned gate_value, void *stream, int *gated)
{
    PCX_TRACE();
    PCX_CHECK_ARG(gate_dev && gated, "null gate");
    return fmchain_process_dev_impl(h, in_dev, in_elems, out_dev, out_cap, consumed, produced, stream, gate_dev, gate_value, gated);
}
static int fmchain_process_dev_impl(pcx_fmchain *h, const void *in_dev, size_t in_elems, void *out_dev, size_t out_cap,
                                    size_t *consumed, size_t *produced, void *stream, const void *gate_word, unsigned gate_value, int *gated)
{
    PCX_CHECK_ARG(h && consumed && produced, "null argument");
    if (gated) *gated = 0;
    DeviceScope dev_scope(h->cx.device);
    *consumed = 0; *produced = 0;
    PCX_TRY(fmchain_sync(h));
    if (in_elems < h->K) return PCX_OK;
    const size_t N = std::min(in_elems - (h->K - 1), out_cap);
    if (N == 0) return PCX_OK;
    PCX_CHECK_ARG(in_dev && out_dev, "null buffer");
    char *base = static_cast<char *>(h->prev.p);
    PCX_TRY(ctx_enter(h->cx, as_stream(stream)));
    int algo = h->algo;
    if (gate_word && !(h->have_ols && (algo == PCX_FIR_AUTO || algo == PCX_FIR_OLS_FFT))) return PCX_OK;   // no gate but in the fused frequency-domain kernel
    if (algo == PCX_FIR_AUTO && !h->have_ols) {
        // K > 2048: two launches (FIR with the folded phasor, then FreqDemod) sharing the chain's carried state
        // (in batches: the intermediate FIR output stays at kRowsWorkspaceCap bytes whatever the call; the demodulator's state
        // walks through the batches exactly as it does through work() calls)
        const size_t nb_max = kRowsWorkspaceCap / 8;
        PCX_TRY(h->long_y.ensure((N < nb_max ? N : nb_max) * 8));
        for (size_t i0 = 0; i0 < N; i0 += nb_max) {
            const size_t nb = N - i0 < nb_max ? N - i0 : nb_max;
            size_t c2 = 0, p2 = 0;
            PCX_TRY(pcx_fir_process_dev(h->long_fir, static_cast<const float2 *>(in_dev) + i0, nb + h->K - 1, h->long_y.p, nb, &c2, &p2, stream));
            if (c2 != nb || p2 != nb) { set_error("fm chain: FIR stage produced %zu of %zu", p2, nb); return PCX_ERR_STATE; }
            PCX_TRY(launch_freqdemod(PCX_F32, h->long_y.p, static_cast<float *>(out_dev) + i0, nb, base + 32 * h->cur, base + 32 * (h->cur ^ 1),
                                     as_stream(stream)));
            h->cur ^= 1;
        }
        h->last_algo = PCX_FIR_AUTO;
        *consumed = N; *produced = N;
        return PCX_OK;
    }
    if (algo == PCX_FIR_AUTO) algo = PCX_FIR_OLS_FFT;
    if (algo == PCX_FIR_OLS_FFT) {
        if (!h->have_ols) { set_error("fm chain: OLS_FFT needs K <= 2048"); return PCX_ERR_UNSUPPORTED; }
        PCX_TRY(launch_fmchain_cf32_ols4096(in_dev, N + h->K - 1, out_dev, N, h->Hspec.p, h->K, h->tw4096.p, base + 32 * h->cur,
                                            base + 32 * (h->cur ^ 1), h->sched.p, as_stream(stream), gate_word, gate_value, gated, h->slots));
        if (gate_word && !*gated) return PCX_OK;      // a short call: the grid-stride kernel has no gate, nothing was queued
    } else {
        PCX_TRY(launch_fmchain_cf32(in_dev, N + h->K - 1, out_dev, N, h->tapsRev.p, h->K, h->Kp, base + 32 * h->cur,
                                    base + 32 * (h->cur ^ 1), as_stream(stream)));
    }
    h->last_algo = algo;
    h->cur ^= 1;
    *consumed = N; *produced = N;
    return PCX_OK;
}
int pcx_fmchain_process(pcx_fmchain *h, const void *in, size_t in_elems, void *out, size_t out_cap, size_t *consumed, size_t *produced)
{
    PCX_TRACE();
    PCX_CHECK_ARG(h && consumed && produced, "null argument");
    DeviceScope dev_scope(h->cx.device);
    *consumed = 0; *produced = 0;
    PCX_TRY(fmchain_sync(h));
    if (in_elems < h->K) return PCX_OK;
    const size_t N = std::min(in_elems - (h->K - 1), out_cap);
    if (N == 0) return PCX_OK;
    PCX_CHECK_ARG(in && out, "null buffer");
    const size_t used = N + h->K - 1;
    hipStream_t st;
    PCX_TRY(ctx_own_stream(h->cx, &st));
    const void *din; void *dout; bool staged;
    PCX_TRY(stage_reserve(out, N * 4, h->wsOut));
    PCX_TRY(stage_in(in, used * 8, h->wsIn, st, &din));
    PCX_TRY(stage_out_begin(out, N * 4, h->wsOut, &dout, &staged));
    const unsigned keep_slots = h->slots;
    int rc;
    {
        LinkBound shape(in, out);                      // (pcx_fir_process: a link-bound call's launch shape)
        if (g_link_grid) h->slots = g_link_grid;
        rc = pcx_fmchain_process_dev(h, din, used, dout, N, consumed, produced, st);
    }
    h->slots = keep_slots;
    PCX_TRY(rc);
    return stage_out_end(out, N * 4, h->wsOut, staged, st);
}
