// pcx_tables.hpp -- the coefficient tables of the transform kernels, built on the host in double precision and rounded once
// (used by the FIR / fused-chain handles, pcx_fir_api.hip, and the FFT handle, pcx_fft_api.hip).  Not installed.
#pragma once
#include <algorithm>
#include <complex>
#include <vector>

#include "pcx_internal.hpp"

namespace pcx {

// lane-constant twiddle table of the radix-16 x3 4096-point transform (fft4096.hpp):
//   p = 0..14:  p < 3 -> (w^4)^(p+1);   p = 3 + (n2-1)*4 + k1 -> w^n2 * W16^(n2 k1)
//   tab[p * 16 + kk]        with w = exp(-j 2 pi kk / 256)    (pass 2, 240 entries)
//   tab[240 + p * 256 + j]  with w = exp(-j 2 pi j / 4096)    (pass 3, 3840 entries)
inline std::vector<float> make_tw4096()
{
    std::vector<float> t(2 * (15 * 16 + 15 * 256));
    const double two_pi = 6.283185307179586476925286766559;
    auto angle = [&](int p, double base /* turns per unit of w */) {
        if (p < 3) return base * 4.0 * (p + 1);
        const int n2 = (p - 3) / 4 + 1, k1 = (p - 3) % 4;
        return base * n2 + (double)(n2 * k1) / 16.0;
    };
    for (int p = 0; p < 15; p++) {
        for (int kk = 0; kk < 16; kk++) {
            const double a = -two_pi * angle(p, (double)kk / 256.0);
            t[2 * (p * 16 + kk)] = (float)std::cos(a);
            t[2 * (p * 16 + kk) + 1] = (float)std::sin(a);
        }
        for (int j = 0; j < 256; j++) {
            const double a = -two_pi * angle(p, (double)j / 4096.0);
            t[2 * (240 + p * 256 + j)] = (float)std::cos(a);
            t[2 * (240 + p * 256 + j) + 1] = (float)std::sin(a);
        }
    }
    return t;
}

// lane-constant table of the radix-16 family (fft_r16.hip): [15][16] pass Ns=16, [15][256] pass
// Ns=256 (numBins >= 4096), then the final radix-R pass: entry (t*(R-1) + r-1, l) = W_N^((l + t*LPF) r)
template <typename T = float>
inline std::vector<T> make_tw_r16(int log2n)
{
    const int N = 1 << log2n, LPF = N / 16, A = log2n / 4, R = 1 << (log2n % 4);
    std::vector<T> t(2 * fft_r16_table_elems(log2n));
    const double two_pi = 6.283185307179586476925286766559;
    auto put = [&](size_t idx, double turns) {
        t[2 * idx] = (T)std::cos(-two_pi * turns);
        t[2 * idx + 1] = (T)std::sin(-two_pi * turns);
    };
    auto angle15 = [](int p, double base) {
        if (p < 3) return base * 4.0 * (p + 1);
        const int n2 = (p - 3) / 4 + 1, k1 = (p - 3) % 4;
        return base * n2 + (double)(n2 * k1) / 16.0;
    };
    size_t off = 0;
    for (int p = 0; p < 15; p++)
        for (int kk = 0; kk < 16; kk++) put(off + p * 16 + kk, angle15(p, (double)kk / 256.0));
    off += 15 * 16;
    if (A >= 3) {
        for (int p = 0; p < 15; p++)
            for (int j = 0; j < 256; j++) put(off + p * 256 + j, angle15(p, (double)j / 4096.0));
        off += 15 * 256;
    }
    if (R > 1) {
        const int G = 16 / R;
        for (int tt = 0; tt < G; tt++)
            for (int r = 1; r < R; r++)
                for (int l = 0; l < LPF; l++)
                    put(off + (size_t)(tt * (R - 1) + (r - 1)) * LPF + l, (double)(((long long)(l + tt * LPF) * r) % N) / (double)N);
    }
    return t;
}

// tables of the double-precision overlap-save kernels (fir_ols_f64.hip).  log2n == 12: the in-place transform pair of
// fft_f64.hpp (ip4096) -- [15][16] W256^((p + 1) c), then [15][256] W4096^((p + 1) idx); otherwise the radix-16 family's.
inline std::vector<double> make_tw_ols64(int log2n)
{
    if (log2n != 12) return make_tw_r16<double>(log2n);
    std::vector<double> t(2 * (15 * 16 + 15 * 256));
    const double two_pi = 6.283185307179586476925286766559;
    auto put = [&](size_t idx, long long num, long long den) {
        const double turns = (double)(num % den) / (double)den;
        t[2 * idx] = std::cos(-two_pi * turns);
        t[2 * idx + 1] = std::sin(-two_pi * turns);
    };
    for (int p = 0; p < 15; p++)
        for (int c = 0; c < 16; c++) put((size_t)p * 16 + c, (long long)(p + 1) * c, 256);
    for (int p = 0; p < 15; p++)
        for (int i = 0; i < 256; i++) put((size_t)240 + (size_t)p * 256 + i, (long long)(p + 1) * i, 4096);
    return t;
}

// H[b] = sum_k h[k] exp(-j 2 pi b k / 4096) / 4096 (the 1/N of the inverse transform folded
// in), accumulated in double, rounded once to float; natural bin order
// `advance`: circular advance of the filter output by that many samples (H[b] *= exp(+j 2 pi b advance / N)) -- the
// decimator's phase for the folded-spectrum kernel (fir_ols_decim.hip)
template <typename T = float>
inline std::vector<T> make_hspec(const std::vector<std::complex<double>> &h, size_t N, size_t advance = 0)
{
    std::vector<double> cs(2 * N);
    const double two_pi = 6.283185307179586476925286766559;
    for (size_t i = 0; i < N; i++) { cs[2 * i] = std::cos(two_pi * (double)i / (double)N); cs[2 * i + 1] = -std::sin(two_pi * (double)i / (double)N); }
    std::vector<T> H(2 * N);
    for (size_t b = 0; b < N; b++) {
        double sr = 0, si = 0;
        for (size_t k = 0; k < h.size(); k++) {
            const size_t e = (b * k) & (N - 1);
            sr += h[k].real() * cs[2 * e] - h[k].imag() * cs[2 * e + 1];
            si += h[k].real() * cs[2 * e + 1] + h[k].imag() * cs[2 * e];
        }
        if (advance) {
            const size_t e = (N - (b * advance) % N) % N;     // cs[e] = exp(-j 2 pi e / N) = exp(+j 2 pi b advance / N)
            const double pr = sr * cs[2 * e] - si * cs[2 * e + 1], pi = sr * cs[2 * e + 1] + si * cs[2 * e];
            sr = pr; si = pi;
        }
        H[2 * b] = (T)(sr / (double)N);
        H[2 * b + 1] = (T)(si / (double)N);
    }
    return H;
}
inline std::vector<float> make_hspec4096(const std::vector<std::complex<double>> &h) { return make_hspec(h, 4096); }
// The resampling kernels (fir_ols_decim.hip) re-read H from L2 in every block, so their copy is stored the way their lanes hold the
// spectrum (fft4096.hpp, spec_lane): entry j + 256 r is bin (j >> 4) + 16 (j & 15) + 256 r, and a wave still reads whole 512-byte rows
inline std::vector<float> turn_spectrum_lanes(const std::vector<float> &H)
{
    std::vector<float> T(H.size());
    for (size_t r = 0; r < 16; r++)
        for (size_t j = 0; j < 256; j++) {
            const size_t src = ((j >> 4) + 16 * (j & 15)) + 256 * r, dst = j + 256 * r;
            T[2 * dst] = H[2 * src];
            T[2 * dst + 1] = H[2 * src + 1];
        }
    return T;
}
// Table of the partitioned overlap-save kernel (fir_ols_part.hip): the taps cut into partitions of 2048 (the last one takes
// what is left, up to 2049), each partition's 4096-bin spectrum in the lanes' order, two partitions to a 16-byte entry
// (plane g: [16][256] entries {H_2g, H_2g+1}; the last plane of an odd count holds one partition in 8-byte entries).
inline std::vector<float> make_hparts(const std::vector<std::complex<double>> &h, int parts)
{
    const size_t B = 2048;
    std::vector<std::vector<float>> T((size_t)parts);
    for (int p = 0; p < parts; p++) {
        const size_t lo = (size_t)p * B, hi = p + 1 == parts ? h.size() : std::min(h.size(), lo + B);
        std::vector<std::complex<double>> hp(h.begin() + (std::ptrdiff_t)std::min(lo, h.size()), h.begin() + (std::ptrdiff_t)hi);
        if (hp.empty()) hp.push_back(0.0);
        T[(size_t)p] = turn_spectrum_lanes(make_hspec(hp, 4096));
    }
    std::vector<float> out(fir_upols_table_bytes(parts) / sizeof(float));
    size_t o = 0;
    for (int g = 0; 2 * g < parts; g++) {
        const bool pair = 2 * g + 1 < parts;
        for (size_t e = 0; e < 4096; e++) {
            out[o++] = T[(size_t)(2 * g)][2 * e];
            out[o++] = T[(size_t)(2 * g)][2 * e + 1];
            if (pair) {
                out[o++] = T[(size_t)(2 * g + 1)][2 * e];
                out[o++] = T[(size_t)(2 * g + 1)][2 * e + 1];
            }
        }
    }
    return out;
}


}  // namespace pcx
