// fir_designer.cpp -- /comms/fir_designer (+ /blocks/fir_designer): the step in front of /comms/fir_filter
// (SURVEY 8f rank 2).  Host-side only: no stream ports, one "tapsChanged" signal carrying the tap vector,
// emitted on activation and whenever a parameter changes while active (filter/FIRDesigner.cpp:142-200,
// 387-477), wired by topologies as  connect(designer, "tapsChanged", filter, "setTaps")
// (filter/TestFIRFilter.cpp:33-48, filter/TestFIRDesigner.cpp:150-170).
//
// Same registry paths, constructor defaults, setters/getters, parameter checks (and their order and
// messages), backwards-compatible filter-type aliases, gain-then-window order and signal payload types
// (std::vector<double>, or std::vector<std::complex<double>> for the COMPLEX_* band types) as the reference.
//
// What is built: every filter type the reference lists -- "SINC", "GAUSSIAN" (the reference's constructor default, so a
// default-constructed designer activates and emits taps as the reference's does), "RAISED_COSINE", "ROOT_RAISED_COSINE",
// "MAXFLAT" and "REMEZ" -- for all six band types, with every window the reference lists.  The reference delegates the
// arithmetic to spuce (design_fir / design_complex_fir / design_window), an un-vendored dependency that is absent from
// the reference tree, so tap VALUES follow the textbook definitions below and are "parity unpinned" against spuce; what
// is pinned is the reference's own acceptance test (TestFIRDesigner.cpp:110-135: pass points above -30 dB, stop points
// below -80 dB of an impulse's spectrum), the windows against scipy.signal.windows, and the defining properties of each
// prototype (tests/test_designer_cpu.py: Nyquist zero crossings of the raised cosine, the root raised cosine convolved
// with itself, the Gaussian's -3 dB point).  "REMEZ" is a Parks-McClellan exchange of our own (checked against
// scipy.signal.remez: the minimax solution is unique) with the pass band ending at the lower frequency and the stop band
// starting one transition bandwidth above it; "MAXFLAT" is Herrmann's maximally flat design (odd tap counts).
#include <algorithm>
#include <cmath>
#include <complex>
#include <stdexcept>
#include <string>
#include <vector>

#include "pcx_framework.hpp"

using pcxfw::Block;
using pcxfw::Exception;
using pcxfw::InvalidArgumentException;

namespace {

const double kPi = 3.14159265358979323846264338327950288;

double besselI0(double x)
{
    // power series: sum ((x/2)^k / k!)^2, converges for every x used as a Kaiser beta
    double sum = 1.0, term = 1.0;
    const double h = 0.5 * x;
    for (int k = 1; k < 200; k++) {
        term *= h / k;
        const double t2 = term * term;
        sum += t2;
        if (t2 < 1e-18 * sum) break;
    }
    return sum;
}

// Chebyshev polynomial T_n(x) for any real x
double chebPoly(int n, double x)
{
    if (std::fabs(x) <= 1.0) return std::cos(n * std::acos(x));
    const double v = std::cosh(n * std::acosh(std::fabs(x)));
    return (x < 0 && (n & 1)) ? -v : v;
}

// window of `n` points; `arg` = Kaiser beta / Chebyshev attenuation in dB (windowArgs[0], 0 when absent)
std::vector<double> designWindow(const std::string &type, size_t n, double arg)
{
    std::vector<double> w(n, 1.0);
    if (n == 1 || type == "rectangular") return w;
    const double m = (double)(n - 1);
    if (type == "hann" || type == "hanning") {
        // the form without zero end points: every designed tap contributes
        for (size_t i = 0; i < n; i++) w[i] = 0.5 * (1.0 - std::cos(2.0 * kPi * (double)(i + 1) / (double)(n + 1)));
    } else if (type == "hamming") {
        for (size_t i = 0; i < n; i++) w[i] = 0.54 - 0.46 * std::cos(2.0 * kPi * i / m);
    } else if (type == "blackman") {
        for (size_t i = 0; i < n; i++) w[i] = 0.42 - 0.5 * std::cos(2.0 * kPi * i / m) + 0.08 * std::cos(4.0 * kPi * i / m);
    } else if (type == "bartlett") {
        for (size_t i = 0; i < n; i++) w[i] = 1.0 - std::fabs(2.0 * i / m - 1.0);
    } else if (type == "flattop") {
        const double a[5] = {0.21557895, 0.41663158, 0.277263158, 0.083578947, 0.006947368};
        for (size_t i = 0; i < n; i++) {
            const double t = 2.0 * kPi * i / m;
            w[i] = a[0] - a[1] * std::cos(t) + a[2] * std::cos(2 * t) - a[3] * std::cos(3 * t) + a[4] * std::cos(4 * t);
        }
    } else if (type == "kaiser") {
        const double d = besselI0(arg);
        for (size_t i = 0; i < n; i++) {
            const double x = 2.0 * i / m - 1.0;
            w[i] = besselI0(arg * std::sqrt(std::max(0.0, 1.0 - x * x))) / d;
        }
    } else if (type == "chebyshev") {
        // Dolph-Chebyshev: W(k) = T_{n-1}(x0 cos(pi k / n)), x0 = cosh(acosh(10^(atten/20)) / (n-1));
        // inverse DFT about the window's centre (a half-sample phase for even n), normalised to a peak of 1
        const double x0 = std::cosh(std::acosh(std::pow(10.0, std::fabs(arg) / 20.0)) / m);
        std::vector<double> W(n);
        for (size_t k = 0; k < n; k++) W[k] = chebPoly((int)n - 1, x0 * std::cos(kPi * (double)k / (double)n));
        double peak = 0;
        for (size_t i = 0; i < n; i++) {
            double acc = 0;
            for (size_t k = 0; k < n; k++) {
                // centre the window: sample i sits at i - (n-1)/2
                const double ph = 2.0 * kPi * (double)k * ((double)i - 0.5 * m) / (double)n;
                acc += W[k] * std::cos(ph);
            }
            w[i] = acc;
            peak = std::max(peak, std::fabs(acc));
        }
        for (size_t i = 0; i < n; i++) w[i] /= peak;
    } else {
        throw InvalidArgumentException("FIRDesigner::setWindowType(" + type + ")", "unknown window type");
    }
    return w;
}

// truncated sin(x)/x low-pass with cut-off fc cycles/sample, unity gain at DC in the limit of many taps
std::vector<double> sincLowPass(size_t n, double fc)
{
    std::vector<double> h(n);
    const double c = 0.5 * (double)(n - 1);
    for (size_t i = 0; i < n; i++) {
        const double t = (double)i - c;
        const double x = 2.0 * kPi * fc * t;
        h[i] = std::fabs(t) < 1e-12 ? 2.0 * fc : std::sin(x) / (kPi * t);
    }
    return h;
}

// Raised cosine low-pass with its -6 dB point at fc cycles/sample (symbol period T = 1/(2 fc) samples) and excess
// bandwidth alpha in [0, 1]: h(t) = 2 fc sinc(2 fc t) cos(2 pi alpha fc t) / (1 - (4 alpha fc t)^2); zero at every
// multiple of T except t = 0 (no inter-symbol interference).  Unity gain at DC.
std::vector<double> raisedCosineLowPass(size_t n, double fc, double alpha)
{
    std::vector<double> h(n);
    const double c = 0.5 * (double)(n - 1);
    for (size_t i = 0; i < n; i++) {
        const double t = (double)i - c, x = 2.0 * fc * t;                 // x = t / T
        const double sinc = std::fabs(x) < 1e-12 ? 1.0 : std::sin(kPi * x) / (kPi * x);
        const double d = 1.0 - (2.0 * alpha * x) * (2.0 * alpha * x);
        // at |2 alpha x| = 1 numerator and denominator vanish together: the limit is (pi/4) sinc(1/(2 alpha))
        const double shape = std::fabs(d) < 1e-9 ? (kPi / 4.0) * (std::sin(kPi / (2.0 * alpha)) / (kPi / (2.0 * alpha)))
                                                 : sinc * std::cos(kPi * alpha * x) / d;
        h[i] = 2.0 * fc * shape;
    }
    return h;
}
// Root raised cosine: the filter whose convolution with itself is the raised cosine above (matched-filter pair).
//   h(t) = 2 fc [sin(pi x (1-a)) + 4 a x cos(pi x (1+a))] / [pi x (1 - (4 a x)^2)],  x = t / T = 2 fc t
std::vector<double> rootRaisedCosineLowPass(size_t n, double fc, double alpha)
{
    std::vector<double> h(n);
    const double c = 0.5 * (double)(n - 1), a = alpha;
    for (size_t i = 0; i < n; i++) {
        const double x = 2.0 * fc * ((double)i - c);
        double v;
        if (std::fabs(x) < 1e-12) v = 1.0 + a * (4.0 / kPi - 1.0);
        else if (a > 0 && std::fabs(std::fabs(4.0 * a * x) - 1.0) < 1e-9)
            v = (a / std::sqrt(2.0)) * ((1.0 + 2.0 / kPi) * std::sin(kPi / (4.0 * a)) + (1.0 - 2.0 / kPi) * std::cos(kPi / (4.0 * a)));
        else
            v = (std::sin(kPi * x * (1.0 - a)) + 4.0 * a * x * std::cos(kPi * x * (1.0 + a))) / (kPi * x * (1.0 - (4.0 * a * x) * (4.0 * a * x)));
        h[i] = 2.0 * fc * v;
    }
    return h;
}
// Gaussian low-pass whose -3 dB bandwidth is `bt` cycles/sample ("Lower Freq specifies the time-bandwidth product",
// FIRDesigner.cpp:38, with the sample as the unit of time): h(t) = sqrt(2 pi / ln 2) bt exp(-2 pi^2 bt^2 t^2 / ln 2).
// Unity gain at DC in the limit of many taps.
std::vector<double> gaussianLowPass(size_t n, double bt)
{
    std::vector<double> h(n);
    const double c = 0.5 * (double)(n - 1), ln2 = std::log(2.0);
    for (size_t i = 0; i < n; i++) {
        const double t = (double)i - c;
        h[i] = std::sqrt(2.0 * kPi / ln2) * bt * std::exp(-2.0 * kPi * kPi * bt * bt * t * t / ln2);
    }
    return h;
}
// amplitude samples A(k/n) of a symmetric n-tap filter -> its taps.  A(f), f in [0, 1/2], is the zero-phase response;
// beyond 1/2 it continues as A(1-f) for odd n and -A(1-f) for even n (half-sample delay), which makes the sum real.
template <typename F>
std::vector<double> tapsFromAmplitude(size_t n, F &&amp)
{
    std::vector<double> a(n);
    for (size_t k = 0; k < n; k++) {
        const double f = (double)k / (double)n;
        a[k] = f <= 0.5 ? amp(f) : ((n & 1) ? amp(1.0 - f) : -amp(1.0 - f));
    }
    std::vector<double> h(n);
    const double c = 0.5 * (double)(n - 1);
    for (size_t i = 0; i < n; i++) {
        double acc = 0;
        for (size_t k = 0; k < n; k++) acc += a[k] * std::cos(2.0 * kPi * (double)k * ((double)i - c) / (double)n);
        h[i] = acc / (double)n;
    }
    return h;
}

// Maximally flat (Herrmann) low-pass, odd n = 2M+1:  A(w) = cos^2K(w/2) sum_{k<L} C(K+k-1, k) sin^2k(w/2),  K + L - 1 = M:
// 2K zeros at w = pi, 2L-1 vanishing derivatives at w = 0, monotone in between.  K follows the half-amplitude frequency
// fc (cycles/sample): K = round((M+1) cos^2(pi fc)), kept inside 1..M.  Every term is positive: no cancellation.
std::vector<double> maxflatLowPass(size_t n, double fc)
{
    if ((n & 1) == 0 || n < 3) throw std::runtime_error("maximally flat design needs an odd number of taps, at least 3");
    const long M = (long)(n - 1) / 2;
    const double cc = std::cos(kPi * fc);
    long K = std::lround((double)(M + 1) * cc * cc);
    K = std::max(1L, std::min(M, K));
    const long L = M + 1 - K;
    return tapsFromAmplitude(n, [&](double f) {
        const double c2 = std::cos(kPi * f) * std::cos(kPi * f), s2 = 1.0 - c2;
        double sum = 0.0, term = 1.0;                     // term = C(K+k-1, k) s2^k
        for (long k = 0; k < L; k++) {
            sum += term;
            term *= s2 * (double)(K + k) / (double)(k + 1);
        }
        return std::pow(c2, (double)K) * sum;
    });
}

// Equiripple (Parks-McClellan) low-pass: n symmetric taps, pass band [0, fp], stop band [fs, 1/2] (cycles/sample), error
// weight 1 in the pass band and `wstop` in the stop band.  The Remez exchange on a dense grid with barycentric Lagrange
// interpolation in x = cos(2 pi f); even n (half-sample delay) is designed as cos(pi f) times a cosine polynomial.
std::vector<double> remezLowPass(size_t n, double fp, double fs, double wstop)
{
    if (!(fp > 0.0 && fs > fp && fs < 0.5)) throw std::runtime_error("remez band edges outside 0 < pass < stop < 1/2");
    if (!(wstop > 0.0)) throw std::runtime_error("remez weight must be positive");
    const bool odd = (n & 1) != 0;
    const size_t r = odd ? (n + 1) / 2 : n / 2;           // cosine terms of the polynomial part
    if (r < 2) throw std::runtime_error("remez needs at least 3 taps");
    const double delf = 0.5 / (16.0 * (double)r);
    std::vector<double> gf, D, W;
    auto band = [&](double lo, double hi, double des, double wt) {
        if (!odd && hi > 0.5 - delf) hi = 0.5 - delf;     // cos(pi f) vanishes at 1/2
        const size_t k = (size_t)std::max(1.0, std::floor((hi - lo) / delf + 0.5));
        for (size_t i = 0; i <= k; i++) {
            const double f = i == k ? hi : lo + (double)i * delf;
            const double q = odd ? 1.0 : std::cos(kPi * f);
            gf.push_back(f); D.push_back(des / q); W.push_back(wt * q);
        }
    };
    band(0.0, fp, 1.0, 1.0);
    band(fs, 0.5, 0.0, wstop);
    const size_t G = gf.size();
    if (G < r + 1) throw std::runtime_error("remez grid too coarse for the bands");
    std::vector<double> x(G), E(G);
    for (size_t g = 0; g < G; g++) x[g] = std::cos(2.0 * kPi * gf[g]);
    std::vector<size_t> ext(r + 1);
    for (size_t i = 0; i <= r; i++) ext[i] = i * (G - 1) / r;
    std::vector<double> bw(r + 1), cv(r + 1);
    double delta = 0.0;
    auto fit = [&]() {                                    // delta and the node values of the current extremal set
        for (size_t i = 0; i <= r; i++) {
            double p = 1.0;
            for (size_t j = 0; j <= r; j++)
                if (j != i) p *= 2.0 * (x[ext[i]] - x[ext[j]]);
            bw[i] = 1.0 / p;
        }
        double num = 0.0, den = 0.0, sgn = 1.0;
        for (size_t i = 0; i <= r; i++, sgn = -sgn) {
            num += bw[i] * D[ext[i]];
            den += bw[i] * sgn / W[ext[i]];
        }
        delta = num / den;
        sgn = 1.0;
        for (size_t i = 0; i <= r; i++, sgn = -sgn) cv[i] = D[ext[i]] - sgn * delta / W[ext[i]];
    };
    auto eval = [&](double xv) {                          // the interpolating polynomial at x
        double num = 0.0, den = 0.0;
        for (size_t i = 0; i <= r; i++) {
            const double d = xv - x[ext[i]];
            if (std::fabs(d) < 1e-15) return cv[i];
            num += bw[i] * cv[i] / d;
            den += bw[i] / d;
        }
        return num / den;
    };
    for (int iter = 0; iter < 60; iter++) {
        fit();
        for (size_t g = 0; g < G; g++) E[g] = W[g] * (D[g] - eval(x[g]));
        // local extrema of the error (band ends count), then strict alternation keeping the larger of equal-signed neighbours
        std::vector<size_t> cand;
        for (size_t g = 0; g < G; g++) {
            const double e = E[g], l = g > 0 ? E[g - 1] : (e > 0 ? -1e300 : 1e300), rr = g + 1 < G ? E[g + 1] : (e > 0 ? -1e300 : 1e300);
            const bool peak = e > 0 ? (e >= l && e >= rr) : (e <= l && e <= rr);
            if (!peak || e == 0.0) continue;
            if (!cand.empty() && (E[cand.back()] > 0) == (e > 0)) {
                if (std::fabs(e) > std::fabs(E[cand.back()])) cand.back() = g;
            } else {
                cand.push_back(g);
            }
        }
        while (cand.size() > r + 1) {                     // too many: the weaker end goes
            if (std::fabs(E[cand.front()]) < std::fabs(E[cand.back()])) cand.erase(cand.begin());
            else cand.pop_back();
        }
        if (cand.size() < r + 1) throw std::runtime_error("remez exchange lost an extremum (specification too loose for the tap count?)");
        const bool same = std::equal(cand.begin(), cand.end(), ext.begin());
        ext = cand;
        if (same) break;
    }
    fit();
    return tapsFromAmplitude(n, [&](double f) {
        const double q = odd ? 1.0 : std::cos(kPi * f);
        return q == 0.0 ? 0.0 : q * eval(std::cos(2.0 * kPi * f));
    });
}
// spuce's remez_estimate_* as FIRDesigner.cpp:423-438 uses them (spuce is absent: the usual Herrmann / Kaiser estimates)
double rippleOfPassDB(double passDB) { const double g = std::pow(10.0, passDB / 20.0); return (g - 1.0) / (g + 1.0); }
double rippleOfStopDB(double stopDB) { return std::pow(10.0, -stopDB / 20.0); }
double remezEstimateWeight(double passDB, double stopDB) { return rippleOfPassDB(passDB) / rippleOfStopDB(stopDB); }
size_t remezEstimateNumTaps(double transBw, double passDB, double stopDB)
{
    const double d = -20.0 * std::log10(std::sqrt(rippleOfPassDB(passDB) * rippleOfStopDB(stopDB)));
    return (size_t)std::max(1.0, std::ceil((d - 13.0) / (14.6 * transBw) + 1.0));
}

// the low-pass prototype of a filter type at cut-off fc.  REMEZ: pass band to fc, stop band from fc + alpha (alpha = the
// transition bandwidth in cycles/sample), stop-band weight `weight`; MAXFLAT: half-amplitude point at fc.
std::vector<double> prototypeLowPass(const std::string &type, size_t n, double fc, double alpha, double weight)
{
    if (type == "SINC") return sincLowPass(n, fc);
    if (type == "GAUSSIAN") return gaussianLowPass(n, fc);
    if (type == "RAISED_COSINE") return raisedCosineLowPass(n, fc, alpha);
    if (type == "MAXFLAT") return maxflatLowPass(n, fc);
    if (type == "REMEZ") return remezLowPass(n, fc, fc + alpha, weight);
    return rootRaisedCosineLowPass(n, fc, alpha);
}

/***********************************************************************
 * |PothosDoc FIR Designer
 *
 * Computes FIR filter taps on the host and publishes them: the block has no stream ports, it emits the signal
 * "tapsChanged" -- an array of taps -- when it is activated and whenever one of its settings changes.  Connect the
 * signal to the setTaps slot of a FIR filter (set that filter's Wait Taps to hold its stream until the first taps arrive).
 *
 * |category /Filter
 * |keywords fir filter taps highpass lowpass bandpass remez designer
 * |alias /blocks/fir_designer
 *
 * |param type[Filter Type] The prototype the taps are derived from.
 * <ul>
 * <li>SINC (box-car): a truncated sin(x)/x, shaped by the window.</li>
 * <li>RAISED_COSINE, ROOT_RAISED_COSINE: pulse-shaping responses with roll-off factor alpha; Lower Freq is the symbol rate.</li>
 * <li>MAXFLAT: maximally flat, non-linear phase; only discrete cut-off frequencies are reachable.</li>
 * <li>GAUSSIAN: Lower Freq sets the time-bandwidth product.</li>
 * <li>REMEZ: equiripple (Parks-McClellan) from the transition width, ripple and attenuation on the Remez tab.</li>
 * </ul>
 * |option [Root Raised Cosine] "ROOT_RAISED_COSINE"
 * |option [Raised Cosine] "RAISED_COSINE"
 * |option [Box-Car] "SINC"
 * |option [Maxflat] "MAXFLAT"
 * |option [Gaussian] "GAUSSIAN"
 * |option [Remez] "REMEZ"
 * |default "SINC"
 *
 * |param band[Band Type] Which band passes.  The complex variants give one-sided (complex) taps.
 * |option [Low Pass] "LOW_PASS"
 * |option [High Pass] "HIGH_PASS"
 * |option [Band Pass] "BAND_PASS"
 * |option [Band Stop] "BAND_STOP"
 * |option [Complex Band Pass] "COMPLEX_BAND_PASS"
 * |option [Complex Band Stop] "COMPLEX_BAND_STOP"
 *
 * |param window[Window Type] The window laid over the prototype; it trades transition width against ripple.
 * |default "hann"
 * |option [Rectangular] "rectangular"
 * |option [Hann] "hann"
 * |option [Hamming] "hamming"
 * |option [Blackman] "blackman"
 * |option [Bartlett] "bartlett"
 * |option [Flat-top] "flattop"
 * |option [Kaiser] "kaiser"
 * |option [Chebyshev] "chebyshev"
 * |tab Window
 *
 * |param windowArgs[Window Args] Arguments of the parameterised windows, as a list:
 * [beta] for Kaiser, [attenuation in dB] for Chebyshev; ignored by the others.
 * |default []
 * |preview valid
 * |tab Window
 *
 * |param gain[Gain] Factor applied to every tap.
 * |default 1.0
 *
 * |param sampRate[Sample Rate] Sample rate of the stream the taps are for; every frequency below must lie under half of it.
 * |default 1e6
 * |units Sps
 *
 * |param freqLower[Lower Freq] The transition frequency of low- and high-pass filters, the lower edge of the band types,
 * the symbol rate of the cosine filters, the time-bandwidth product of the Gaussian.
 * |default 1000
 * |units Hz
 *
 * |param freqUpper[Upper Freq] The upper edge of band-pass and band-stop filters.
 * |default 2000
 * |units Hz
 * |preview when(enum=band, "BAND_PASS", "BAND_STOP", "COMPLEX_BAND_PASS", "COMPLEX_BAND_STOP")
 *
 * |param transBw[Transition Width] Width of the transition band of a Remez design.
 * |default 1000
 * |units Hz
 * |preview when(enum=type, "REMEZ")
 * |tab Remez
 *
 * |param numTaps[Num Taps] How many taps to produce: the filter's length, and its cost per sample.
 * |default 51
 * |widget SpinBox(minimum=1)
 *
 * |param alpha[Alpha] Roll-off (excess bandwidth) of the cosine filters, 0.0 ... 1.0.
 * |default 0.5
 * |preview when(enum=type, "RAISED_COSINE", "ROOT_RAISED_COSINE")
 * |tab Cosine
 *
 * |param stopDB[Attenuation] Stop-band attenuation a Remez design aims for.
 * |default 60.0
 * |units dB
 * |preview when(enum=type, "REMEZ")
 * |tab Remez
 *
 * |param passDB[Passband Ripple] Pass-band ripple a Remez design may leave.
 * |default 0.1
 * |units dB
 * |preview when(enum=type, "REMEZ")
 * |tab Remez
 *
 * |factory /comms/fir_designer()
 * |setter setFilterType(type)
 * |setter setBandType(band)
 * |setter setWindowType(window)
 * |setter setWindowArgs(windowArgs)
 * |setter setSampleRate(sampRate)
 * |setter setFrequencyLower(freqLower)
 * |setter setFrequencyUpper(freqUpper)
 * |setter setBandwidthTrans(transBw)
 * |setter setNumTaps(numTaps)
 * |setter setAlpha(alpha)
 * |setter setStopDB(stopDB)
 * |setter setPassDB(passDB)
 * |setter setGain(gain)
 **********************************************************************/
class FIRDesigner : public Block {
public:
    static Block *make() { return new FIRDesigner(); }

    // constructor defaults of FIRDesigner.cpp:148-161
    FIRDesigner()
        : _filterType("GAUSSIAN"), _bandType("LOW_PASS"), _windowType("hann"), _gain(1.0), _sampRate(1.0), _freqLower(0.1),
          _freqUpper(0.2), _transBw(0.1), _alpha(0.5), _stopDB(60.0), _passDB(0.1), _numTaps(51)
    {
        this->registerCall(this, "setBandType", &FIRDesigner::setBandType);
        this->registerCall(this, "bandType", &FIRDesigner::bandType);
        this->registerCall(this, "setFilterType", &FIRDesigner::setFilterType);
        this->registerCall(this, "filterType", &FIRDesigner::filterType);
        this->registerCall(this, "setWindowType", &FIRDesigner::setWindowType);
        this->registerCall(this, "windowType", &FIRDesigner::windowType);
        this->registerCall(this, "setWindowArgs", &FIRDesigner::setWindowArgs);
        this->registerCall(this, "windowArgs", &FIRDesigner::windowArgs);
        this->registerCall(this, "setSampleRate", &FIRDesigner::setSampleRate);
        this->registerCall(this, "sampleRate", &FIRDesigner::sampleRate);
        this->registerCall(this, "setFrequencies", &FIRDesigner::setFrequencies);
        this->registerCall(this, "setFrequencyLower", &FIRDesigner::setFrequencyLower);
        this->registerCall(this, "frequencyLower", &FIRDesigner::frequencyLower);
        this->registerCall(this, "setFrequencyUpper", &FIRDesigner::setFrequencyUpper);
        this->registerCall(this, "frequencyUpper", &FIRDesigner::frequencyUpper);
        this->registerCall(this, "setBandwidthTrans", &FIRDesigner::setBandwidthTrans);
        this->registerCall(this, "bandwidthTrans", &FIRDesigner::bandwidthTrans);
        this->registerCall(this, "setNumTaps", &FIRDesigner::setNumTaps);
        this->registerCall(this, "numTaps", &FIRDesigner::numTaps);
        this->registerCall(this, "setAlpha", &FIRDesigner::setAlpha);
        this->registerCall(this, "alpha", &FIRDesigner::alpha);
        this->registerCall(this, "setStopDB", &FIRDesigner::setStopDB);
        this->registerCall(this, "stopDB", &FIRDesigner::stopDB);
        this->registerCall(this, "setPassDB", &FIRDesigner::setPassDB);
        this->registerCall(this, "passDB", &FIRDesigner::passDB);
        this->registerCall(this, "lastWarning", &FIRDesigner::lastWarning);   // (ours: the text the reference logs)
        this->registerCall(this, "setGain", &FIRDesigner::setGain);
        this->registerCall(this, "gain", &FIRDesigner::gain);
        this->registerSignal("tapsChanged");
        this->recalculate();
    }

    void setFilterType(const std::string &type)
    {
        // band-type names were filter types once: kept working, as the reference does (FIRDesigner.cpp:197-212)
        if (type == "LOW_PASS" || type == "HIGH_PASS" || type == "BAND_PASS" || type == "BAND_STOP" ||
            type == "COMPLEX_BAND_PASS" || type == "COMPLEX_BAND_STOP") {
            _filterType = "SINC";
            _bandType = type;
            this->recalculate();
            return;
        }
        _filterType = type;
        this->recalculate();
    }
    std::string filterType() const { return _filterType; }
    void setBandType(const std::string &type) { _bandType = type; this->recalculate(); }
    std::string bandType() const { return _bandType; }
    void setWindowType(const std::string &type) { _windowType = type; this->recalculate(); }
    std::string windowType() const { return _windowType; }
    void setWindowArgs(const std::vector<double> &args) { _windowArgs = args; this->recalculate(); }
    std::vector<double> windowArgs() const { return _windowArgs; }
    void setSampleRate(const double rate) { _sampRate = rate; this->recalculate(); }
    double sampleRate() const { return _sampRate; }
    void setFrequencies(const std::vector<double> &freqs)
    {
        if (freqs.size() > 0) _freqLower = freqs.at(0);
        if (freqs.size() > 1) _freqUpper = freqs.at(1);
        this->recalculate();
    }
    void setFrequencyLower(const double freq) { _freqLower = freq; this->recalculate(); }
    double frequencyLower() const { return _freqLower; }
    void setFrequencyUpper(const double freq) { _freqUpper = freq; this->recalculate(); }
    double frequencyUpper() const { return _freqUpper; }
    void setBandwidthTrans(const double freq) { _transBw = freq; this->recalculate(); }
    double bandwidthTrans() const { return _transBw; }
    void setNumTaps(const size_t num) { _numTaps = num; this->recalculate(); }
    size_t numTaps() const { return _numTaps; }
    void setAlpha(const double alpha) { _alpha = alpha; this->recalculate(); }
    double alpha() const { return _alpha; }
    void setPassDB(const double w) { _passDB = w; this->recalculate(); }
    double passDB() const { return _passDB; }
    std::string lastWarning() const { return _lastWarning; }
    void setStopDB(const double w) { _stopDB = w; this->recalculate(); }
    double stopDB() const { return _stopDB; }
    void setGain(const double gain) { _gain = gain; this->recalculate(); }
    double gain() const { return _gain; }

    void activate() override { this->recalculate(); }

private:
    void recalculate();

    std::string _filterType, _bandType, _windowType;
    std::vector<double> _windowArgs;
    double _gain, _sampRate, _freqLower, _freqUpper, _transBw, _alpha, _stopDB, _passDB;
    size_t _numTaps;
    std::string _lastWarning;    // what the reference sends to its logger (FIRDesigner.cpp:429-437): kept for the caller to read
};

void FIRDesigner::recalculate()
{
    if (!this->isActive()) return;

    const bool isComplex = _bandType.find("COMPLEX") != std::string::npos;
    const bool isStop = _bandType.find("STOP") != std::string::npos;
    const bool isBand = _bandType == "BAND_PASS" || _bandType == "BAND_STOP" || _bandType == "COMPLEX_BAND_PASS" ||
                        _bandType == "COMPLEX_BAND_STOP";

    // parameter checks in the reference's order (FIRDesigner.cpp:395-413)
    if (_numTaps == 0) throw Exception("FIRDesigner()", "num taps must be positive");
    if (_sampRate <= 0) throw Exception("FIRDesigner()", "sample rate must be positive");
    if (isComplex && _freqLower <= -_sampRate / 2) throw Exception("FIRDesigner()", "lower frequency below Nyquist range");
    if (!isComplex && _freqLower <= 0) throw Exception("FIRDesigner()", "lower frequency must be positive");
    if (_freqLower >= _sampRate / 2) throw Exception("FIRDesigner()", "lower frequency above Nyquist range");
    if (isBand) {
        if ((_numTaps % 2) == 0) throw Exception("FIRDesigner()", "Band pass or Band stop FIRs must have an odd number of taps");
        if (isComplex && _freqUpper <= -_sampRate / 2) throw Exception("FIRDesigner()", "upper frequency below Nyquist range");
        if (!isComplex && _freqUpper <= 0) throw Exception("FIRDesigner()", "upper frequency must be positive");
        if (_freqUpper >= _sampRate / 2) throw Exception("FIRDesigner()", "upper frequency above Nyquist range");
        if (_freqUpper <= _freqLower) throw Exception("FIRDesigner()", "upper frequency <= lower frequency");
    }
    if (_filterType == "MAXFLAT" && isStop)
        throw Exception("FIRDesigner()", "Can not use MAXFLAT as prototype for stop-band filter, please choose another type");
    if (_filterType == "REMEZ") {
        if (_transBw <= 0) throw Exception("FIRDesigner()", "Transition Bandwidth must be > 0");
        if (_passDB <= 0) throw Exception("FIRDesigner()", "Passband Attenuation must be > 0");
        if (_stopDB <= 0) throw Exception("FIRDesigner()", "Stopband Attenuation must be > 0");
    }
    const bool known = _filterType == "SINC" || _filterType == "GAUSSIAN" || _filterType == "RAISED_COSINE" || _filterType == "ROOT_RAISED_COSINE" ||
                       _filterType == "MAXFLAT" || _filterType == "REMEZ";
    if (!known)
        throw InvalidArgumentException("Problem with creating taps for FIRDesigner(" + _filterType + "/" + _bandType + "): unknown filter type",
                                       "problem with input parameters?");
    double alpha = _alpha, weight = 1.0;
    if (_filterType == "REMEZ") {
        // FIRDesigner.cpp:421-439: alpha becomes the transition bandwidth, the weight follows the two ripples, and a tap
        // count below the estimate is reported, not refused
        alpha = _transBw / _sampRate;
        const size_t est = remezEstimateNumTaps(alpha, _passDB, _stopDB);
        if (est > _numTaps)
            _lastWarning = "Remez order not large enough to meet specification: increase filter order to " + std::to_string(est) + " taps";
        else
            _lastWarning.clear();
        weight = remezEstimateWeight(_passDB, _stopDB);
    }
    if ((_filterType == "RAISED_COSINE" || _filterType == "ROOT_RAISED_COSINE") && !(_alpha >= 0.0 && _alpha <= 1.0))
        throw InvalidArgumentException("Problem with creating taps for FIRDesigner(" + _filterType + "/" + _bandType + "): alpha outside 0.0 to 1.0",
                                       "problem with input parameters?");
    if (!(_bandType == "LOW_PASS" || _bandType == "HIGH_PASS" || isBand))
        throw InvalidArgumentException("Problem with creating taps for FIRDesigner(" + _filterType + "/" + _bandType + "): unknown band type",
                                       "problem with input parameters?");

    const double fl = _freqLower / _sampRate, fu = _freqUpper / _sampRate;   // cycles per sample
    const size_t n = _numTaps;
    const double c = 0.5 * (double)(n - 1);
    std::vector<double> taps;
    std::vector<std::complex<double>> complexTaps;
    try {
        if (_bandType == "LOW_PASS") {
            taps = prototypeLowPass(_filterType, n, fl, alpha, weight);
        } else if (_bandType == "HIGH_PASS") {
            // low-pass of width 1/2 - fl moved to the Nyquist frequency
            taps = prototypeLowPass(_filterType, n, 0.5 - fl, alpha, weight);
            for (size_t i = 0; i < n; i++) taps[i] *= std::cos(kPi * ((double)i - c));
        } else if (!isComplex) {
            // low-pass of half the band's width moved to +- the band centre; the stop form is its complement
            taps = prototypeLowPass(_filterType, n, 0.5 * (fu - fl), alpha, weight);
            const double f0 = 0.5 * (fu + fl);
            for (size_t i = 0; i < n; i++) taps[i] *= 2.0 * std::cos(2.0 * kPi * f0 * ((double)i - c));
            if (isStop) {
                for (size_t i = 0; i < n; i++) taps[i] = -taps[i];
                taps[n / 2] += 1.0;
            }
        } else {
            const std::vector<double> lp = prototypeLowPass(_filterType, n, 0.5 * (fu - fl), alpha, weight);
            const double f0 = 0.5 * (fu + fl);
            complexTaps.resize(n);
            for (size_t i = 0; i < n; i++) complexTaps[i] = lp[i] * std::polar(1.0, 2.0 * kPi * f0 * ((double)i - c));
            if (isStop) {
                for (size_t i = 0; i < n; i++) complexTaps[i] = -complexTaps[i];
                complexTaps[n / 2] += 1.0;
            }
        }
    } catch (const std::runtime_error &error) {
        // FIRDesigner.cpp:455-457
        throw InvalidArgumentException("Problem with creating taps for FIRDesigner(" + _filterType + "/" + _bandType + "):" + error.what(),
                                       "problem with input parameters?");
    }

    // gain, then the window (FIRDesigner.cpp:455-471)
    for (auto &t : complexTaps) t *= _gain;
    for (auto &t : taps) t *= _gain;
    const std::vector<double> window = designWindow(_windowType, n, _windowArgs.empty() ? 0.0 : _windowArgs.at(0));
    if (!complexTaps.empty()) {
        for (size_t i = 0; i < n; i++) complexTaps[i] *= window[i];
        this->emitSignal("tapsChanged", complexTaps);
    } else if (!taps.empty()) {
        for (size_t i = 0; i < n; i++) taps[i] *= window[i];
        this->emitSignal("tapsChanged", taps);
    }
}

pcxfw::BlockRegistry registerFIRDesigner("/comms/fir_designer", &FIRDesigner::make);
pcxfw::BlockRegistry registerFIRDesignerOldPath("/blocks/fir_designer", &FIRDesigner::make);

}  // namespace
