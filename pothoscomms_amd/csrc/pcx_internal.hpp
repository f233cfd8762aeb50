// pcx_internal.hpp -- shared host-side helpers of libpcx_hip.so (not installed)
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <atomic>
#include <cstring>
#include <vector>

#include "pcx.h"
#include "pcx_qformat.hpp"

namespace pcx {

void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

#define PCX_HIP(expr)                                                                              \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess) {                                                                   \
            ::pcx::set_error("%s: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return PCX_ERR_HIP;                                                                    \
        }                                                                                          \
    } while (0)

#define PCX_TRY(expr)              \
    do {                           \
        int rc__ = (expr);         \
        if (rc__ != PCX_OK) return rc__; \
    } while (0)

#define PCX_LAUNCH_CHECK() PCX_HIP(hipGetLastError())

inline int scalar_bytes(int s)
{
    switch (s) {
    case PCX_F64: case PCX_I64: case PCX_U64: return 8;
    case PCX_F32: case PCX_I32: case PCX_U32: return 4;
    case PCX_I16: case PCX_U16: return 2;
    case PCX_I8: case PCX_U8: return 1;
    }
    return 0;
}
inline bool valid_scalar(int s) { return s >= PCX_F64 && s <= PCX_I8; }
inline bool valid_arith_scalar(int s) { return s >= PCX_F64 && s <= PCX_U8; }   // + the unsigned types
inline bool is_float_scalar(int s) { return s == PCX_F64 || s == PCX_F32; }
// Q (accumulator) width for an integer element type: FIRFilter.cpp:377-382,
// Rotate.cpp:151-154, Scale.cpp:150-153
inline int q_bits(int s)
{
    switch (s) {
    case PCX_I64: case PCX_I32: return 64;
    case PCX_I16: return 32;
    case PCX_I8: return 16;
    }
    return 0;
}
// fractional bits of an integer element type's Q format under a reading: half the Q word, or half the element word
inline int q_frac_bits(const QFormat &f, int scalar) { return f.frac == PCX_Q_FRAC_HALF_ELEM ? 4 * scalar_bytes(scalar) : q_bits(scalar) / 2; }
inline QShift q_shift(const QFormat &f, int scalar) { return QShift{q_frac_bits(f, scalar), f.from}; }
// Pothos::Util::floatToQ<T>(x) for integer T of qbits bits: T(std::ldexp(x, frac_bits)), the value truncated by the cast
// (PCX_Q_TRUNCATE) or rounded to nearest first (PCX_Q_NEAREST)
int64_t float_to_q(double x, int qbits, int frac_bits, int rounding);
inline int64_t float_to_q(double x, int scalar, const QFormat &f) { return float_to_q(x, q_bits(scalar), q_frac_bits(f, scalar), f.to); }
QFormat process_qformat();                                   // pcx_set_qformat's current value
int qformat_from_api(const pcx_qformat *q, QFormat *out);    // validates; NULL -> the process-wide reading

// growable device workspace owned by a handle
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes);
    int ensure_zeroed(size_t bytes);   // ensure + zero fill, COMPLETE on return (control plane)
    void release();
    ~DevBuf() { release(); }
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
};

// page-locked host memory of the library's own (the bounce buffer of a pageable host-pointer call)
struct PinBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes);
    void release();
    ~PinBuf() { release(); }
    PinBuf() = default;
    PinBuf(const PinBuf &) = delete;
    PinBuf &operator=(const PinBuf &) = delete;
};
// ROCTx range around a C-ABI entry point while pcx_trace(1) is in force (pcx_api.hip); one acquire load otherwise (it pairs with the
// store that publishes the function pointers)
extern std::atomic<int> g_trace_on;
extern int (*g_roctx_push)(const char *);
extern int (*g_roctx_pop)();
struct TraceRange {
    bool on;
    explicit TraceRange(const char *name) : on(g_trace_on.load(std::memory_order_acquire) != 0) { if (on) (void)g_roctx_push(name); }
    ~TraceRange() { if (on) (void)g_roctx_pop(); }
    TraceRange(const TraceRange &) = delete;
    TraceRange &operator=(const TraceRange &) = delete;
};
#define PCX_TRACE() ::pcx::TraceRange pcx_trace_range_(__func__)

// (pcx_fir_api.hip, for pcx_shard.hip) upload a FIR handle's tables now instead of at its next call
int fir_prepare(struct ::pcx_fir *h);
int fmchain_prepare(struct ::pcx_fmchain *h);
// (pcx_shard.hip) resident workgroups a handle's persistent launches may take: 1024 / the number of shards that share the device
void fir_set_slots(struct ::pcx_fir *h, unsigned slots);
void fmchain_set_slots(struct ::pcx_fmchain *h, unsigned slots);
// device-visible alias of a host pointer when it is page-locked (pcx_api.hip), else nullptr
void *device_alias(const void *p);
// staging pair of one direction of a host-pointer call (pcx_api.hip stage_in / stage_out_*)
struct StageBuf {
    DevBuf dev;
    PinBuf pin;
    void release();
    StageBuf() = default;
    StageBuf(const StageBuf &) = delete;
    StageBuf &operator=(const StageBuf &) = delete;
    ~StageBuf() { release(); }
};

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// bytes of a handle's block-dealer state (pcx_sched.hpp SchedState: sixteen counters a cache line apart + the exit count)
constexpr size_t kSchedBytes = 4096;

// A/B and diagnostic switches.  The PRODUCT library (libpcx_hip.so) ignores the process environment
// entirely: every switch is its measured-best default, compiled in.  Only the diagnostic build
// (make diag -> libpcx_hip_diag.so, -DPCX_DIAG, loaded by tools/ through PCX_HIP_LIBRARY) reads the
// PCX_* variables -- once per process, cached in a function-local static.
#ifdef PCX_DIAG
#include <cstdlib>
#define PCX_ENV_INT(name, dflt) ([]() -> long { static const long v_ = []() -> long { const char *e_ = getenv(name); return e_ ? atol(e_) : (long)(dflt); }(); return v_; }())
#define PCX_ENV_SET(name) ([]() -> bool { static const bool v_ = getenv(name) != nullptr; return v_; }())
#else
inline long env_off(long v) { return v; }
inline bool env_off(bool v) { return v; }
#define PCX_ENV_INT(name, dflt) (::pcx::env_off((long)(dflt)))
#define PCX_ENV_SET(name) (::pcx::env_off(false))
#endif

// LINK-BOUND LAUNCHES.  A kernel launched by a host-pointer entry point on page-locked HOST memory is bound by the PCIe link, which is
// full duplex: what matters is that reads and writes overlap, and they do when FEW workgroups each walk MANY pieces (a piece's loads then
// run beside the previous piece's stores) instead of a device-filling grid whose workgroups all load, then all store.  The host entry
// points (pcx_api.hip) set these two for the duration of their launch -- thread-local, 0 = the device-resident shape:
//   g_link_grid      persistent block kernels (overlap-save FIR plans, the fused chain): that many workgroups on the grid stride, no dealer
//   g_link_map_grid  grid-stride map kernels: that many blocks
// Measured over PCIe (profiles/r05/host_grid_sweep.txt, host_other_sweep.txt): FIR 255 taps +9-14 % at 48 workgroups (decimating / interpolating /
// real-stream plans and the fused chain +4-11 % on large calls); /comms/conjugate +15-29 % at 32 blocks, /comms/freq_demod +6-21 % at 64.
extern thread_local unsigned g_link_grid, g_link_map_grid;
struct LinkBoundScope {
    unsigned keep_grid, keep_map;
    LinkBoundScope(unsigned grid, unsigned map_grid) : keep_grid(g_link_grid), keep_map(g_link_map_grid) { g_link_grid = grid; g_link_map_grid = map_grid; }
    ~LinkBoundScope() { g_link_grid = keep_grid; g_link_map_grid = keep_map; }
    LinkBoundScope(const LinkBoundScope &) = delete;
    LinkBoundScope &operator=(const LinkBoundScope &) = delete;
};

// grid size for an HBM-bound grid-stride kernel.  The cap was 256 CUs x 8 resident blocks (every block the same share of the
// stream); with 8 queued per slot on top the dispatcher evens out the CUs' unequal rates: /comms/rotate on 64 Mi cf32 samples
// 0.1777 -> 0.1674..0.1681 ms (0.755 -> 0.80 of the HBM peak) at caps of 4096 ... 65536 (PCX_MAP_GRID, diagnostic library: A/B)
inline unsigned stream_grid(size_t work_items, unsigned block)
{
    size_t g = (work_items + block - 1) / block;
    const size_t cap = g_link_map_grid ? (size_t)g_link_map_grid : (size_t)PCX_ENV_INT("PCX_MAP_GRID", 256 * 64);
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

// grid of a persistent-workgroup kernel that walks `units` work items with a grid stride: `slots` = CUs x resident
// workgroups per CU, sized so that every workgroup gets the same number of rounds -- with units/grid = 17.06 an 18th
// round would run at 6 % occupancy.
// `oversub` > 1 queues that many workgroups per slot: the CUs do not all run at one rate (profiles/r02/ols_lab.md), and
// the hardware dispatcher then hands the next workgroup to whichever slot frees first.  It pays only where the per-workgroup
// set-up is small against a block (measured, tools/ab_oversub.sh: double-precision overlap-save +5 % at 8; decimating
// cf32 FIR -7 % at 4, interpolating and time-domain kernels +-0), so it is per kernel, default 1.  The three headline
// kernels have their own dealer instead (pcx_sched.hpp).  PCX_OVERSUB (diagnostic library only) overrides for A/B.
inline unsigned persistent_grid(size_t units, unsigned slots, unsigned oversub = 1)
{
    const long forced = PCX_ENV_INT("PCX_OVERSUB", 0);
    // (a link-bound launch, above: at most g_link_grid workgroups, equal rounds like any other persistent grid)
    const size_t cap = g_link_grid ? (size_t)g_link_grid : (size_t)slots * (size_t)(forced > 0 ? (unsigned)forced : oversub);
    if (units <= cap) return (unsigned)(units ? units : 1);
    const size_t rounds = (units + cap - 1) / cap;
    return (unsigned)((units + rounds - 1) / rounds);
}

// grid of a kernel whose workgroups should each walk about `rounds` work items, never fewer workgroups than the resident
// `slots`.  The radix-16 FFT family runs best at 2-4 items per workgroup WHATEVER the size of the call (tools/
// ab_fft_family_oversub.sh, profiles/r02/ab_fft_family_oversub.txt: enough to amortise the per-workgroup tables, short enough
// for the dispatcher to even out the CUs' unequal rates) -- a fixed oversubscription factor is right for one call size only.
// PCX_ROUNDS (diagnostic library only) overrides for A/B.
inline unsigned rounds_grid(size_t units, unsigned slots, unsigned rounds)
{
    const long forced = PCX_ENV_INT("PCX_ROUNDS", 0);
    if (forced > 0) rounds = (unsigned)forced;
    if (units <= slots) return (unsigned)(units ? units : 1);
    size_t g = (units + rounds - 1) / rounds;
    if (g < slots) return persistent_grid(units, slots);
    const size_t cap = (size_t)1 << 30;
    return (unsigned)(g < cap ? g : cap);
}

// ---- kernel launchers implemented in the .hip files ----
int launch_zero_words(void *p, size_t nwords, hipStream_t st);   // (elementwise.hip) a kernel, so that a captured reset replays
int launch_rotate(int scalar, double pr, double pi, const QFormat &qf, const void *in, void *out, size_t n, hipStream_t st);
int launch_scale(int scalar, int is_complex, double factor, const QFormat &qf, const void *in, void *out, size_t n, hipStream_t st);
int launch_abs(int scalar, int is_complex, const void *in, void *out, size_t n, hipStream_t st);
int launch_conj(int scalar, const void *in, void *out, size_t n, hipStream_t st);
int launch_angle(int scalar, const void *in, void *out, size_t n, hipStream_t st);
int launch_arith(int scalar, int is_complex, int op, const void *in0, const void *in1, void *out, size_t n, hipStream_t st);
int launch_split_complex(int scalar, const void *in, void *re, void *im, size_t n, hipStream_t st);
int launch_combine_complex(int scalar, const void *re, const void *im, void *out, size_t n, hipStream_t st);
// out[i] = angle(in[i]*_prev); _prev(i=0) := *prev_in (already conjugated); *prev_out := conj(in[n-1])
int launch_freqdemod(int scalar, const void *in, void *out, size_t n, const void *prev_in, void *prev_out, hipStream_t st);
int launch_fill_uniform_f32(float *dst, size_t n, uint64_t seed, uint64_t offset, hipStream_t st);
int launch_clock_probe(float *mhz_dev, unsigned spin_us, hipStream_t st);

// FIR: generic polyphase kernel (all types; EXACT = reference order, unfused)
struct FirGeom {
    size_t L, M, K;
    const uint32_t *rowLen;  // device, L entries
    const void *rowTaps;     // device, L*K entries of the tap type (Q precision), row-major
};
// (qs: the integer types' fromQ -- shift and rounding, pcx_qformat.hpp; ignored by the float types)
int launch_fir_generic(int scalar, int is_complex, int complex_taps, bool exact, const FirGeom &g, const void *in,
                       void *out, size_t n_out, QShift qs, hipStream_t st);
// complex_int16 stream, complex taps within int16 after floatToQ, M = L = 1: packed dot-product kernel; tapsP = 2K dwords
int launch_fir_ci16_dot2(const void *in, void *out, size_t n_out, size_t K, const void *tapsP, bool in8, QShift qs, hipStream_t st);
// M = L = 1, every type: register sliding window (fir_generic.hip); taps24 = all Q taps fit 24 signed bits
int launch_fir_slide(int scalar, int is_complex, int complex_taps, bool exact, bool taps24, const FirGeom &g, const void *in,
                     void *out, size_t n_out, QShift qs, hipStream_t st);
// FIR: fast LDS-tiled direct form, complex_float32, M=L=1.  taps_rev: device array of
// Kp (K rounded up to 8) cf32 taps in reversed order g[m] = h[K-1-m], zero padded
int launch_fir_cf32_direct(const void *in, size_t in_elems, void *out, size_t n_out, const void *taps_rev, size_t K,
                           size_t Kp, hipStream_t st);
// FIR: frequency-domain overlap-save, complex_float32, M=L=1, K <= 2049.
// Hspec: device array of 4096 cf32 = FFT_4096(h)/4096 in natural bin order.
// sched: the handle's SchedState pair (pcx_sched.hpp: dynamic block assignment) or nullptr for the static grid stride
int launch_fir_cf32_ols4096(const void *in, size_t in_elems, void *out, size_t n_out, const void *Hspec, size_t K,
                            const void *tw4096, void *sched, hipStream_t st, const void *gate_word = nullptr, unsigned gate_value = 0,
                            int *gated = nullptr, unsigned slots = 1024, size_t lead_valid = 0);
int launch_fir_cf32_ols4096_interp(const void *in, size_t in_elems, void *out, size_t n_iter, const void *Hspec, size_t K, size_t L,
                                   const void *tw4096, void *sched, hipStream_t st);
size_t fir_decim_fold_factor(size_t M);
int launch_fir_cf32_ols4096_decim(const void *in, size_t in_elems, void *out, size_t n_iter, const void *Hspec, size_t K, size_t M,
                                  const void *tw4096, void *sched, hipStream_t st);
// (fir_ols_part.hip) 2049 < K <= 8193: 4096-sample blocks, the taps in `parts` = ceil((K - 1) / 2048) partitions; complex_float32, or
// (real_stream) float32 with real taps; M > 1: a decimating filter (n_out full-rate outputs, one in M stored)
int launch_fir_cf32_upols(const void *in, size_t in_elems, void *out, size_t n_out, const void *Hparts, size_t K, int parts, const void *tw,
                          hipStream_t st, bool real_stream = false, size_t M = 1);
size_t fir_upols_table_bytes(int parts);
int launch_fir_cf64_ols(const void *in, size_t in_elems, void *out, size_t n_out, const void *Hspec, size_t K, int log2n,
                        const void *tw, int io, size_t M, QShift qs, hipStream_t st, void *sched = nullptr);
int launch_fir_real_ols(const void *in, size_t in_elems, void *out, size_t n_out, const void *Hspec, size_t K, int log2n, const void *tw,
                        int io, size_t M, QShift qs, hipStream_t st, void *sched = nullptr);

// real float32 stream, real taps, M=L=1: two real blocks per complex transform
int launch_fir_f32_ols4096(const void *in, size_t in_elems, void *out, size_t n_out, const void *Hspec, size_t K,
                           const void *tw4096, void *sched, hipStream_t st);
// same pipeline with interpolation L / decimation M: one launch per polyphase row;
// Hspec_rows: L spectra of 4096 cf32 (row j = FFT(taps[j + k*L]) / 4096); n_iter = inputs consumed
int launch_interleave_rows_cf32(const void *rows, void *out, size_t n, size_t L, hipStream_t st);
int launch_interleave_rows(const void *rows, void *out, size_t n, size_t L, size_t elem_bytes, size_t M, hipStream_t st);
int launch_fir_cf32_ols4096_poly(const void *in, size_t in_elems, void *out, size_t n_iter, const void *Hspec_rows, size_t K,
                                 size_t L, size_t M, const void *tw4096, hipStream_t st);

// FFT
// batched complex transpose (+ four-step twiddle): out[b][c][r] = in[b][r][c] * {1, W_N^(rc), W_N^(-rc)}, N = rows*cols
int launch_transpose(int scalar, const void *in, void *out, size_t rows, size_t cols, size_t batch, int mode, hipStream_t st);
// short four-step plans (complex_float32): column transforms (n1 in {128, 256}) with the twiddle; row transforms with transposed store
int launch_fft_columns(const void *in, void *out, int log2n1, size_t n2, size_t batch, bool inverse, const void *tw_table, hipStream_t st);
int launch_fft_rows_transposed(const void *in, void *out, size_t n1, int log2n2, size_t batch, bool inverse, const void *tw_table, hipStream_t st);
int launch_fft_columns_f64(const void *in, void *out, int log2n1, size_t n2, size_t batch, bool inverse, const void *tw_table, hipStream_t st);
int launch_fft_rows_transposed_f64(const void *in, void *out, size_t n1, int log2n2, size_t batch, bool inverse, const void *tw_table, hipStream_t st);
int launch_fft4096_cf32(const void *in, void *out, size_t nframes, bool inverse, const void *tw4096, void *sched, hipStream_t st);
int launch_fft_pow2_cf32(const void *in, void *out, size_t nbins, size_t nframes, bool inverse, const void *tw,
                         hipStream_t st);
int launch_fft_pow2_cf64(const void *in, void *out, size_t nbins, size_t nframes, bool inverse, const void *tw,
                         hipStream_t st);
// kiss_fft Q15 (bit-exact): nbins = product of radix 4/2 stages; tw = int16 pairs
int launch_fft_q15_global(const void *in, void *out, void *ws, size_t nbins, size_t nframes, bool inverse, const void *tw, const int *radix,
                          int nstages, hipStream_t st);   // (fft_mixed.hip) Q15 frames beyond one workgroup's LDS: one launch per stage
int launch_fft_q15_one(const void *in, void *out, size_t nframes, hipStream_t st);   // numBins = 1, Q15: x * 32767/32768 rounded
int launch_fft_q15(const void *in, void *out, size_t nbins, size_t nframes, bool inverse, const void *tw, const void *perm,
                   const int *radix_host, int nstages, hipStream_t st);

// complex_float32, numBins = 2^log2n in 16..16384 (except 4096): register-resident radix-16 family
size_t fft_r16_table_elems(int log2n);
int launch_fft_r16_cf32(const void *in, void *out, int log2n, size_t nframes, bool inverse, const void *tw, hipStream_t st);
int launch_fft_r16_cf64(const void *in, void *out, int log2n, size_t nframes, bool inverse, const void *tw, hipStream_t st);

// any numBins: kissfft's mixed-radix plan (radix 2/3/4/5 + generic), f32 / f64 / Q15 (bit-exact)
int launch_fft_smooth(int scalar, const void *in, void *out, size_t nbins, size_t nframes, bool inverse, const void *tw, const void *iperm,
                      const int *radix_host, int nstages, hipStream_t st);
int launch_fft_mixed(int scalar, const void *in, void *out, size_t nbins, size_t nframes, bool inverse, const void *tw, const void *iperm,
                     const int *radix_host, int nstages, hipStream_t st);

// Bluestein passes (fft_bluestein.hip): chirp multiply + zero pad, spectrum product, chirp multiply + scale + truncate
int launch_bluestein_pre(int scalar, const void *x, void *y, const void *w, size_t N, size_t M, size_t nframes, bool inverse, hipStream_t st);
int launch_bluestein_mul(int scalar, void *Y, const void *B, size_t M, size_t nframes, hipStream_t st);
int launch_bluestein_post(int scalar, const void *z, void *X, const void *w, size_t N, size_t M, size_t nframes, bool inverse, hipStream_t st);

// fused Rotate -> FIR -> FreqDemod, frequency domain (Hspec already carries the phasor)
int launch_fmchain_cf32_ols4096(const void *in, size_t in_elems, void *out, size_t n_out, const void *Hspec, size_t K,
                                const void *tw4096, const void *prev_in, void *prev_out, void *sched, hipStream_t st,
                                const void *gate_word = nullptr, unsigned gate_value = 0, int *gated = nullptr, unsigned slots = 1024);
int launch_gate_signal(void *gate_word, unsigned value, hipStream_t st);   // (elementwise.hip) one thread: the word <- value, system-scope release
// fused Rotate -> FIR -> FreqDemod, time domain
int launch_fmchain_cf32(const void *in, size_t in_elems, void *out, size_t n_out, const void *taps_rev, size_t K,
                        size_t Kp, const void *prev_in, void *prev_out, hipStream_t st);

}  // namespace pcx
