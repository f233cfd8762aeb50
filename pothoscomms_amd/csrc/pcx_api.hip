// pcx_api.hip -- the extern "C" boundary of libpcx_hip.so (include/pcx.h).
// Host-side only: handle bookkeeping, coefficient tables, algorithm choice, staging
// of host buffers.  All arithmetic on stream data happens in the HIP kernels.
#include <algorithm>
#include <chrono>
#include <complex>
#include <cstdio>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>

#include <dlfcn.h>

#include "pcx_internal.hpp"

namespace pcx {

static thread_local std::string g_err;

void set_error(const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

int64_t float_to_q(double x, int qbits, int frac_bits, int rounding)
{
    double v = std::ldexp(x, frac_bits);
    if (rounding == PCX_Q_NEAREST) v = std::round(v);      // ties away from zero; PCX_Q_TRUNCATE leaves the truncation to the casts below
    if (qbits == 64) {
        if (!(v >= -9223372036854775808.0 && v < 9223372036854775808.0)) return INT64_MIN;
        return (int64_t)v;
    }
    if (qbits == 32) {
        if (!(v >= -2147483648.0 && v < 2147483648.0)) return INT32_MIN;
        return (int32_t)v;
    }
    if (!(v >= -2147483648.0 && v < 2147483648.0)) return 0;
    return (int16_t)(uint16_t)(uint32_t)(int32_t)v;
}

// the process-wide Q-format reading (pcx_set_qformat): three ints, read whole by the control plane of a call
static std::atomic<int> g_qf_frac{kDefaultQFormat.frac}, g_qf_to{kDefaultQFormat.to}, g_qf_from{kDefaultQFormat.from};
QFormat process_qformat() { return QFormat{g_qf_frac.load(), g_qf_to.load(), g_qf_from.load()}; }
static bool qformat_valid(const pcx_qformat &q)
{
    return (q.frac == PCX_Q_FRAC_HALF_Q || q.frac == PCX_Q_FRAC_HALF_ELEM) && (q.float_to_q == PCX_Q_TRUNCATE || q.float_to_q == PCX_Q_NEAREST) &&
           (q.from_q == PCX_Q_FLOOR || q.from_q == PCX_Q_TOWARD_ZERO || q.from_q == PCX_Q_ROUND);
}
int qformat_from_api(const pcx_qformat *q, QFormat *out)
{
    if (!q) { *out = process_qformat(); return PCX_OK; }
    if (!qformat_valid(*q)) {
        set_error("pcx_qformat {frac %d, float_to_q %d, from_q %d}: unknown reading (pcx_q_frac / pcx_q_to / pcx_q_from)", q->frac, q->float_to_q, q->from_q);
        return PCX_ERR_ARG;
    }
    *out = QFormat{q->frac, q->float_to_q, q->from_q};
    return PCX_OK;
}

int DevBuf::ensure(size_t bytes)
{
    if (bytes <= cap) return PCX_OK;
    release();
    size_t want = bytes < 4096 ? 4096 : bytes;
    PCX_HIP(hipMalloc(&p, want));
    cap = want;
    return PCX_OK;
}
// Control-plane transfers (tables at set_taps / create, zeroed state) must be COMPLETE on the device when the call returns: the kernels
// that use them run on non-blocking streams, which wait for nothing.  What proved reliable for that, under every runtime mode tried
// (eight processes on the GPU; AMD_DIRECT_DISPATCH=0; GPU_MAX_HW_QUEUES=1/8; HSA_ENABLE_SDMA=0), is ONE recipe: a page-locked source
// of the library's own, hipMemcpyAsync / a kernel on a non-blocking stream of the library's own, hipStreamSynchronize on that stream.
// What did not: hipMemset (returns before the fill has run) and hipMemcpy from pageable memory followed by a null-stream
// synchronise (under AMD_DIRECT_DISPATCH=0 the first kernel of a new handle still read an empty table five times out of six:
// tools/shard_dd_probe.py; in the default mode about one first call in 10^5 under load).
namespace {
struct ControlLane {           // per host thread: a stream and a staging buffer for the device that is current
    int device = -1;
    hipStream_t st = nullptr;
    PinBuf pin;
    void drop()
    {
        if (st) (void)hipStreamDestroy(st);
        st = nullptr;
        pin.release();
        device = -1;
    }
    ~ControlLane() { drop(); }
};
thread_local ControlLane g_lane;
int control_lane(ControlLane **out)
{
    int cur = -1;
    PCX_HIP(hipGetDevice(&cur));
    if (g_lane.device != cur) {
        if (g_lane.device >= 0) {
            (void)hipSetDevice(g_lane.device);
            g_lane.drop();
            PCX_HIP(hipSetDevice(cur));
        }
        g_lane.device = cur;
    }
    if (!g_lane.st) PCX_HIP(hipStreamCreateWithFlags(&g_lane.st, hipStreamNonBlocking));
    *out = &g_lane;
    return PCX_OK;
}
}  // namespace
int DevBuf::ensure_zeroed(size_t bytes)
{
    PCX_TRY(ensure(bytes));
    ControlLane *lane;
    PCX_TRY(control_lane(&lane));
    PCX_TRY(launch_zero_words(p, (bytes + 3) / 4, lane->st));     // (ensure() rounds the allocation up to 4 KiB: whole words exist)
    PCX_HIP(hipStreamSynchronize(lane->st));
    return PCX_OK;
}
void DevBuf::release()
{
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}

// upload a host vector into a DevBuf (control plane: COMPLETE on return; the recipe above)
template <typename T>
static int upload(DevBuf &b, const std::vector<T> &v)
{
    const size_t bytes = v.size() * sizeof(T);
    PCX_TRY(b.ensure(bytes));
    if (bytes == 0) return PCX_OK;
    ControlLane *lane;
    PCX_TRY(control_lane(&lane));
    PCX_TRY(lane->pin.ensure(bytes));
    std::memcpy(lane->pin.p, v.data(), bytes);
    PCX_HIP(hipMemcpyAsync(b.p, lane->pin.p, bytes, hipMemcpyHostToDevice, lane->st));
    PCX_HIP(hipStreamSynchronize(lane->st));
    return PCX_OK;
}

// lane-constant twiddle table of the radix-16 x3 4096-point transform (fft4096.hpp):
//   p = 0..14:  p < 3 -> (w^4)^(p+1);   p = 3 + (n2-1)*4 + k1 -> w^n2 * W16^(n2 k1)
//   tab[p * 16 + kk]        with w = exp(-j 2 pi kk / 256)    (pass 2, 240 entries)
//   tab[240 + p * 256 + j]  with w = exp(-j 2 pi j / 4096)    (pass 3, 3840 entries)
static std::vector<float> make_tw4096()
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
static std::vector<T> make_tw_r16(int log2n)
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
static std::vector<double> make_tw_ols64(int log2n)
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
static std::vector<T> make_hspec(const std::vector<std::complex<double>> &h, size_t N, size_t advance = 0)
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
static std::vector<float> make_hspec4096(const std::vector<std::complex<double>> &h) { return make_hspec(h, 4096); }
// The resampling kernels (fir_ols_decim.hip) re-read H from L2 in every block, so their copy is stored the way their lanes hold the
// spectrum (fft4096.hpp, spec_lane): entry j + 256 r is bin (j >> 4) + 16 (j & 15) + 256 r, and a wave still reads whole 512-byte rows
static std::vector<float> turn_spectrum_lanes(const std::vector<float> &H)
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
static std::vector<float> make_hparts(const std::vector<std::complex<double>> &h, int parts)
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

using namespace pcx;

#define PCX_CHECK_ARG(cond, ...)        \
    do {                                \
        if (!(cond)) {                  \
            set_error(__VA_ARGS__);     \
            return PCX_ERR_ARG;         \
        }                               \
    } while (0)

// every function below is declared extern "C" in pcx.h and keeps that linkage

const char *pcx_last_error(void) { return g_err.c_str(); }
const char *pcx_version(void) { return "pothoscomms_amd 0.1 (gfx950)"; }

// ROCTx ranges (include/pcx.h pcx_trace): the library is loaded on request only, nothing links against it
namespace pcx {
std::atomic<int> g_trace_on{0};
int (*g_roctx_push)(const char *) = nullptr;
int (*g_roctx_pop)() = nullptr;
}  // namespace pcx
int pcx_trace(int on)
{
    if (!on) { g_trace_on.store(0); return PCX_OK; }
    if (!g_roctx_push) {
        void *lib = nullptr;
        for (const char *name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"})
            if ((lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!lib) { set_error("pcx_trace: no ROCTx library found (librocprofiler-sdk-roctx.so.1, libroctx64.so.4)"); return PCX_ERR_UNSUPPORTED; }
        auto push = reinterpret_cast<int (*)(const char *)>(dlsym(lib, "roctxRangePushA"));
        auto pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
        if (!push || !pop) { set_error("pcx_trace: roctxRangePushA / roctxRangePop not exported"); return PCX_ERR_UNSUPPORTED; }
        g_roctx_pop = pop;
        g_roctx_push = push;
    }
    g_trace_on.store(1);
    return PCX_OK;
}

int pcx_device_count(int *count)
{
    PCX_CHECK_ARG(count, "null count");
    PCX_HIP(hipGetDeviceCount(count));
    return PCX_OK;
}
int pcx_set_device(int ordinal) { PCX_HIP(hipSetDevice(ordinal)); return PCX_OK; }
int pcx_get_device(int *ordinal)
{
    PCX_CHECK_ARG(ordinal, "null ordinal");
    PCX_HIP(hipGetDevice(ordinal));
    return PCX_OK;
}
int pcx_dev_alloc(void **dptr, size_t bytes)
{
    PCX_CHECK_ARG(dptr, "null dptr");
    PCX_HIP(hipMalloc(dptr, bytes ? bytes : 1));
    return PCX_OK;
}
int pcx_dev_free(void *dptr) { PCX_HIP(hipFree(dptr)); return PCX_OK; }
int pcx_memcpy_h2d(void *d, const void *s, size_t n, void *st) { PCX_HIP(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, as_stream(st))); return PCX_OK; }
int pcx_memcpy_d2h(void *d, const void *s, size_t n, void *st) { PCX_HIP(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, as_stream(st))); return PCX_OK; }
int pcx_memcpy_d2d(void *d, const void *s, size_t n, void *st) { PCX_HIP(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, as_stream(st))); return PCX_OK; }
int pcx_host_alloc(void **hptr, size_t bytes)
{
    PCX_CHECK_ARG(hptr, "null hptr");
    PCX_HIP(hipHostMalloc(hptr, bytes ? bytes : 1, hipHostMallocPortable | hipHostMallocMapped));   // visible to every device of the process
    return PCX_OK;
}
int pcx_host_free(void *hptr) { PCX_HIP(hipHostFree(hptr)); return PCX_OK; }

// ---- page-locking memory the framework owns (include/pcx.h) ----
namespace {
std::mutex g_reg_mutex;
// the ranges THIS library page-locked: base -> bytes, how many callers hold it, and -- for a range found through
// pcx_host_register_mapping -- which object was mapped there (device + inode of the shared file), so that a caller can ask later
// whether the registration still describes what is mapped at that address
struct Registration { size_t bytes; unsigned holders; unsigned long long inode; std::string dev; };
std::map<uintptr_t, Registration> g_registered;
struct Vma { uintptr_t lo, hi; bool rw, shared; unsigned long long inode; std::string dev; };
// the mappings of this process, ascending (/proc/self/maps: "lo-hi perms offset dev inode path")
std::vector<Vma> read_maps()
{
    std::vector<Vma> v;
    FILE *f = std::fopen("/proc/self/maps", "r");
    if (!f) return v;
    char line[1024];
    while (std::fgets(line, sizeof line, f)) {
        unsigned long long lo, hi, off, ino;
        char perms[8] = {0}, dev[16] = {0};
        if (std::sscanf(line, "%llx-%llx %7s %llx %15s %llu", &lo, &hi, perms, &off, dev, &ino) != 6) continue;
        v.push_back({(uintptr_t)lo, (uintptr_t)hi, perms[0] == 'r' && perms[1] == 'w', perms[3] == 's', ino, dev});
    }
    std::fclose(f);
    return v;
}
// is [lo, hi) still covered, without a gap, by shared read-write mappings of the object (dev, inode)?
bool still_mapped(const std::vector<Vma> &maps, uintptr_t lo, uintptr_t hi, unsigned long long inode, const std::string &dev)
{
    uintptr_t at = lo;
    for (const Vma &m : maps) {
        if (m.hi <= at) continue;
        if (m.lo > at) return false;
        if (!(m.shared && m.rw && m.inode == inode && m.dev == dev)) return false;
        at = m.hi;
        if (at >= hi) return true;
    }
    return false;
}
}  // namespace
int pcx_host_register(void *ptr, size_t bytes)
{
    PCX_CHECK_ARG(ptr && bytes, "pcx_host_register: empty range");
    {
        std::lock_guard<std::mutex> lk(g_reg_mutex);
        auto it = g_registered.find((uintptr_t)ptr);
        if (it != g_registered.end() && it->second.bytes >= bytes) { it->second.holders++; return PCX_OK; }      // held already: one more holder
    }
    int kind = PCX_PTR_PAGEABLE;
    PCX_TRY(pcx_pointer_kind(ptr, &kind));
    if (kind == PCX_PTR_PAGE_LOCKED) return PCX_OK;
    PCX_CHECK_ARG(kind == PCX_PTR_PAGEABLE, "pcx_host_register: %p is device memory", ptr);
    const hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterPortable | hipHostRegisterMapped);
    if (e == hipErrorHostMemoryAlreadyRegistered) { (void)hipGetLastError(); return PCX_OK; }
    PCX_HIP(e);
    std::lock_guard<std::mutex> lk(g_reg_mutex);
    g_registered[(uintptr_t)ptr] = Registration{bytes, 1u, 0ull, std::string()};
    return PCX_OK;
}
int pcx_host_unregister(void *ptr)
{
    {
        std::lock_guard<std::mutex> lk(g_reg_mutex);
        auto it = g_registered.find((uintptr_t)ptr);
        PCX_CHECK_ARG(it != g_registered.end(), "pcx_host_unregister: %p is not the base of a range this library page-locked", ptr);
        if (--it->second.holders > 0) return PCX_OK;         // another block still runs in place on it
        g_registered.erase(it);
    }
    PCX_HIP(hipHostUnregister(ptr));
    return PCX_OK;
}
int pcx_host_register_mapping(const void *p, size_t bytes, size_t max_bytes, void **base, size_t *len)
{
    PCX_CHECK_ARG(p && bytes && base && len, "pcx_host_register_mapping: null argument");
    *base = nullptr; *len = 0;
    const uintptr_t lo = (uintptr_t)p, hi = lo + bytes;
    {
        // a range this library locked already (for another block, or for this one under another window): one more holder
        std::lock_guard<std::mutex> lk(g_reg_mutex);
        auto it = g_registered.upper_bound(lo);
        if (it != g_registered.begin()) {
            --it;
            if (it->first <= lo && hi <= it->first + it->second.bytes) {
                it->second.holders++;
                *base = (void *)it->first; *len = it->second.bytes;
                return PCX_OK;
            }
        }
    }
    int kind = PCX_PTR_PAGEABLE;
    PCX_TRY(pcx_pointer_kind(p, &kind));
    if (kind != PCX_PTR_PAGEABLE) return PCX_OK;            // page-locked by somebody else (or device memory): nothing to do
    const std::vector<Vma> maps = read_maps();
    size_t first = maps.size();
    for (size_t i = 0; i < maps.size(); i++)
        if (maps[i].lo <= lo && lo < maps[i].hi) { first = i; break; }
    if (first == maps.size() || !maps[first].shared || !maps[first].rw || maps[first].inode == 0) return PCX_OK;   // not a shared file object
    const Vma m = maps[first];
    auto same = [&](const Vma &o) { return o.shared && o.rw && o.inode == m.inode && o.dev == m.dev; };
    // [p, p + bytes) must lie in consecutive mappings of that one object ...
    size_t last = first;
    while (maps[last].hi < hi) {
        if (last + 1 >= maps.size() || maps[last + 1].lo != maps[last].hi || !same(maps[last + 1])) return PCX_OK;
        last++;
    }
    // ... and every adjacent mapping of it comes along: the other half of a double mapping, whichever side this window is on
    while (first > 0 && maps[first - 1].hi == maps[first].lo && same(maps[first - 1])) first--;
    while (last + 1 < maps.size() && maps[last + 1].lo == maps[last].hi && same(maps[last + 1])) last++;
    const uintptr_t rlo = maps[first].lo, rhi = maps[last].hi;
    if (rhi - rlo > (max_bytes ? max_bytes : ((size_t)1 << 30))) return PCX_OK;
    const hipError_t e = hipHostRegister((void *)rlo, rhi - rlo, hipHostRegisterPortable | hipHostRegisterMapped);
    if (e == hipErrorHostMemoryAlreadyRegistered) { (void)hipGetLastError(); return PCX_OK; }
    PCX_HIP(e);
    {
        std::lock_guard<std::mutex> lk(g_reg_mutex);
        g_registered[rlo] = Registration{rhi - rlo, 1u, m.inode, m.dev};
    }
    *base = (void *)rlo; *len = rhi - rlo;
    return PCX_OK;
}
int pcx_host_mapping_alive(const void *base, int *alive)
{
    PCX_CHECK_ARG(base && alive, "pcx_host_mapping_alive: null argument");
    Registration r;
    {
        std::lock_guard<std::mutex> lk(g_reg_mutex);
        auto it = g_registered.find((uintptr_t)base);
        PCX_CHECK_ARG(it != g_registered.end(), "pcx_host_mapping_alive: %p is not the base of a range this library page-locked", base);
        r = it->second;
    }
    // a range registered by address alone (pcx_host_register) has no identity to compare: the caller vouches for it
    *alive = r.inode == 0 ? 1 : (still_mapped(read_maps(), (uintptr_t)base, (uintptr_t)base + r.bytes, r.inode, r.dev) ? 1 : 0);
    return PCX_OK;
}
int pcx_host_release_range(const void *p, size_t bytes)
{
    PCX_CHECK_ARG(p && bytes, "pcx_host_release_range: empty range");
    std::vector<uintptr_t> gone;
    {
        std::lock_guard<std::mutex> lk(g_reg_mutex);
        const uintptr_t lo = (uintptr_t)p, hi = lo + bytes;
        for (auto it = g_registered.begin(); it != g_registered.end();) {
            if (it->first < hi && lo < it->first + it->second.bytes) { gone.push_back(it->first); it = g_registered.erase(it); }
            else ++it;
        }
    }
    // (the mapping may be gone already, or partly: the runtime's complaint about that is not the caller's problem)
    for (uintptr_t b : gone) if (hipHostUnregister((void *)b) != hipSuccess) (void)hipGetLastError();
    return PCX_OK;
}
int pcx_pointer_kind(const void *p, int *kind)
{
    PCX_CHECK_ARG(kind, "null kind");
    *kind = PCX_PTR_PAGEABLE;
    if (!p) return PCX_OK;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return PCX_OK; }   // unknown to the runtime: pageable
    if (a.type == hipMemoryTypeDevice) *kind = PCX_PTR_DEVICE;
    else if (a.type == hipMemoryTypeHost || a.type == hipMemoryTypeManaged) *kind = PCX_PTR_PAGE_LOCKED;
    return PCX_OK;
}
int pcx_memcpy_to_host(void *dst_host, const void *src, size_t bytes)
{
    PCX_CHECK_ARG(dst_host && src, "null buffer");
    int kind = PCX_PTR_PAGEABLE;
    PCX_TRY(pcx_pointer_kind(src, &kind));
    if (kind != PCX_PTR_DEVICE) { std::memcpy(dst_host, src, bytes); return PCX_OK; }
    // device memory: the CPU must not touch it.  A blocking copy, and the null stream drained behind it (profiles/r02/contention.md:
    // a blocking copy is not complete on return for the purposes of a non-blocking stream)
    PCX_HIP(hipMemcpy(dst_host, src, bytes, hipMemcpyDeviceToHost));
    PCX_HIP(hipStreamSynchronize(nullptr));
    return PCX_OK;
}
int pcx_stream_sync(void *st) { PCX_HIP(hipStreamSynchronize(as_stream(st))); return PCX_OK; }
int pcx_pcie_probe(size_t bytes, int reps, double *h2d_gbs, double *d2h_gbs, double *both_gbs)
{
    PCX_CHECK_ARG(bytes >= 4096 && reps >= 1 && h2d_gbs && d2h_gbs && both_gbs, "pcx_pcie_probe: bad argument");
    void *hin = nullptr, *hout = nullptr, *din = nullptr, *dout = nullptr;
    hipStream_t s0 = nullptr, s1 = nullptr;
    int rc = PCX_OK;
    auto fail = [&](hipError_t e) { if (e != hipSuccess && rc == PCX_OK) { set_error("pcx_pcie_probe: %s", hipGetErrorString(e)); rc = PCX_ERR_HIP; } return e != hipSuccess; };
    if (!fail(hipHostMalloc(&hin, bytes, hipHostMallocDefault)) && !fail(hipHostMalloc(&hout, bytes, hipHostMallocDefault)) &&
        !fail(hipMalloc(&din, bytes)) && !fail(hipMalloc(&dout, bytes)) && !fail(hipMemset(dout, 0, bytes)) &&
        !fail(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)) && !fail(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking))) {
        std::memset(hin, 1, bytes);
        std::memset(hout, 0, bytes);
        // `reps` transfers per direction queued back to back, one synchronisation behind them (the steady state of a stream of calls;
        // tools/pcie_lab.hip times the same way); a warm-up transfer first
        auto timed = [&](bool up, bool down) -> double {
            double dt = 0.0;
            for (int round = 0; round < 2 && rc == PCX_OK; round++) {
                const int n = round == 0 ? 1 : reps;
                const auto t0 = std::chrono::steady_clock::now();
                for (int r = 0; r < n; r++) {
                    if (up) fail(hipMemcpyAsync(din, hin, bytes, hipMemcpyHostToDevice, s0));
                    if (down) fail(hipMemcpyAsync(hout, dout, bytes, hipMemcpyDeviceToHost, s1));
                }
                fail(hipStreamSynchronize(s0));
                fail(hipStreamSynchronize(s1));
                dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / n;
            }
            return dt > 0.0 ? (double)bytes / dt / 1e9 : 0.0;
        };
        *h2d_gbs = timed(true, false);
        *d2h_gbs = timed(false, true);
        *both_gbs = timed(true, true);
    }
    if (s0) (void)hipStreamDestroy(s0);
    if (s1) (void)hipStreamDestroy(s1);
    if (din) (void)hipFree(din);
    if (dout) (void)hipFree(dout);
    if (hin) (void)hipHostFree(hin);
    if (hout) (void)hipHostFree(hout);
    return rc;
}
int pcx_fill_uniform_f32_dev(float *dst, size_t n, uint64_t seed, uint64_t offset, void *st)
{
    return launch_fill_uniform_f32(dst, n, seed, offset, as_stream(st));
}
int pcx_set_qformat(const pcx_qformat *q)
{
    QFormat f = kDefaultQFormat;
    if (q) PCX_TRY(qformat_from_api(q, &f));
    g_qf_frac.store(f.frac); g_qf_to.store(f.to); g_qf_from.store(f.from);
    return PCX_OK;
}
int pcx_get_qformat(pcx_qformat *q)
{
    PCX_CHECK_ARG(q, "null output");
    const QFormat f = process_qformat();
    q->frac = f.frac; q->float_to_q = f.to; q->from_q = f.from;
    return PCX_OK;
}
int pcx_clock_probe_dev(float *mhz_dev, unsigned spin_us, void *st)
{
    PCX_CHECK_ARG(mhz_dev, "null output");
    PCX_CHECK_ARG(spin_us >= 1 && spin_us <= 100000, "pcx_clock_probe_dev: spin of %u us (1 .. 100000)", spin_us);
    return launch_clock_probe(mhz_dev, spin_us, as_stream(st));
}


// A handle belongs to ONE device: the one current on the calling thread at the first call that
// touches the device.  Later calls (any thread -- Pothos runs every block on its own) switch to
// it for the duration of the call and restore the caller's device afterwards.
struct DeviceScope {
    int prev = -1; bool switched = false;
    explicit DeviceScope(int &bound)
    {
        if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; return; }
        if (bound < 0) { bound = prev; return; }
        if (bound != prev) switched = (hipSetDevice(bound) == hipSuccess);
    }
    ~DeviceScope() { if (switched) (void)hipSetDevice(prev); }
};
// Execution context of a handle (one per block instance, include/pcx.h "Conventions"):
//   device  bound in *_create (the creating thread's current device) -- or, when no device was reachable then, at
//           the first device-touching call;
//   own     the handle's own non-blocking stream: staging copies and kernels of the HOST-pointer entry points, so two
//           blocks on two Pothos actor threads overlap instead of serialising on the legacy default stream;
//   last    the stream of the handle's most recent enqueue.  A call that arrives on a different stream is ordered
//           behind it with an event (carried state such as FreqDemod's prev, and the tables, are read by kernels);
//           control-plane rewrites of device tables first wait for it (ctx_quiesce).
struct ExecCtx {
    int device = -1;
    hipStream_t own = nullptr;
    hipStream_t last = nullptr;
    bool have_last = false;
    hipEvent_t ev = nullptr;
    // the DRAINED output of a host-pointer call (drain_* below): a second stream whose copy engine moves finished chunks of the
    // result from a device workspace into the caller's page-locked buffer while the kernels are still reading the input over PCIe
    static constexpr int kDrainChunks = 8;
    hipStream_t drain = nullptr;
    hipEvent_t drain_ev[kDrainChunks] = {};
    ExecCtx() = default;
    ExecCtx(const ExecCtx &) = delete;
    ExecCtx &operator=(const ExecCtx &) = delete;
    ~ExecCtx()
    {
        if (ev) (void)hipEventDestroy(ev);
        for (hipEvent_t e : drain_ev) if (e) (void)hipEventDestroy(e);
        if (drain) (void)hipStreamDestroy(drain);
        if (own) (void)hipStreamDestroy(own);
    }
};
// (call with the handle's DeviceScope alive)
static int ctx_own_stream(ExecCtx &c, hipStream_t *out)
{
    if (!c.own) PCX_HIP(hipStreamCreateWithFlags(&c.own, hipStreamNonBlocking));
    *out = c.own;
    return PCX_OK;
}
// before enqueuing on `st`: order it behind the handle's previous enqueue if that went to another stream
static int ctx_enter(ExecCtx &c, hipStream_t st)
{
    if (c.have_last && c.last != st) {
        if (!c.ev) PCX_HIP(hipEventCreateWithFlags(&c.ev, hipEventDisableTiming));
        PCX_HIP(hipEventRecord(c.ev, c.last));
        PCX_HIP(hipStreamWaitEvent(st, c.ev, 0));
    }
    c.last = st;
    c.have_last = true;
    return PCX_OK;
}
// control plane: nothing the handle enqueued may still be reading the tables about to be rewritten
static int ctx_quiesce(ExecCtx &c)
{
    if (c.have_last) PCX_HIP(hipStreamSynchronize(c.last));
    return PCX_OK;
}

// Device-visible alias of a HOST pointer when it is page-locked (pcx_host_alloc / hipHostMalloc / hipHostRegister; any
// offset inside the allocation), else nullptr.  The host-pointer entry points launch their kernels straight on such
// buffers -- measured on MI355X (tools/pcie_lab.hip, 128 MiB each way): a kernel reading and writing pinned host memory
// moves 43 GB/s in BOTH directions at once, against 28 GB/s for H2D, kernel, D2H through a staging workspace -- and stage
// only pageable memory, which the device cannot address.
namespace pcx {
void *device_alias(const void *p)
{
    if (!p) return nullptr;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if ((a.type == hipMemoryTypeHost || a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged) && a.devicePointer) return a.devicePointer;
    return nullptr;
}
}  // namespace pcx
// One direction of a host-pointer call: the alias when there is one, else the staging pair (grown to `bytes`): a device
// buffer and a page-locked bounce buffer of the library's own.  Pageable memory is copied by the CPU into / out of the
// bounce buffer and moved by plain pinned <-> device transfers on the call's stream.  (hipMemcpyAsync straight on the
// caller's pageable pointer was the first implementation; under eight processes sharing the GPU about one call in 10^5
// came back with the head and tail of its output never written -- profiles/r02/contention.md.)
int PinBuf::ensure(size_t bytes)
{
    if (bytes <= cap) return PCX_OK;
    release();
    size_t want = bytes < 65536 ? 65536 : bytes + bytes / 4;
    PCX_HIP(hipHostMalloc(&p, want, hipHostMallocDefault));
    cap = want;
    return PCX_OK;
}
void PinBuf::release()
{
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
}
// Transfers above a couple of MiB go in pieces so that the CPU's copy of one piece runs while the DMA engine moves the
// previous one (measured, 128 MiB each way: 23.1 ms in one piece; tools/host_path.py)
static size_t stage_piece(size_t bytes)
{
    constexpr size_t kMin = (size_t)1 << 20, kMaxPieces = 16;
    size_t piece = (bytes + kMaxPieces - 1) / kMaxPieces;
    piece = (piece + 4095) & ~(size_t)4095;
    return piece < kMin ? kMin : piece;
}
// The CPU side of a large staged transfer: one core copies 10-14 GB/s, the DMA engine moves 50.  Pieces of 4 MiB and more
// are split over up to four short-lived helper threads (the caller copies the first share itself); smaller copies, i.e.
// every call below 64 MiB, stay on the calling thread.
static void stage_copy(void *dst, const void *src, size_t bytes)
{
    constexpr size_t kParallelFrom = (size_t)4 << 20;
    unsigned hw = std::thread::hardware_concurrency();
    const unsigned nt = bytes >= kParallelFrom ? std::min(4u, hw > 1 ? hw / 2 : 1u) : 1u;
    if (nt <= 1) { std::memcpy(dst, src, bytes); return; }
    const size_t share = ((bytes / nt) + 4095) & ~(size_t)4095;
    std::thread helpers[3];
    unsigned started = 0;
    for (unsigned t = 1; t < nt; t++) {
        const size_t off = (size_t)t * share;
        if (off >= bytes) break;
        const size_t c = bytes - off < share ? bytes - off : share;
        try {
            helpers[started] = std::thread([=] { std::memcpy(static_cast<char *>(dst) + off, static_cast<const char *>(src) + off, c); });
            started++;
        } catch (...) {      // no thread to be had: the caller copies this share as well
            std::memcpy(static_cast<char *>(dst) + off, static_cast<const char *>(src) + off, c);
        }
    }
    std::memcpy(dst, src, share < bytes ? share : bytes);
    for (unsigned t = 0; t < started; t++) helpers[t].join();
}
void StageBuf::release()
{
    dev.release();
    pin.release();
}
// Every staging buffer of a call is in place BEFORE its first transfer is queued (no allocation between queuing a transfer and its
// completion: profiles/r02/contention.md section 4).  The host entry points reserve all their directions first.
static int stage_reserve(const void *host, size_t bytes, StageBuf &ws)
{
    if (device_alias(host)) return PCX_OK;
    PCX_TRY(ws.dev.ensure(bytes));
    PCX_TRY(ws.pin.ensure(bytes));
    return PCX_OK;
}
static int stage_in(const void *host, size_t bytes, StageBuf &ws, hipStream_t st, const void **dev)
{
    if (void *a = device_alias(host)) { *dev = a; return PCX_OK; }
    PCX_TRY(ws.dev.ensure(bytes));
    PCX_TRY(ws.pin.ensure(bytes));
    const size_t piece = stage_piece(bytes);
    for (size_t off = 0; off < bytes; off += piece) {
        const size_t c = bytes - off < piece ? bytes - off : piece;
        stage_copy(static_cast<char *>(ws.pin.p) + off, static_cast<const char *>(host) + off, c);
        PCX_HIP(hipMemcpyAsync(static_cast<char *>(ws.dev.p) + off, static_cast<const char *>(ws.pin.p) + off, c, hipMemcpyHostToDevice, st));
    }
    *dev = ws.dev.p;
    return PCX_OK;
}
static int stage_out_begin(void *host, size_t bytes, StageBuf &ws, void **dev, bool *staged)
{
    if (void *a = device_alias(host)) { *dev = a; *staged = false; return PCX_OK; }
    PCX_TRY(ws.dev.ensure(bytes));
    PCX_TRY(ws.pin.ensure(bytes));
    *dev = ws.dev.p; *staged = true;
    return PCX_OK;
}
// behind the kernels of the call: device -> bounce buffer -> the caller's memory.  Piece i+1 is on its way while the CPU
// copies piece i out; the only completion primitive used is hipStreamSynchronize (a variant with one event per piece
// produced a wrong call in the soak)
static int stage_out_first(StageBuf &ws, size_t bytes, bool staged, hipStream_t st)
{
    if (!staged || !bytes) return PCX_OK;
    const size_t piece = stage_piece(bytes), c = bytes < piece ? bytes : piece;
    PCX_HIP(hipMemcpyAsync(ws.pin.p, ws.dev.p, c, hipMemcpyDeviceToHost, st));
    return PCX_OK;
}
static int stage_out_rest(void *host, StageBuf &ws, size_t bytes, bool staged, hipStream_t st)
{
    if (!staged || !bytes) return PCX_OK;
    const size_t piece = stage_piece(bytes);
    for (size_t off = 0; off < bytes; off += piece) {
        const size_t c = bytes - off < piece ? bytes - off : piece;
        PCX_HIP(hipStreamSynchronize(st));          // piece at `off` has landed in the bounce buffer
        const size_t nxt = off + piece;
        if (nxt < bytes) {
            const size_t cn = bytes - nxt < piece ? bytes - nxt : piece;
            PCX_HIP(hipMemcpyAsync(static_cast<char *>(ws.pin.p) + nxt, static_cast<const char *>(ws.dev.p) + nxt, cn, hipMemcpyDeviceToHost, st));
        }
        stage_copy(static_cast<char *>(host) + off, static_cast<const char *>(ws.pin.p) + off, c);
    }
    return PCX_OK;
}
static int stage_out_end(void *host, size_t bytes, StageBuf &ws, bool staged, hipStream_t st)
{
    PCX_TRY(stage_out_first(ws, bytes, staged, st));
    PCX_TRY(stage_out_rest(host, ws, bytes, staged, st));
    PCX_HIP(hipStreamSynchronize(st));      // (a call that ran in place on page-locked buffers ends here)
    return PCX_OK;
}

// ---- the DRAINED output direction (diagnostic library only: measured, NOT adopted) -------------------------------------------------
// Idea: a kernel that reads page-locked host memory AND writes page-locked host memory moves 43 GB/s each way; a kernel
// that reads it and writes DEVICE memory reads at 55, and a copy engine drains device memory to the host at 57 (each alone).  So a
// host-pointer call would go in chunks: chunk c's kernel writes a device workspace, and behind an event the copy engine of a second
// stream moves that chunk out while chunk c+1's kernel reads its input.
// Measured (tools/pcie_lab.hip, tools/drain_ab.py, profiles/r05/pcie_lab.txt, drain_ab.txt): the two do NOT overlap on this platform.
// "kernel pinned->device || D2H copy" runs at 29.7 GB/s per direction -- the sum of the two times -- and the whole FIR call at 27.8
// against 40.0 in place (16 Mi samples); copy engines on BOTH sides reach 48.4 unchunked but 26-41 in chunks of 1-8 MiB (15-20 us per
// queued copy), so no chunked form beats the in-place kernel (43.3) below calls of ~100 MiB.  The in-place form stays the product's;
// this form stays reachable in libpcx_hip_diag.so (PCX_DRAIN_FROM = bytes of output from which a call is drained, PCX_DRAIN_CHUNK)
// so that the finding can be re-measured, and its chunks are bit-identical to the uncut call (tests/test_hostpath_gpu.py).
static size_t drain_from() { return (size_t)PCX_ENV_INT("PCX_DRAIN_FROM", (long)1 << 62); }     // product: never
static size_t drain_chunk_bytes() { return (size_t)PCX_ENV_INT("PCX_DRAIN_CHUNK", 2 << 20); }
// A call whose input or output is HOST memory the kernel addresses over PCIe is bound by the link, not by the device -- and the link is
// full duplex.  On the device-resident grid (1024 persistent workgroups, one or a few blocks each) a call of a Pothos slab's size
// is a few hundred blocks that all load, then all compute, then all store: reads and writes never overlap (1 Mi samples: 0.281 ms).
// On ~48 workgroups that walk several blocks each on the grid stride, block k+1's fetch (issued at the foot of block k) runs beside block
// k's stores: 0.255 ms at 1 Mi samples, 0.833 against 0.947 at 4 Mi, 3.08 against 3.35 at 16 Mi = 43.6 GB/s each way, the rate
// tools/pcie_lab.hip measures for a plain copy kernel on the same buffers (profiles/r05/host_slots_static.txt; 32 and 64 are within 2 %).
static unsigned host_grid() { return (unsigned)PCX_ENV_INT("PCX_HOST_GRID", 48); }
// the grid-stride map kernels (pcx_internal.hpp LINK-BOUND LAUNCHES): 32 blocks for the one-to-one maps, 64 for /comms/freq_demod (two reads per sample)
static unsigned host_map_grid(unsigned dflt = 32) { const unsigned e = (unsigned)PCX_ENV_INT("PCX_HOST_MAP_GRID", -1); return e == (unsigned)-1 ? dflt : e; }
namespace pcx { thread_local unsigned g_link_grid = 0, g_link_map_grid = 0; }
// page-locked HOST memory, reached over the link: what the link-bound launch shape is for.  Managed memory is addressed in place as
// well (pcx_pointer_kind files it under page-locked), but it may be resident in HBM: it keeps the device-resident grid.
static bool host_page_locked(const void *p)
{
    if (!p) return false;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}
// (grid, map grid) for a launch whose input or output is page-locked host memory the kernel addresses in place, else (0, 0)
struct LinkBound : LinkBoundScope {
    static bool any(const void *a, const void *b, const void *c) { return host_page_locked(a) || host_page_locked(b) || (c && host_page_locked(c)); }
    LinkBound(const void *a, const void *b, const void *c = nullptr, unsigned map_blocks = 32) : LinkBound(any(a, b, c), map_blocks) {}

private:
    // (the pointers are looked up ONCE per call: each lookup is a hipPointerGetAttributes)
    LinkBound(bool link, unsigned map_blocks) : LinkBoundScope(link ? host_grid() : 0, link ? host_map_grid(map_blocks) : 0) {}
};
// how many chunks a drained output of `bytes` goes in (2 .. kDrainChunks)
static int drain_chunks(size_t bytes)
{
    const size_t c = drain_chunk_bytes();
    size_t n = (bytes + c - 1) / (c ? c : 1);
    if (n < 2) n = 2;
    if (n > (size_t)ExecCtx::kDrainChunks) n = ExecCtx::kDrainChunks;
    return (int)n;
}
// (everything is created BEFORE the first transfer of a call is queued: profiles/r02/contention.md section 4)
static int drain_setup(ExecCtx &c, int nchunks)
{
    if (!c.drain) PCX_HIP(hipStreamCreateWithFlags(&c.drain, hipStreamNonBlocking));
    for (int i = 0; i < nchunks; i++)
        if (!c.drain_ev[i]) PCX_HIP(hipEventCreateWithFlags(&c.drain_ev[i], hipEventDisableTiming));
    return PCX_OK;
}
// chunk i's kernels are queued on `compute`: its bytes leave for the caller's (page-locked) buffer behind them
static int drain_chunk(ExecCtx &c, int i, hipStream_t compute, void *host_dst, const void *dev_src, size_t bytes)
{
    if (!bytes) return PCX_OK;
    PCX_HIP(hipEventRecord(c.drain_ev[i], compute));
    PCX_HIP(hipStreamWaitEvent(c.drain, c.drain_ev[i], 0));
    PCX_HIP(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, c.drain));
    return PCX_OK;
}
// the result is in the caller's buffer on return (the drain stream is behind every chunk's kernels)
static int drain_finish(ExecCtx &c, hipStream_t compute)
{
    PCX_HIP(hipStreamSynchronize(c.drain));
    PCX_HIP(hipStreamSynchronize(compute));
    return PCX_OK;
}

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
    bool have_ols64 = false;      // complex_float64 stream, M=L=1 (Hspec / tw4096 then hold doubles)
    bool have_ols_int = false;    // complex_int16 / complex_int8 stream, M=L=1: exact integer convolution on the double transform
    bool have_ols_real64 = false; // REAL float64 / int16 / int8 stream (real taps), M=L=1: two real blocks per double transform
    bool have_interp64 = false;   // complex_float64 / int16 / int8, M = 1, L > 1: polyphase rows on the double pipeline (HrowsD) + interleave
    bool have_interp_real = false; // REAL float64 / float32 / int16 / int8, L > 1: the same with the two-real-blocks kernel
    DevBuf HrowsD;
    DevBuf HspecRows;
    int ols_parts = 0;        // complex_float32 M = L = 1: 0 = fir_ols.hip's 4096 kernel alone, 2 .. 4 = that many tap partitions (fir_ols_part.hip)
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
    if ((h->scalar == PCX_F64 || h->scalar == PCX_F32 || h->scalar == PCX_I16 || h->scalar == PCX_I8) && !h->cplx && h->M <= 65535 && h->L > 1 &&
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
    h->have_ols_real64 = false;
    // (real float32 joins for decimating filters only: its undecimated stream has the float kernel below)
    if ((h->scalar == PCX_F64 || h->scalar == PCX_I16 || h->scalar == PCX_I8 || (h->scalar == PCX_F32 && h->M > 1)) && !h->cplx && h->M <= 65535 &&
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
    if (h->scalar == PCX_F32 && !h->cplx && h->M == 1 && h->L == 1 && h->K <= 2049) {
        std::vector<std::complex<double>> hq(h->K);
        for (size_t k = 0; k < h->K; k++) hq[k] = std::complex<double>((double)(float)h->taps[k], 0.0);
        PCX_TRY(upload(h->Hspec, make_hspec4096(hq)));
        PCX_TRY(upload(h->tw4096, make_tw4096()));
        h->have_real_ols = true;
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
// their outputs stay within the 1e-5 of the oracle either way, but are not bit-identical to an uncut call's.
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
        else if ((fast && h->have_ols) || h->have_poly || (h->have_real_ols && h->K > 1) ||
                 (h->have_ols64 && h->K >= (h->M > 1 ? 16 : kOls64MinTaps)) ||
                 (h->have_ols_int && h->K >= (h->M > 1 ? 32 : ols_int_min_taps(h->scalar))) ||
                 (h->have_ols_real64 && h->K >= (h->M > 1 ? 16 : ols_real64_min_taps(h->scalar))) ||
                 ((h->have_interp64 || h->have_interp_real) && h->K >= 16)) algo = PCX_FIR_OLS_FFT;
        // longer than every frequency-domain plan (K > 8193): the sliding-window kernel in the reference's own
        // operation order -- 8k-term float sums accumulate enough rounding that a reordered sum would sit on the 1e-5 bar
        else if (fast) algo = h->K > kOlsMaxTaps ? PCX_FIR_EXACT : PCX_FIR_DIRECT;
        else algo = is_float_scalar(h->scalar) ? PCX_FIR_DIRECT : PCX_FIR_EXACT;
    }
    if (algo == PCX_FIR_OLS_FFT && !((fast && h->have_ols) || h->have_poly || h->have_real_ols || h->have_ols64 || h->have_ols_int || h->have_ols_real64 ||
                                     h->have_interp64 || h->have_interp_real)) {
        set_error("fir: OLS_FFT needs complex_float32 and K<=8193 (resampling: K<=2049, L<=64 rows) or complex_float64 / complex_int16 / complex_int8 with M=L=1, 2<=K<=4097");
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
    if (algo == PCX_FIR_OLS_FFT && h->have_interp_real) {
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
