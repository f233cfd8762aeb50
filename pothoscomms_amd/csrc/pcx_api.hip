// pcx_api.hip -- the extern "C" boundary of libpcx_hip.so (include/pcx.h), part 1: errors, devices, memory, page-locking, the
// link probe, and what every handle shares (pcx_host.hpp): control-plane uploads, execution contexts, staging of pageable
// host buffers.  The handles themselves: pcx_fir_api.hip (FIR, fused chain), pcx_fft_api.hip (FFT, FreqDemod, the maps).
// Host-side only: all arithmetic on stream data happens in the HIP kernels.
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

#include "pcx_host.hpp"

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

// upload host bytes into a DevBuf (control plane: COMPLETE on return; the recipe above)
int upload_bytes(DevBuf &b, const void *src, size_t bytes)
{
    PCX_TRY(b.ensure(bytes));
    if (bytes == 0) return PCX_OK;
    ControlLane *lane;
    PCX_TRY(control_lane(&lane));
    PCX_TRY(lane->pin.ensure(bytes));
    std::memcpy(lane->pin.p, src, bytes);
    PCX_HIP(hipMemcpyAsync(b.p, lane->pin.p, bytes, hipMemcpyHostToDevice, lane->st));
    PCX_HIP(hipStreamSynchronize(lane->st));
    return PCX_OK;
}

}  // namespace pcx

using namespace pcx;

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



namespace pcx {

// (call with the handle's DeviceScope alive)
int ctx_own_stream(ExecCtx &c, hipStream_t *out)
{
    if (!c.own) PCX_HIP(hipStreamCreateWithFlags(&c.own, hipStreamNonBlocking));
    *out = c.own;
    return PCX_OK;
}
// before enqueuing on `st`: order it behind the handle's previous enqueue if that went to another stream
int ctx_enter(ExecCtx &c, hipStream_t st)
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
int ctx_quiesce(ExecCtx &c)
{
    if (c.have_last) PCX_HIP(hipStreamSynchronize(c.last));
    return PCX_OK;
}

// Device-visible alias of a HOST pointer when it is page-locked (pcx_host_alloc / hipHostMalloc / hipHostRegister; any
// offset inside the allocation), else nullptr.  The host-pointer entry points launch their kernels straight on such
// buffers -- measured on MI355X (tools/pcie_lab.hip, 128 MiB each way): a kernel reading and writing pinned host memory
// moves 43 GB/s in BOTH directions at once, against 28 GB/s for H2D, kernel, D2H through a staging workspace -- and stage
// only pageable memory, which the device cannot address.
void *device_alias(const void *p)
{
    if (!p) return nullptr;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if ((a.type == hipMemoryTypeHost || a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged) && a.devicePointer) return a.devicePointer;
    return nullptr;
}
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
size_t stage_piece(size_t bytes)
{
    constexpr size_t kMin = (size_t)1 << 20, kMaxPieces = 16;
    size_t piece = (bytes + kMaxPieces - 1) / kMaxPieces;
    piece = (piece + 4095) & ~(size_t)4095;
    return piece < kMin ? kMin : piece;
}
// The CPU side of a large staged transfer: one core copies 10-14 GB/s, the DMA engine moves 50.  Pieces of 4 MiB and more
// are split over up to four short-lived helper threads (the caller copies the first share itself); smaller copies, i.e.
// every call below 64 MiB, stay on the calling thread.
void stage_copy(void *dst, const void *src, size_t bytes)
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
int stage_reserve(const void *host, size_t bytes, StageBuf &ws)
{
    if (device_alias(host)) return PCX_OK;
    PCX_TRY(ws.dev.ensure(bytes));
    PCX_TRY(ws.pin.ensure(bytes));
    return PCX_OK;
}
int stage_in(const void *host, size_t bytes, StageBuf &ws, hipStream_t st, const void **dev)
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
int stage_out_begin(void *host, size_t bytes, StageBuf &ws, void **dev, bool *staged)
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
int stage_out_first(StageBuf &ws, size_t bytes, bool staged, hipStream_t st)
{
    if (!staged || !bytes) return PCX_OK;
    const size_t piece = stage_piece(bytes), c = bytes < piece ? bytes : piece;
    PCX_HIP(hipMemcpyAsync(ws.pin.p, ws.dev.p, c, hipMemcpyDeviceToHost, st));
    return PCX_OK;
}
int stage_out_rest(void *host, StageBuf &ws, size_t bytes, bool staged, hipStream_t st)
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
int stage_out_end(void *host, size_t bytes, StageBuf &ws, bool staged, hipStream_t st)
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
size_t drain_from() { return (size_t)PCX_ENV_INT("PCX_DRAIN_FROM", (long)1 << 62); }     // product: never
size_t drain_chunk_bytes() { return (size_t)PCX_ENV_INT("PCX_DRAIN_CHUNK", 2 << 20); }
// A call whose input or output is HOST memory the kernel addresses over PCIe is bound by the link, not by the device -- and the link is
// full duplex.  On the device-resident grid (1024 persistent workgroups, one or a few blocks each) a call of a Pothos slab's size
// is a few hundred blocks that all load, then all compute, then all store: reads and writes never overlap (1 Mi samples: 0.281 ms).
// On ~48 workgroups that walk several blocks each on the grid stride, block k+1's fetch (issued at the foot of block k) runs beside block
// k's stores: 0.255 ms at 1 Mi samples, 0.833 against 0.947 at 4 Mi, 3.08 against 3.35 at 16 Mi = 43.6 GB/s each way, the rate
// tools/pcie_lab.hip measures for a plain copy kernel on the same buffers (profiles/r05/host_slots_static.txt; 32 and 64 are within 2 %).
unsigned host_grid() { return (unsigned)PCX_ENV_INT("PCX_HOST_GRID", 48); }
// the grid-stride map kernels (pcx_internal.hpp LINK-BOUND LAUNCHES): 32 blocks for the one-to-one maps, 64 for /comms/freq_demod (two reads per sample)
unsigned host_map_grid(unsigned dflt) { const unsigned e = (unsigned)PCX_ENV_INT("PCX_HOST_MAP_GRID", -1); return e == (unsigned)-1 ? dflt : e; }
thread_local unsigned g_link_grid = 0, g_link_map_grid = 0;
// page-locked HOST memory, reached over the link: what the link-bound launch shape is for.  Managed memory is addressed in place as
// well (pcx_pointer_kind files it under page-locked), but it may be resident in HBM: it keeps the device-resident grid.
bool host_page_locked(const void *p)
{
    if (!p) return false;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}
// how many chunks a drained output of `bytes` goes in (2 .. kDrainChunks)
int drain_chunks(size_t bytes)
{
    const size_t c = drain_chunk_bytes();
    size_t n = (bytes + c - 1) / (c ? c : 1);
    if (n < 2) n = 2;
    if (n > (size_t)ExecCtx::kDrainChunks) n = ExecCtx::kDrainChunks;
    return (int)n;
}
// (everything is created BEFORE the first transfer of a call is queued: profiles/r02/contention.md section 4)
int drain_setup(ExecCtx &c, int nchunks)
{
    if (!c.drain) PCX_HIP(hipStreamCreateWithFlags(&c.drain, hipStreamNonBlocking));
    for (int i = 0; i < nchunks; i++)
        if (!c.drain_ev[i]) PCX_HIP(hipEventCreateWithFlags(&c.drain_ev[i], hipEventDisableTiming));
    return PCX_OK;
}
// chunk i's kernels are queued on `compute`: its bytes leave for the caller's (page-locked) buffer behind them
int drain_chunk(ExecCtx &c, int i, hipStream_t compute, void *host_dst, const void *dev_src, size_t bytes)
{
    if (!bytes) return PCX_OK;
    PCX_HIP(hipEventRecord(c.drain_ev[i], compute));
    PCX_HIP(hipStreamWaitEvent(c.drain, c.drain_ev[i], 0));
    PCX_HIP(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, c.drain));
    return PCX_OK;
}
// the result is in the caller's buffer on return (the drain stream is behind every chunk's kernels)
int drain_finish(ExecCtx &c, hipStream_t compute)
{
    PCX_HIP(hipStreamSynchronize(c.drain));
    PCX_HIP(hipStreamSynchronize(compute));
    return PCX_OK;
}

}  // namespace pcx
