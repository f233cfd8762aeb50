// pcx_shard.hip -- ONE complex_float32 sample stream over the GPUs of a node, behind the C ABI (include/pcx.h,
// "pcx_shard_*"): overlap-save sharding with the tap-length halo moved between neighbouring devices by RCCL
// send/recv over xGMI (BASELINE.json configs[3]).
//
// The dependency that makes this possible is in the reference's loop: output n reads inputs n .. n+K-1 only
// (filter/FIRFilter.cpp:296-299), and a work() call leaves the last K-1 inputs un-consumed as the next call's history
// (:305-307).  A stream of G*C samples therefore splits into G contiguous shards of C samples; shard g needs the LAST K-1
// samples of shard g-1 in front of its own -- the same "history at the front" buffer pcx_fir_process_dev takes -- and
// shard 0 keeps the stream's own history.  One process, one communicator over the devices (ncclCommInitAll), one stream
// pair and one FIR handle per device, 2,032 bytes per boundary and pass for 255 taps, no other collective.
//
// Per device g and pass (pcx_shard_step) -- ONE kernel launch per shard:
//     compute stream   [record in_ready] .. the whole shard, its blocks walked back to front; block 0, the only one whose window
//                                            reaches into the halo slot, is dealt LAST and waits for the gate word
//     halo stream      [wait in_ready(g, g-1)]  recv halo <- g-1 / send tail -> g+1  [gate word <- pass number] [record halo_done]
// (pcx_fir_process_dev_gated / pcx_sched.hpp Gate).  The exchange -- pure latency for 2 KB -- hides behind the rest of the
// shard, and there is no second launch with its own start-up, ragged end and kernel boundary (round 2 ran a body launch and
// a head launch behind an event: +14 % per pass with two shards on one device, profiles/r02/shard_probe.txt).  Configurations
// without a gated kernel (anything but the 4096-sample complex_float32 plan, or shards of fewer than ~2048 blocks) keep the two
// launches: body = outputs head..C-1 with head >= K-1 so that it never touches the halo slot, then the head behind halo_done.
// in_ready also orders the NEXT pass's receive behind this pass's kernel, which reads the halo slot; halo_done(g) is waited for
// by compute stream g-1 at the end of the step, so that nothing queued on it later (the caller's next fill) can overwrite the
// tail of shard g-1 while the exchange is still reading it.
//
// The fused chain Rotate -> FIR -> FreqDemod (pcx_shard_set_chain, BASELINE configs[4]) shards the same way with a halo of K
// samples -- the FIR's K-1 and the one sample FreqDemod's `_prev` needs (FreqDemod.cpp:63-65) -- and one extra output in front
// of every shard but the first, computed only to be that predecessor and dropped.  Every pass starts from the reset state, as
// a single-device run of the whole stream does.
//
// RCCL is loaded on first use (dlopen "librccl.so.1"): a single-GPU Pothos process never maps the 570 MB library, and
// a process that already holds a copy (PyTorch bundles one under the same SONAME) shares it instead of loading a second.
// The declarations come from <rccl/rccl.h>; only the symbol lookup is dynamic.
//
// PCX_SHARD_PEER_COPY moves the halo with hipMemcpyPeerAsync instead.  It exists so the sharding logic (offsets, the
// head/body split, the event ordering) can be exercised with several shards on ONE device, which RCCL refuses
// ("duplicate GPU"); it is also a correct multi-device transport wherever peer access works.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "pcx_internal.hpp"
#include "pcx_sched.hpp"

namespace {

struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
};

int load_rccl(const RcclApi **out)
{
    static RcclApi api;
    static int state = 0;   // 0 untried, 1 ok, -1 failed
    if (state == 0) {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (api.lib) break;
        }
        state = -1;
        if (api.lib) {
            bool ok = true;
            auto sym = [&](const char *name) { void *p = dlsym(api.lib, name); if (!p) ok = false; return p; };
            api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(sym("ncclCommInitAll"));
            api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
            api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
            api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
            api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
            api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
            api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
            api.GetVersion = reinterpret_cast<decltype(api.GetVersion)>(sym("ncclGetVersion"));
            if (ok) state = 1;
        }
    }
    if (state != 1) {
        pcx::set_error("pcx_shard: RCCL is not loadable (%s)", api.lib ? "a symbol of the send/recv API is missing" : dlerror());
        return PCX_ERR_UNSUPPORTED;
    }
    *out = &api;
    return PCX_OK;
}

#define PCX_RCCL(api, expr)                                                                                       \
    do {                                                                                                          \
        ncclResult_t r__ = (expr);                                                                                \
        if (r__ != ncclSuccess) {                                                                                 \
            ::pcx::set_error("%s: %s (%s:%d)", #expr, (api)->GetErrorString(r__), __FILE__, __LINE__);            \
            return PCX_ERR_HIP;                                                                                   \
        }                                                                                                         \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; } }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

}  // namespace

using namespace pcx;

// (diagnostic library, PCX_SHARD_TIMING=1) host time per phase of a pass, printed when a handle is destroyed: tools/shard_probe.py
#ifdef PCX_DIAG
namespace {
struct PhaseClock {
    static constexpr int kPhases = 8;
    double acc[kPhases] = {};
    unsigned long long passes = 0;
    std::chrono::steady_clock::time_point t;
    void start() { t = std::chrono::steady_clock::now(); }
    void lap(int ph) { const auto n = std::chrono::steady_clock::now(); acc[ph] += std::chrono::duration<double>(n - t).count(); t = n; }
};
PhaseClock g_phase;
}
#define PCX_PHASE_START() do { if (PCX_ENV_SET("PCX_SHARD_TIMING")) g_phase.start(); } while (0)
#define PCX_PHASE_LAP(ph) do { if (PCX_ENV_SET("PCX_SHARD_TIMING")) g_phase.lap(ph); } while (0)
#else
#define PCX_PHASE_START() do { } while (0)
#define PCX_PHASE_LAP(ph) do { } while (0)
#endif

// ---- submit threads ---------------------------------------------------------------------------------------------------------------
// Queueing a pass costs the host 16-20 us PER SHARD from one thread (3-4 us per cross-stream wait, 7 per gated launch, the records, the
// signal kernel: PCX_SHARD_TIMING in the diagnostic library itemises it) -- 135-165 us for eight shards against a pass of 195 us.
// With submit threads every DEVICE's share of a pass is queued by a thread of its own (bound to that device once), the caller's thread
// posts the parts of the pass in turn and waits for each to be queued: what the host pays is the longest device plus a hand-over per part.  The threads spin for a while behind a part (a stream of passes finds them awake) and sleep after that.
struct pcx_shard;
static int shard_part(pcx_shard *s, int g, int part);
struct ShardWorkers {
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv;
    std::atomic<unsigned> gen{0};
    std::atomic<int> pending{0}, sleepers{0};
    std::atomic<bool> stop{false};
    int part = 0;
    std::vector<int> rc;
    std::vector<std::string> err;
    static constexpr int kSpinUs = 400;      // how long a thread stays awake behind its last part (two passes of a 64 Mi-sample stream)

    // ONE THREAD PER DEVICE: the shards of a device are queued by that device's thread, in shard order, part by part -- several threads
    // calling into ONE device's runtime convoy on its locks (eight threads on one device: 146 -> 102 us to queue a pass on one box,
    // 137 -> 403 on another, profiles/r05/README.md), threads on different devices do not share them
    void start(pcx_shard *s, int G, const std::vector<int> &dev)
    {
        std::vector<int> devices;
        for (int g = 0; g < G; g++)
            if (std::find(devices.begin(), devices.end(), dev[g]) == devices.end()) devices.push_back(dev[g]);
        rc.assign(devices.size(), 0);
        err.assign(devices.size(), std::string());
        for (size_t w = 0; w < devices.size(); w++) {
            std::vector<int> mine;
            for (int g = 0; g < G; g++)
                if (dev[g] == devices[w]) mine.push_back(g);
            th.emplace_back([this, s, w, mine, d = devices[w]] { run(s, (int)w, mine, d); });
        }
    }
    void run(pcx_shard *s, int g, const std::vector<int> &shards, int device)
    {
        (void)hipSetDevice(device);
        unsigned seen = 0;
        for (;;) {
            const auto t0 = std::chrono::steady_clock::now();
            while (gen.load(std::memory_order_acquire) == seen && !stop.load(std::memory_order_acquire)) {
                if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(kSpinUs)) {
                    std::unique_lock<std::mutex> lk(m);
                    sleepers.fetch_add(1);
                    // (sequentially consistent on both sides: the caller bumps `gen` and THEN looks at `sleepers`, this thread bumps `sleepers` and THEN
                    // looks at `gen` -- one of the two must see the other's write, or a wake-up is lost)
                    cv.wait(lk, [&] { return gen.load() != seen || stop.load(); });
                    sleepers.fetch_sub(1);
                    break;
                }
                __builtin_ia32_pause();
            }
            if (stop.load(std::memory_order_acquire)) return;
            seen = gen.load(std::memory_order_acquire);
            rc[g] = PCX_OK;
            for (int q = part & 0xff; q <= (part >> 8) && rc[g] == PCX_OK; q++)      // parts [first, last], part by part, of this device's shards
                for (size_t k = 0; k < shards.size() && rc[g] == PCX_OK; k++) rc[g] = shard_part(s, shards[k], q);
            if (rc[g] != PCX_OK) err[g] = pcx_last_error();
            pending.fetch_sub(1, std::memory_order_acq_rel);
        }
    }
    // every shard's parts first .. last, each shard on its own thread; returns when all of it is queued
    int all(int first, int last)
    {
        part = first | (last << 8);
        pending.store((int)th.size(), std::memory_order_release);
        gen.fetch_add(1);
        if (sleepers.load() > 0) { std::lock_guard<std::mutex> lk(m); cv.notify_all(); }
        while (pending.load(std::memory_order_acquire) != 0) __builtin_ia32_pause();
        for (size_t g = 0; g < rc.size(); g++)
            if (rc[g] != PCX_OK) { set_error("%s", err[g].c_str()); return rc[g]; }
        return PCX_OK;
    }
    ~ShardWorkers()
    {
        stop.store(true, std::memory_order_release);
        { std::lock_guard<std::mutex> lk(m); cv.notify_all(); }
        for (auto &t : th) if (t.joinable()) t.join();
    }
};

struct pcx_shard {
    std::unique_ptr<ShardWorkers> workers;        // pcx_shard_set_submit_threads: one thread per device queues that device's share of a pass
    bool tables_ready = false;                    // every shard's tables are on its device (reset by whatever changes them)
    unsigned cur_pass = 0;                        // the pass number the parts of the pass in flight use
    int G = 0;
    int transport = PCX_SHARD_RCCL;
    std::vector<int> dev;
    const RcclApi *rccl = nullptr;
    std::vector<ncclComm_t> comm;
    std::vector<hipStream_t> st, hst;             // compute / halo stream per shard
    std::vector<hipEvent_t> in_ready, halo_ready;  // halo_ready(g): the exchange that WROTE shard g's halo slot and READ shard g-1's tail is done
    std::vector<pcx_fir *> fir;
    std::vector<pcx_fmchain *> chain;             // chain mode: the fused Rotate -> FIR -> FreqDemod handle of each device
    std::vector<void *> gate;                     // per shard: the 32-bit gate word (device memory), holds the pass number
    unsigned *gate_seen = nullptr;                // page-locked, two words per shard: what shard_check_gates last read of gate[g]
    bool chain_mode = false;
    double phase = 0.0;
    std::vector<double> taps;                     // as given to set_taps (chain mode re-applies them with the phase)
    int complex_taps = 0;
    std::vector<void *> alloc, out;               // per shard: [lead | halo | C] (cf32) and the outputs (cf32, or float32 in chain mode: 1 + C)
    std::vector<std::unique_ptr<PinBuf>> bounce_in, bounce_out;   // scatter / gather of PAGEABLE host memory (pcx_api.hip stage_in)
    size_t K = 1, C = 0, head = 0;
    std::vector<size_t> lead;                     // per shard: samples in front of the halo slot (alignment, pcx_shard_configure)
    bool have_taps = false;
    bool use_gate = true;                         // pcx_shard_set_gated: false = body launch, halo event, head launch (two launches per shard)
    bool exchanged_once = false;                  // the first exchange has completed (RCCL sets its connections up lazily: pcx_shard_step)
    bool posted = false;                          // pcx_shard_post_exchange has queued this pass's exchange, pcx_shard_compute has not run yet
    unsigned long long steps = 0;
    size_t halo() const { return chain_mode ? K : K - 1; }      // samples in front of every shard
    float2 *in_ptr(int g) const { return static_cast<float2 *>(alloc[g]) + lead[g]; }       // the halo slot
    float2 *hist_ptr(int g) const { return in_ptr(g) + (halo() - (K - 1)); }                  // the K-1 history samples in front of the shard
};

static constexpr size_t kHead = 4096;   // two-launch fallback: outputs computed after the halo has landed (at least K-1: the body must not read the halo slot)

#define PCX_CHECK_ARG(cond, ...)        \
    do {                                \
        if (!(cond)) {                  \
            set_error(__VA_ARGS__);     \
            return PCX_ERR_ARG;         \
        }                               \
    } while (0)

#define PCX_CHECK_STATE(cond, ...)      \
    do {                                \
        if (!(cond)) {                  \
            set_error(__VA_ARGS__);     \
            return PCX_ERR_STATE;       \
        }                               \
    } while (0)

static void shard_free_buffers(pcx_shard *s)
{
    for (int g = 0; g < s->G; g++) {
        (void)hipSetDevice(s->dev[g]);
        if (g < (int)s->alloc.size() && s->alloc[g]) (void)hipFree(s->alloc[g]);
        if (g < (int)s->out.size() && s->out[g]) (void)hipFree(s->out[g]);
    }
    s->alloc.assign(s->G, nullptr);
    s->out.assign(s->G, nullptr);
    s->C = 0;
}

int pcx_shard_destroy(pcx_shard *s)
{
#ifdef PCX_DIAG
    if (PCX_ENV_SET("PCX_SHARD_TIMING") && g_phase.passes) {
        static const char *const names[] = {"tables", "in_ready records", "halo-stream waits", "exchange (copies / RCCL group)", "gate signals + halo_ready records",
                                            "launches", "compute-stream waits"};
        fprintf(stderr, "pcx(diag): host time per pass over %llu passes:", g_phase.passes);
        for (int i = 0; i < 7; i++) fprintf(stderr, " %s %.1f us;", names[i], g_phase.acc[i] / g_phase.passes * 1e6);
        fprintf(stderr, "\n");
        g_phase = PhaseClock();
    }
#endif
    if (!s) return PCX_OK;
    s->workers.reset();             // (the submit threads first: they hold the handle)
    DeviceGuard guard;
    for (int g = 0; g < s->G; g++) {
        (void)hipSetDevice(s->dev[g]);
        if (g < (int)s->st.size() && s->st[g]) (void)hipStreamSynchronize(s->st[g]);
        if (g < (int)s->hst.size() && s->hst[g]) (void)hipStreamSynchronize(s->hst[g]);
    }
    if (s->rccl)
        for (ncclComm_t c : s->comm)
            if (c) (void)s->rccl->CommDestroy(c);
    shard_free_buffers(s);
    if (s->gate_seen) (void)hipHostFree(s->gate_seen);
    for (int g = 0; g < s->G; g++) {
        (void)hipSetDevice(s->dev[g]);
        if (g < (int)s->fir.size() && s->fir[g]) (void)pcx_fir_destroy(s->fir[g]);
        if (g < (int)s->chain.size() && s->chain[g]) (void)pcx_fmchain_destroy(s->chain[g]);
        if (g < (int)s->gate.size() && s->gate[g]) (void)hipFree(s->gate[g]);
        if (g < (int)s->in_ready.size() && s->in_ready[g]) (void)hipEventDestroy(s->in_ready[g]);
        if (g < (int)s->halo_ready.size() && s->halo_ready[g]) (void)hipEventDestroy(s->halo_ready[g]);
        if (g < (int)s->st.size() && s->st[g]) (void)hipStreamDestroy(s->st[g]);
        if (g < (int)s->hst.size() && s->hst[g]) (void)hipStreamDestroy(s->hst[g]);
    }
    delete s;
    return PCX_OK;
}

int pcx_shard_create(int nshards, const int *devices, int transport, pcx_shard **out)
{
    PCX_CHECK_ARG(out, "null out");
    PCX_CHECK_ARG(nshards >= 1 && nshards <= 64, "pcx_shard: %d shards", nshards);
    PCX_CHECK_ARG(transport == PCX_SHARD_RCCL || transport == PCX_SHARD_PEER_COPY, "pcx_shard: unknown transport %d", transport);
    int ndev = 0;
    PCX_HIP(hipGetDeviceCount(&ndev));
    std::vector<int> dev(nshards);
    for (int g = 0; g < nshards; g++) {
        dev[g] = devices ? devices[g] : g;
        PCX_CHECK_ARG(dev[g] >= 0 && dev[g] < ndev, "pcx_shard: shard %d on device %d, %d visible", g, dev[g], ndev);
    }
    if (transport == PCX_SHARD_RCCL) {
        std::vector<int> sorted(dev);
        std::sort(sorted.begin(), sorted.end());
        PCX_CHECK_ARG(std::adjacent_find(sorted.begin(), sorted.end()) == sorted.end(),
                      "pcx_shard: RCCL needs one distinct device per shard (PCX_SHARD_PEER_COPY takes several shards per device)");
    }
    pcx_shard *s = new (std::nothrow) pcx_shard();
    if (!s) { set_error("out of memory"); return PCX_ERR_STATE; }
    s->G = nshards; s->transport = transport; s->dev = dev;
    s->comm.assign(nshards, nullptr);
    s->st.assign(nshards, nullptr); s->hst.assign(nshards, nullptr);
    s->in_ready.assign(nshards, nullptr); s->halo_ready.assign(nshards, nullptr);
    s->fir.assign(nshards, nullptr); s->chain.assign(nshards, nullptr); s->gate.assign(nshards, nullptr);
    s->alloc.assign(nshards, nullptr); s->out.assign(nshards, nullptr);
    s->lead.assign(nshards, 0);
    for (int g = 0; g < nshards; g++) { s->bounce_in.emplace_back(new PinBuf()); s->bounce_out.emplace_back(new PinBuf()); }
    DeviceGuard guard;
    auto fail = [&](int rc) { (void)pcx_shard_destroy(s); return rc; };
    if (hipHostMalloc(reinterpret_cast<void **>(&s->gate_seen), 2 * sizeof(unsigned) * nshards, hipHostMallocDefault) != hipSuccess) {
        s->gate_seen = nullptr;
        set_error("pcx_shard: page-locked allocation failed: %s", hipGetErrorString(hipGetLastError()));
        return fail(PCX_ERR_HIP);
    }
    std::memset(s->gate_seen, 0, 2 * sizeof(unsigned) * nshards);
    for (int g = 0; g < nshards; g++) {
        // (halo streams at the device's highest stream priority were tried -- the exchange ahead of the passes' own kernels -- and
        // made every pass SLOWER: 0.2242 -> 0.2411 ms with two shards on one device, 0.2674 -> 0.4570 with eight,
        // profiles/r03/shard_probe.txt; PCX_SHARD_HALO_PRIO=1 in the diagnostic library repeats it)
        int prio_lo = 0, prio_hi = 0, halo_prio = 0;
        const long want_prio = PCX_ENV_INT("PCX_SHARD_HALO_PRIO", 0);      // (diagnostic library) 1: highest, -1: lowest, 0: default
        if (want_prio != 0 && hipSetDevice(dev[g]) == hipSuccess) {
            if (hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) != hipSuccess) { (void)hipGetLastError(); prio_lo = prio_hi = 0; }
            halo_prio = want_prio > 0 ? prio_hi : prio_lo;
        }
        if (hipSetDevice(dev[g]) != hipSuccess || hipStreamCreateWithFlags(&s->st[g], hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithPriority(&s->hst[g], hipStreamNonBlocking, halo_prio) != hipSuccess ||
            hipEventCreateWithFlags(&s->in_ready[g], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s->halo_ready[g], hipEventDisableTiming) != hipSuccess ||
            hipMalloc(&s->gate[g], 256) != hipSuccess || hipMemset(s->gate[g], 0, 256) != hipSuccess) {
            set_error("pcx_shard: stream/event setup on device %d failed: %s", dev[g], hipGetErrorString(hipGetLastError()));
            return fail(PCX_ERR_HIP);
        }
        // one /comms/fir_filter handle per device, created with that device current: it stays bound to it (pcx.h)
        const int rc = pcx_fir_create(PCX_F32, 1, 1, &s->fir[g]);
        if (rc != PCX_OK) return fail(rc);
    }
    if (transport == PCX_SHARD_RCCL) {
        int rc = load_rccl(&s->rccl);
        if (rc != PCX_OK) return fail(rc);
        const ncclResult_t r = s->rccl->CommInitAll(s->comm.data(), nshards, dev.data());
        if (r != ncclSuccess) {
            set_error("ncclCommInitAll over %d device(s): %s", nshards, s->rccl->GetErrorString(r));
            s->comm.assign(nshards, nullptr);
            return fail(PCX_ERR_HIP);
        }
    } else {
        // peer copies between distinct devices want peer access; without it the runtime stages through the host (still correct)
        for (int g = 1; g < nshards; g++)
            if (dev[g] != dev[g - 1]) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, dev[g], dev[g - 1]) == hipSuccess && can) {
                    (void)hipSetDevice(dev[g]);
                    if (hipDeviceEnablePeerAccess(dev[g - 1], 0) != hipSuccess) (void)hipGetLastError();   // already enabled is fine
                }
            }
    }
    *out = s;
    return PCX_OK;
}

// taps (and, in chain mode, the phase) on every device's handle
static int shard_apply_taps(pcx_shard *s)
{
    const size_t ntaps = s->K;
    if (s->chain_mode) {
        for (int g = 0; g < s->G; g++) {
            PCX_TRY(pcx_fmchain_set_phase(s->chain[g], s->phase));
            PCX_TRY(pcx_fmchain_set_taps(s->chain[g], s->taps.data(), ntaps, s->complex_taps));
        }
        return PCX_OK;
    }
    // the per-device FIR handles are COMPLEX-tap filters; REAL taps are the same filter with zero imaginary parts
    std::vector<double> t(2 * ntaps);
    for (size_t k = 0; k < ntaps; k++) {
        t[2 * k] = s->complex_taps ? s->taps[2 * k] : s->taps[k];
        t[2 * k + 1] = s->complex_taps ? s->taps[2 * k + 1] : 0.0;
    }
    for (int g = 0; g < s->G; g++) PCX_TRY(pcx_fir_set_taps(s->fir[g], t.data(), ntaps));
    return PCX_OK;
}

int pcx_shard_set_taps(pcx_shard *s, const double *taps, size_t ntaps, int complex_taps)
{
    PCX_CHECK_ARG(s, "null handle");
    PCX_CHECK_ARG(ntaps > 0 && taps, "FIRFilter::setTaps(): taps cannot be empty");
    // (new taps change the spectrum the posted pass would multiply by, a new K frees the buffers its exchange is writing into)
    PCX_CHECK_STATE(!s->posted, "pcx_shard_set_taps: this handle's exchange is posted; pcx_shard_compute comes first");
    const size_t Kold = s->K;
    s->taps.assign(taps, taps + ntaps * (complex_taps ? 2 : 1));
    s->complex_taps = complex_taps ? 1 : 0;
    s->K = ntaps;
    PCX_TRY(shard_apply_taps(s));
    s->tables_ready = false;
    s->have_taps = true;
    if (s->C && ntaps != Kold) {   // the halo slot in front of every shard changes size: the buffers must be laid out again
        DeviceGuard guard;
        shard_free_buffers(s);
    }
    return PCX_OK;
}

int pcx_shard_set_chain(pcx_shard *s, int enable, double phase)
{
    PCX_CHECK_ARG(s, "null handle");
    PCX_CHECK_STATE(!s->posted, "pcx_shard_set_chain: this handle's exchange is posted; pcx_shard_compute comes first");
    DeviceGuard guard;
    const bool was = s->chain_mode;
    if (enable) {
        for (int g = 0; g < s->G; g++)
            if (!s->chain[g]) {
                PCX_HIP(hipSetDevice(s->dev[g]));
                PCX_TRY(pcx_fmchain_create(&s->chain[g]));
            }
    }
    s->chain_mode = enable != 0;
    s->phase = phase;
    s->tables_ready = false;
    if (s->have_taps) PCX_TRY(shard_apply_taps(s));
    if (s->C && was != s->chain_mode) shard_free_buffers(s);   // another halo, another output type: lay the buffers out again
    return PCX_OK;
}

int pcx_shard_set_submit_threads(pcx_shard *s, int enable)
{
    PCX_CHECK_ARG(s, "null handle");
    PCX_CHECK_STATE(!s->posted, "pcx_shard_set_submit_threads: this handle's exchange is posted; pcx_shard_compute comes first");
    if (!enable) { s->workers.reset(); return PCX_OK; }
    if (s->workers || s->G < 2) return PCX_OK;
    try {
        s->workers.reset(new ShardWorkers());
        s->workers->start(s, s->G, s->dev);
    } catch (const std::exception &e) {
        s->workers.reset();
        set_error("pcx_shard_set_submit_threads: %s", e.what());
        return PCX_ERR_STATE;
    }
    return PCX_OK;
}

int pcx_shard_set_gated(pcx_shard *s, int enable)
{
    PCX_CHECK_ARG(s, "null handle");
    s->use_gate = enable != 0;
    return PCX_OK;
}

int pcx_shard_set_algo(pcx_shard *s, int algo)
{
    PCX_CHECK_ARG(s, "null handle");
    PCX_CHECK_STATE(!s->posted, "pcx_shard_set_algo: this handle's exchange is posted; pcx_shard_compute comes first");
    s->tables_ready = false;
    for (int g = 0; g < s->G; g++) {
        PCX_TRY(pcx_fir_set_algo(s->fir[g], algo));
        if (s->chain[g] && (algo == PCX_FIR_AUTO || algo == PCX_FIR_DIRECT || algo == PCX_FIR_OLS_FFT)) PCX_TRY(pcx_fmchain_set_algo(s->chain[g], algo));
    }
    return PCX_OK;
}

int pcx_shard_configure(pcx_shard *s, size_t shard_elems)
{
    PCX_CHECK_ARG(s, "null handle");
    PCX_CHECK_ARG(s->have_taps, "pcx_shard_configure: set the taps first (the halo is K-1 samples)");
    PCX_CHECK_STATE(!s->posted, "pcx_shard_configure: this handle's exchange is posted; pcx_shard_compute comes first");
    PCX_CHECK_ARG(shard_elems >= 1, "pcx_shard_configure: empty shard");
    const size_t halo = s->halo();
    PCX_CHECK_ARG(s->G == 1 || shard_elems >= halo, "pcx_shard_configure: a shard of %zu samples is shorter than the %zu-sample halo its neighbour needs",
                  shard_elems, halo);
    DeviceGuard guard;
    shard_free_buffers(s);
    s->tables_ready = false;
    // [lead | halo | shard (C)], placed so that every 2 KiB row the overlap-save kernels load and every row they store starts on a
    // 128-byte line (measured 0.2245 -> 0.2187 ms per 64 Mi samples against a line-aligned halo).  FIR: the kernel rounds its block
    // overlap K-1 up to 16 samples and starts its windows `pad` samples before the history, so the history goes pad samples behind
    // a line -- which puts the SHARD on a line.  Chain: the overlap is K rounded up to 32 and a window starts 1 + pad samples
    // before the call's first sample: shard 0 is called on [K-1 history | C] (reset state, as a single-device run), every other
    // shard on [K halo | C] (one extra output in front, dropped), so their slots differ by one sample.
    for (int g = 0; g < s->G; g++) {
        if (!s->chain_mode) s->lead[g] = (16 - (s->K - 1) % 16) % 16;
        else {
            const size_t pad = (s->K + 31) / 32 * 32 - s->K;
            s->lead[g] = (pad + (g == 0 ? 0 : 1)) % 16;
        }
    }
    for (int g = 0; g < s->G; g++) {
        PCX_HIP(hipSetDevice(s->dev[g]));
        PCX_HIP(hipMalloc(&s->alloc[g], (s->lead[g] + halo + shard_elems) * sizeof(float2)));
        PCX_HIP(hipMalloc(&s->out[g], s->chain_mode ? (shard_elems + 1) * sizeof(float) : shard_elems * sizeof(float2)));
        PCX_HIP(hipMemsetAsync(s->alloc[g], 0, (s->lead[g] + halo) * sizeof(float2), s->st[g]));   // stream start: zero history
    }
    s->C = shard_elems;
    // Several shards on ONE device (the one-GPU rehearsal; a node with fewer devices than shards) share its 1024 resident
    // workgroup slots.  TWO shards side by side want half each, so that both launches are resident at once and end together
    // (two 32 Mi-sample shards: 0.2196 ms with 1024 each, 0.1988 with 512 each; one 64 Mi launch 0.1971; tools/ab_gated_slots.sh,
    // profiles/r03/shard_probe.txt).  MORE shards must NOT go on dividing: the runtime maps a process's streams onto four hardware
    // queues, so no more than four of the launches run side by side whatever their size, and eight launches of 128 workgroups leave
    // half the device idle -- configs[3]'s eight 64 Mi-sample shards on one device: 3.00 ms per pass at 128 slots each, 1.82 at 256,
    // 1.57 at 512, 1.55 at 1024 (one 512 Mi launch: 1.50); 512 is within 1.5 % of the best for every shard count from 2 to 8
    // (profiles/r04/shard_probe_c3.txt, shard_probe_c3_slots.txt).  A device that carries ONE shard gives it all 1024.
    for (int g = 0; g < s->G; g++) {
        unsigned same = 0;
        for (int k = 0; k < s->G; k++) same += s->dev[k] == s->dev[g];
        const unsigned slots = same >= 2 ? 512 : 1024;
        // (Fewer slots under the RCCL transport, room for its protocol kernel, looked like 6 % on an all-zero probe and is nothing on data:
        // profiles/r04/rccl_pass_slots.txt.  A shard alone on its device takes all 1024.)
        const unsigned fir_slots = slots;
        fir_set_slots(s->fir[g], fir_slots);
        if (s->chain[g]) fmchain_set_slots(s->chain[g], slots);
    }
    // the two-launch fallback's split: the body (outputs head .. C-1) reads in[head ..], which must lie behind the halo slot
    // in[0 .. K-2] whatever K -- a fixed 4096 let filters of more than 4097 taps read a halo that had not arrived yet
    s->head = std::min(shard_elems, std::max(kHead, (halo + kHead - 1) / kHead * kHead));
    return PCX_OK;
}

int pcx_shard_info(const pcx_shard *s, int *nshards, size_t *K, size_t *shard_elems, int *transport)
{
    PCX_CHECK_ARG(s, "null handle");
    if (nshards) *nshards = s->G;
    if (K) *K = s->K;
    if (shard_elems) *shard_elems = s->C;
    if (transport) *transport = s->transport;
    return PCX_OK;
}

int pcx_shard_buffers(pcx_shard *s, int g, void **in_dev, void **out_dev, void **stream, int *device)
{
    PCX_CHECK_ARG(s, "null handle");
    PCX_CHECK_ARG(g >= 0 && g < s->G, "pcx_shard_buffers: shard %d of %d", g, s->G);
    PCX_CHECK_ARG(s->C, "pcx_shard_buffers: call pcx_shard_configure first");
    if (in_dev) *in_dev = s->hist_ptr(g);
    if (out_dev) *out_dev = s->chain_mode ? static_cast<void *>(static_cast<float *>(s->out[g]) + 1) : s->out[g];
    if (stream) *stream = s->st[g];
    if (device) *device = s->dev[g];
    return PCX_OK;
}

// Page-locked caller memory is given to the DMA engine as it is; pageable memory goes through a page-locked bounce buffer of the
// shard's own, copied by the CPU -- the library never hands a pageable pointer to hipMemcpyAsync (profiles/r02/contention.md).
int pcx_shard_scatter(pcx_shard *s, const void *host_stream, size_t elems)
{
    PCX_TRACE();
    PCX_CHECK_ARG(s && host_stream, "null argument");
    PCX_CHECK_ARG(s->C, "pcx_shard_scatter: call pcx_shard_configure first");
    PCX_CHECK_STATE(!s->posted, "pcx_shard_scatter: this handle's exchange is posted and reading the shard buffers; pcx_shard_compute comes first");
    PCX_CHECK_ARG(elems == s->K - 1 + (size_t)s->G * s->C, "pcx_shard_scatter: %zu elements, expected K-1 + shards*C = %zu", elems,
                  s->K - 1 + (size_t)s->G * s->C);
    DeviceGuard guard;
    const float2 *x = static_cast<const float2 *>(host_stream);
    const bool locked = device_alias(host_stream) != nullptr;
    if (!locked) {
        // every bounce buffer is in place BEFORE the first transfer is queued: a page-locked allocation made while another shard's
        // copy was pending left that copy's destination zero (first pass only, AMD_DIRECT_DISPATCH=0: tools/shard_dd_probe.py)
        // -- and nothing of an earlier pass is still running on ANY shard while they are allocated
        for (int g = 0; g < s->G; g++) {
            PCX_HIP(hipSetDevice(s->dev[g]));
            PCX_HIP(hipStreamSynchronize(s->st[g]));      // the bounce buffer's previous transfer
            PCX_HIP(hipStreamSynchronize(s->hst[g]));
        }
        for (int g = 0; g < s->G; g++) {
            PCX_HIP(hipSetDevice(s->dev[g]));
            PCX_TRY(s->bounce_in[g]->ensure((s->K - 1 + s->C) * sizeof(float2)));
        }
    }
    for (int g = 0; g < s->G; g++) {
        PCX_HIP(hipSetDevice(s->dev[g]));
        // shard 0 also takes the stream's own K-1 history; every other halo slot is filled by the exchange of each pass
        const size_t skip = g == 0 ? 0 : s->K - 1;
        const size_t bytes = (s->K - 1 - skip + s->C) * sizeof(float2);
        const float2 *src = x + (size_t)g * s->C + skip;
        float2 *dst = s->hist_ptr(g) + skip;
        if (!locked) {
            PinBuf &b = *s->bounce_in[g];
            constexpr size_t kPiece = (size_t)4 << 20;     // the CPU copies piece i+1 while piece i is on the wire
            for (size_t off = 0; off < bytes; off += kPiece) {
                const size_t c = bytes - off < kPiece ? bytes - off : kPiece;
                std::memcpy(static_cast<char *>(b.p) + off, reinterpret_cast<const char *>(src) + off, c);
                PCX_HIP(hipMemcpyAsync(reinterpret_cast<char *>(dst) + off, static_cast<const char *>(b.p) + off, c, hipMemcpyHostToDevice, s->st[g]));
            }
        } else {
            PCX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s->st[g]));
        }
    }
    return PCX_OK;
}

// A gated launch that gave up waiting for its halo (pcx_sched.hpp gate_wait: two seconds) went ahead on whatever the halo slot held
// and said so in the word behind its gate word.  Every call that hands results to the caller ends here: queue_gate_reads() puts the
// read of each shard's two words BEHIND the pass on its compute stream, check_gate_reads() looks at them once those streams have
// been waited for.  A reported timeout is cleared (the next pass starts clean) and the call fails: the seam outputs of that pass
// are not the stream's.
static int queue_gate_reads(pcx_shard *s)
{
    for (int g = 1; g < s->G && s->steps; g++) {
        PCX_HIP(hipSetDevice(s->dev[g]));
        PCX_HIP(hipMemcpyAsync(s->gate_seen + 2 * g, s->gate[g], 2 * sizeof(unsigned), hipMemcpyDeviceToHost, s->st[g]));
    }
    return PCX_OK;
}
static int check_gate_reads(pcx_shard *s)
{
    int bad = -1;
    unsigned word = 0;
    for (int g = 1; g < s->G && s->steps; g++)
        if (s->gate_seen[2 * g + 1] == kGateTimedOut) {
            if (bad < 0) { bad = g; word = s->gate_seen[2 * g]; }
            s->gate_seen[2 * g + 1] = 0;
            PCX_HIP(hipSetDevice(s->dev[g]));
            PCX_HIP(hipMemsetAsync(static_cast<unsigned *>(s->gate[g]) + 1, 0, sizeof(unsigned), s->st[g]));
            PCX_HIP(hipStreamSynchronize(s->st[g]));
        }
    if (bad >= 0) {
        set_error("pcx_shard: shard %d did not receive its halo within two seconds of a pass (gate word %u, pass %llu): "
                  "the outputs at that seam were computed on a stale halo", bad, word, s->steps);
        return PCX_ERR_STATE;
    }
    return PCX_OK;
}

int pcx_shard_gather(pcx_shard *s, void *host_out, size_t elems)
{
    PCX_TRACE();
    PCX_CHECK_ARG(s && host_out, "null argument");
    PCX_CHECK_ARG(s->C && elems == (size_t)s->G * s->C, "pcx_shard_gather: %zu elements, expected shards*C = %zu", elems, (size_t)s->G * s->C);
    DeviceGuard guard;
    const size_t esz = s->chain_mode ? sizeof(float) : sizeof(float2);
    char *y = static_cast<char *>(host_out);
    const bool locked = device_alias(host_out) != nullptr;
    const size_t bytes = s->C * esz;
    if (!locked) {
        // allocations first, transfers afterwards (see pcx_shard_scatter) -- and not while the pass is still running: gather waits
        // for it in any case, so it waits BEFORE the first page-locked allocation (the halo copies of a pass came out zero when
        // the first gather allocated behind them; AMD_DIRECT_DISPATCH=0 with eight processes on the device)
        bool grow = false;
        for (int g = 0; g < s->G; g++) grow = grow || s->bounce_out[g]->cap < bytes;
        if (grow) PCX_TRY(pcx_shard_sync(s));
        for (int g = 0; g < s->G; g++) {
            PCX_HIP(hipSetDevice(s->dev[g]));
            PCX_TRY(s->bounce_out[g]->ensure(bytes));
        }
    }
    for (int g = 0; g < s->G; g++) {
        PCX_HIP(hipSetDevice(s->dev[g]));
        void *dst = locked ? static_cast<void *>(y + (size_t)g * bytes) : s->bounce_out[g]->p;
        const void *src = s->chain_mode ? static_cast<const void *>(static_cast<const float *>(s->out[g]) + 1) : s->out[g];
        PCX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s->st[g]));
    }
    PCX_TRY(queue_gate_reads(s));
    for (int g = 0; g < s->G; g++) {
        PCX_HIP(hipSetDevice(s->dev[g]));
        PCX_HIP(hipStreamSynchronize(s->st[g]));
        if (!locked) std::memcpy(y + (size_t)g * bytes, s->bounce_out[g]->p, bytes);
    }
    return check_gate_reads(s);      // the outputs are copied either way; the caller learns that a seam is wrong
}

int pcx_shard_sync(pcx_shard *s)
{
    PCX_CHECK_ARG(s, "null handle");
    DeviceGuard guard;
    PCX_TRY(queue_gate_reads(s));
    for (int g = 0; g < s->G; g++) {
        PCX_HIP(hipSetDevice(s->dev[g]));
        PCX_HIP(hipStreamSynchronize(s->st[g]));
        PCX_HIP(hipStreamSynchronize(s->hst[g]));
    }
    return check_gate_reads(s);
}

// FIR mode: outputs [first_out, first_out + n_out) of shard g, which read in[first_out : first_out + n_out + K - 1].
// gate != nullptr: through the gated entry point; *gated = 0 means nothing was queued.
static int shard_run_fir(pcx_shard *s, int g, size_t first_out, size_t n_out, const void *gate, unsigned value, int *gated)
{
    size_t c = 0, p = 0;
    const float2 *in = s->in_ptr(g) + first_out;
    float2 *out = static_cast<float2 *>(s->out[g]) + first_out;
    if (gate) {
        PCX_TRY(pcx_fir_process_dev_gated(s->fir[g], in, n_out + s->K - 1, out, n_out, &c, &p, gate, value, s->st[g], gated));
        if (!*gated) return PCX_OK;
    } else {
        PCX_TRY(pcx_fir_process_dev(s->fir[g], in, n_out + s->K - 1, out, n_out, &c, &p, s->st[g]));
    }
    if (c != n_out || p != n_out) { set_error("pcx_shard: shard %d produced %zu of %zu outputs", g, p, n_out); return PCX_ERR_STATE; }
    return PCX_OK;
}
// chain mode: the whole shard from the reset state.  Shard 0: [K-1 history | C] -> C outputs at out + 1; every other shard:
// [K halo | C] -> 1 + C outputs at out (the first exists only to be the demodulator's predecessor of the second).
static int shard_run_chain(pcx_shard *s, int g, const void *gate, unsigned value, int *gated)
{
    size_t c = 0, p = 0;
    const size_t extra = g == 0 ? 0 : 1, n_out = s->C + extra;
    const float2 *in = s->in_ptr(g) + (1 - extra);
    float *out = static_cast<float *>(s->out[g]) + (1 - extra);
    PCX_TRY(pcx_fmchain_reset(s->chain[g]));
    if (gate) {
        PCX_TRY(pcx_fmchain_process_dev_gated(s->chain[g], in, n_out + s->K - 1, out, n_out, &c, &p, gate, value, s->st[g], gated));
        if (!*gated) return PCX_OK;
    } else {
        PCX_TRY(pcx_fmchain_process_dev(s->chain[g], in, n_out + s->K - 1, out, n_out, &c, &p, s->st[g]));
    }
    if (c != n_out || p != n_out) { set_error("pcx_shard: chain shard %d produced %zu of %zu outputs", g, p, n_out); return PCX_ERR_STATE; }
    return PCX_OK;
}

// ---- a pass, shard by shard ------------------------------------------------------------------------------------------------
// What shard g contributes to each part of a pass.  The parts are separated where one shard's calls depend on an event ANOTHER shard's
// thread records (hipStreamWaitEvent takes the event as it stands when the call is made): every shard's part p is queued before any
// shard's part p + 1.  One thread walks the shards part by part; with submit threads (pcx_shard_set_submit_threads) every shard's part is
// queued by its own thread, all at once.
enum ShardPart {
    kPartTables,      // this shard's tables on its device (nothing of the pass is queued yet: the rule of pcx_shard_scatter)
    kPartInReady,     // 1. inputs in place: everything queued on the compute stream so far (the caller's fill / scatter, and the previous
                      //    pass's kernel, which READ the halo slot this pass overwrites)
    kPartHaloWaits,   // 2a. the halo stream behind that -- and, peer copies, behind the SOURCE shard's inputs as well
    kPartPeerCopy,    // 2b. peer copies: tail of shard g-1 -> halo slot of shard g, in place  [RCCL: one group call from the caller's thread]
    kPartSignal,      // 2c. behind the exchange: the gate word, the halo_ready event
    kPartLaunch,      // 3. the shard in ONE launch, its first block behind the gate (shard 0: a plain launch)
    kPartFence,       // 4. later writers on the compute stream wait for the exchange that reads this shard's tail
};
static int shard_part(pcx_shard *s, int g, int part)
{
    const int G = s->G;
    const size_t halo = s->halo(), hbytes = halo * sizeof(float2);
    const bool rccl = s->transport == PCX_SHARD_RCCL;
    const unsigned pass = s->cur_pass;
    PCX_HIP(hipSetDevice(s->dev[g]));
    switch (part) {
    case kPartTables:
        if (s->chain_mode) PCX_TRY(fmchain_prepare(s->chain[g]));
        else PCX_TRY(fir_prepare(s->fir[g]));
        return PCX_OK;
    case kPartInReady:
        PCX_HIP(hipEventRecord(s->in_ready[g], s->st[g]));
        return PCX_OK;
    case kPartHaloWaits:
        PCX_HIP(hipStreamWaitEvent(s->hst[g], s->in_ready[g], 0));
        if (!rccl && g > 0) PCX_HIP(hipStreamWaitEvent(s->hst[g], s->in_ready[g - 1], 0));   // the source shard's samples
        return PCX_OK;
    case kPartPeerCopy:
        if (!rccl && g > 0) PCX_HIP(hipMemcpyPeerAsync(s->in_ptr(g), s->dev[g], s->in_ptr(g - 1) + s->C, s->dev[g - 1], hbytes, s->hst[g]));
        return PCX_OK;
    case kPartSignal: {
        const long drop = PCX_ENV_INT("PCX_SHARD_DROP_SIGNAL", 0);   // (diagnostic library only) the pass whose gate signals are left out: the timeout path's test
        if (g > 0 && !(drop > 0 && (long)pass == drop)) PCX_TRY(pcx_gate_signal_dev(s->gate[g], pass, s->hst[g]));
        // halo_ready(g): RCCL -- the send that reads shard g's tail and the receive into its halo slot are done;
        // peer copies -- the copy that reads shard g-1's tail and writes shard g's halo slot is done
        if (g > 0 || rccl) PCX_HIP(hipEventRecord(s->halo_ready[g], s->hst[g]));
        return PCX_OK;
    }
    case kPartLaunch: {
        auto whole = [&]() -> int {   // the whole shard in one plain call
            return s->chain_mode ? shard_run_chain(s, g, nullptr, 0, nullptr) : shard_run_fir(s, g, 0, s->C, nullptr, 0, nullptr);
        };
        if (g == 0 || G == 1 || halo == 0) return whole();
        int gated = 0;
        if (s->use_gate) {
            if (s->chain_mode) PCX_TRY(shard_run_chain(s, g, s->gate[g], pass, &gated));
            else PCX_TRY(shard_run_fir(s, g, 0, s->C, s->gate[g], pass, &gated));
        }
        if (gated) return PCX_OK;
        // no gated kernel for this configuration: the round-2 scheme.  FIR: the body while the halo is in flight, then the head
        // behind it; chain (long filters, short shards): the halo first, then the shard
        if (!s->chain_mode && s->C > s->head) PCX_TRY(shard_run_fir(s, g, s->head, s->C - s->head, nullptr, 0, nullptr));
        PCX_HIP(hipStreamWaitEvent(s->st[g], s->halo_ready[g], 0));
        if (s->chain_mode) PCX_TRY(shard_run_chain(s, g, nullptr, 0, nullptr));
        else PCX_TRY(shard_run_fir(s, g, 0, s->head, nullptr, 0, nullptr));
        return PCX_OK;
    }
    case kPartFence:
        // whatever is queued on a compute stream after this step -- the caller's next fill, the next scatter -- must not overwrite
        // the tail of its shard while the exchange is still reading it
        if (g + 1 < G) PCX_HIP(hipStreamWaitEvent(s->st[g], s->halo_ready[rccl ? g : g + 1], 0));
        return PCX_OK;
    }
    return PCX_OK;
}
// Parts [first, last] of every shard.  The caller names ranges whose parts have no dependency ACROSS shards between them (a shard's
// calls inside the range only concern events its own thread records, or events of an EARLIER range), so with submit threads a range
// goes out as ONE hand-over; one thread queues it part by part, shard by shard.
static int shard_parts(pcx_shard *s, int first, int last)
{
    if (s->workers) {
        PCX_TRY(s->workers->all(first, last));
        PCX_PHASE_LAP(last);
        return PCX_OK;
    }
    for (int part = first; part <= last; part++) {
        for (int g = 0; g < s->G; g++) PCX_TRY(shard_part(s, g, part));
        PCX_PHASE_LAP(part);
    }
    return PCX_OK;
}

// pcx_shard_step in its two halves (include/pcx.h): the exchange of the halos of what the shard buffers hold NOW, and the pass over it.
int pcx_shard_post_exchange(pcx_shard *s)
{
    PCX_TRACE();
    PCX_CHECK_ARG(s, "null handle");
    PCX_CHECK_ARG(s->C, "pcx_shard_post_exchange: call pcx_shard_configure first");
    PCX_CHECK_STATE(!s->posted, "pcx_shard_post_exchange: the exchange of this handle is already posted; pcx_shard_compute comes next");
    DeviceGuard guard;
    const int G = s->G;
    const size_t halo = s->halo(), hbytes = halo * sizeof(float2);
    PCX_PHASE_START();
    // every shard's tables are on its device BEFORE anything of the pass is queued: uploading them lazily -- allocations and
    // transfers of the control plane -- between other shards' queued work lost one shard's pass (AMD_DIRECT_DISPATCH=0,
    // tests/test_shard_gpu.py retap test; the rule of pcx_shard_scatter)
    if (!s->tables_ready) {
        PCX_TRY(shard_parts(s, kPartTables, kPartTables));
        s->tables_ready = true;
    }
    if (G == 1 || halo == 0) {      // nothing to exchange
        s->posted = true;
        return PCX_OK;
    }
    // the value the gate words take in this pass (compared by signed distance).  Taken BEFORE anything is queued: a step that fails
    // half-way has used its number up -- the next one must not find its gate already open
    s->cur_pass = (unsigned)++s->steps;
    if (s->transport == PCX_SHARD_RCCL) {
        PCX_TRY(shard_parts(s, kPartInReady, kPartHaloWaits));
        PCX_RCCL(s->rccl, s->rccl->GroupStart());
        for (int g = 0; g < G; g++) {
            if (g + 1 < G) PCX_RCCL(s->rccl, s->rccl->Send(s->in_ptr(g) + s->C, hbytes, ncclChar, g + 1, s->comm[g], s->hst[g]));   // the last `halo` samples
            if (g > 0) PCX_RCCL(s->rccl, s->rccl->Recv(s->in_ptr(g), hbytes, ncclChar, g - 1, s->comm[g], s->hst[g]));
        }
        PCX_RCCL(s->rccl, s->rccl->GroupEnd());
        PCX_PHASE_LAP(kPartPeerCopy);
        PCX_TRY(shard_parts(s, kPartSignal, kPartSignal));
    } else {
        PCX_TRY(shard_parts(s, kPartInReady, kPartInReady));      // (the copy into shard g waits for shard g-1's inputs: a range of its own)
        PCX_TRY(shard_parts(s, kPartHaloWaits, kPartSignal));
    }
    // The FIRST exchange of a handle is waited for on the host before any launch is queued that depends on it: RCCL sets its
    // point-to-point connections up lazily, inside the first send / receive, and that can take longer than the two seconds a gated
    // launch waits for its halo -- the first pass of a run must not be the one that reports a timeout.
    if (!s->exchanged_once) {
        for (int g = 0; g < G; g++) {
            PCX_HIP(hipSetDevice(s->dev[g]));
            PCX_HIP(hipStreamSynchronize(s->hst[g]));
        }
        s->exchanged_once = true;
    }
    s->posted = true;
    return PCX_OK;
}

int pcx_shard_compute(pcx_shard *s)
{
    PCX_TRACE();
    PCX_CHECK_ARG(s, "null handle");
    PCX_CHECK_STATE(s->posted, "pcx_shard_compute: no exchange posted (pcx_shard_post_exchange first, or pcx_shard_step for both)");
    s->posted = false;              // whatever happens below, this pass's exchange is used up
    PCX_CHECK_ARG(s->C, "pcx_shard_compute: the shard buffers are not laid out (pcx_shard_configure first)");   // (never launch on null + lead)
    DeviceGuard guard;
    PCX_PHASE_START();
    if (s->G == 1 || s->halo() == 0) {
        // nothing was exchanged: each shard is one plain call (with one device, exactly pcx_fir_process_dev on the whole stream)
        PCX_TRY(shard_parts(s, kPartLaunch, kPartLaunch));
        s->steps++;
        return PCX_OK;
    }
    PCX_TRY(shard_parts(s, kPartLaunch, kPartFence));
#ifdef PCX_DIAG
    g_phase.passes++;
#endif
    return PCX_OK;
}

int pcx_shard_step(pcx_shard *s)
{
    PCX_TRACE();
    PCX_CHECK_ARG(s, "null handle");
    PCX_CHECK_ARG(s->C, "pcx_shard_step: call pcx_shard_configure first");
    PCX_TRY(pcx_shard_post_exchange(s));
    return pcx_shard_compute(s);
}
