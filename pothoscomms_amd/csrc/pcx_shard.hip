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
// Per device g and pass (pcx_shard_step):
//     compute stream   [record in_ready] ......... body: outputs head..C-1 .......... [wait halo_ready] head: outputs 0..head-1
//     halo stream      [wait in_ready(g, g-1)]  recv halo <- g-1 / send tail -> g+1  [record halo_ready]
// Only the first `head` = 4096 outputs read the halo, so the exchange (pure latency for 2 KB) hides behind the body of
// the pass.  in_ready also orders the NEXT pass's receive behind this pass's head kernel, which reads the halo slot.
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
#include <memory>
#include <new>
#include <vector>

#include "pcx_internal.hpp"

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

struct pcx_shard {
    int G = 0;
    int transport = PCX_SHARD_RCCL;
    std::vector<int> dev;
    const RcclApi *rccl = nullptr;
    std::vector<ncclComm_t> comm;
    std::vector<hipStream_t> st, hst;             // compute / halo stream per shard
    std::vector<hipEvent_t> in_ready, halo_ready;
    std::vector<pcx_fir *> fir;
    std::vector<void *> alloc, out;               // per shard: [lead | halo K-1 | C] and C outputs (cf32)
    std::vector<std::unique_ptr<PinBuf>> bounce_in, bounce_out;   // scatter / gather of PAGEABLE host memory (pcx_api.hip stage_in)
    size_t K = 1, C = 0, lead = 0, head = 0;
    bool have_taps = false;
    unsigned long long steps = 0;
    float2 *in_ptr(int g) const { return static_cast<float2 *>(alloc[g]) + lead; }
};

static constexpr size_t kHead = 4096;   // outputs computed after the halo has landed: one overlap-save block's worth, whatever K

#define PCX_CHECK_ARG(cond, ...)        \
    do {                                \
        if (!(cond)) {                  \
            set_error(__VA_ARGS__);     \
            return PCX_ERR_ARG;         \
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
    if (!s) return PCX_OK;
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
    for (int g = 0; g < s->G; g++) {
        (void)hipSetDevice(s->dev[g]);
        if (g < (int)s->fir.size() && s->fir[g]) (void)pcx_fir_destroy(s->fir[g]);
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
    s->fir.assign(nshards, nullptr);
    s->alloc.assign(nshards, nullptr); s->out.assign(nshards, nullptr);
    for (int g = 0; g < nshards; g++) { s->bounce_in.emplace_back(new PinBuf()); s->bounce_out.emplace_back(new PinBuf()); }
    DeviceGuard guard;
    auto fail = [&](int rc) { (void)pcx_shard_destroy(s); return rc; };
    for (int g = 0; g < nshards; g++) {
        if (hipSetDevice(dev[g]) != hipSuccess || hipStreamCreateWithFlags(&s->st[g], hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithFlags(&s->hst[g], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&s->in_ready[g], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s->halo_ready[g], hipEventDisableTiming) != hipSuccess) {
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

int pcx_shard_set_taps(pcx_shard *s, const double *taps, size_t ntaps, int complex_taps)
{
    PCX_CHECK_ARG(s, "null handle");
    PCX_CHECK_ARG(ntaps > 0 && taps, "FIRFilter::setTaps(): taps cannot be empty");
    // the per-device handles are COMPLEX-tap filters; REAL taps are the same filter with zero imaginary parts
    std::vector<double> t(2 * ntaps);
    for (size_t k = 0; k < ntaps; k++) { t[2 * k] = complex_taps ? taps[2 * k] : taps[k]; t[2 * k + 1] = complex_taps ? taps[2 * k + 1] : 0.0; }
    const size_t Kold = s->K;
    for (int g = 0; g < s->G; g++) PCX_TRY(pcx_fir_set_taps(s->fir[g], t.data(), ntaps));
    s->K = ntaps;
    s->have_taps = true;
    if (s->C && ntaps != Kold) {   // the halo slot in front of every shard changes size: the buffers must be laid out again
        DeviceGuard guard;
        shard_free_buffers(s);
    }
    return PCX_OK;
}

int pcx_shard_set_algo(pcx_shard *s, int algo)
{
    PCX_CHECK_ARG(s, "null handle");
    for (int g = 0; g < s->G; g++) PCX_TRY(pcx_fir_set_algo(s->fir[g], algo));
    return PCX_OK;
}

int pcx_shard_configure(pcx_shard *s, size_t shard_elems)
{
    PCX_CHECK_ARG(s, "null handle");
    PCX_CHECK_ARG(s->have_taps, "pcx_shard_configure: set the taps first (the halo is K-1 samples)");
    PCX_CHECK_ARG(shard_elems >= 1, "pcx_shard_configure: empty shard");
    PCX_CHECK_ARG(s->G == 1 || shard_elems >= s->K - 1, "pcx_shard_configure: a shard of %zu samples is shorter than the %zu-sample halo its neighbour needs",
                  shard_elems, s->K - 1);
    DeviceGuard guard;
    shard_free_buffers(s);
    // [lead | halo (K-1) | shard (C)] with the SHARD on a 128-byte line: the overlap-save kernel rounds its block overlap
    // up to 16 samples, so with this placement every 2 KiB row it loads and every row it stores starts on a line
    // (measured 0.2245 -> 0.2187 ms per 64 Mi samples against a line-aligned halo)
    s->lead = (16 - (s->K - 1) % 16) % 16;
    for (int g = 0; g < s->G; g++) {
        PCX_HIP(hipSetDevice(s->dev[g]));
        PCX_HIP(hipMalloc(&s->alloc[g], (s->lead + s->K - 1 + shard_elems) * sizeof(float2)));
        PCX_HIP(hipMalloc(&s->out[g], shard_elems * sizeof(float2)));
        PCX_HIP(hipMemsetAsync(s->alloc[g], 0, (s->lead + s->K - 1) * sizeof(float2), s->st[g]));   // stream start: zero history
    }
    s->C = shard_elems;
    s->head = std::min(kHead, shard_elems);
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
    if (in_dev) *in_dev = s->in_ptr(g);
    if (out_dev) *out_dev = s->out[g];
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
        if (!locked) {
            PinBuf &b = *s->bounce_in[g];
            constexpr size_t kPiece = (size_t)4 << 20;     // the CPU copies piece i+1 while piece i is on the wire
            for (size_t off = 0; off < bytes; off += kPiece) {
                const size_t c = bytes - off < kPiece ? bytes - off : kPiece;
                std::memcpy(static_cast<char *>(b.p) + off, reinterpret_cast<const char *>(src) + off, c);
                PCX_HIP(hipMemcpyAsync(reinterpret_cast<char *>(s->in_ptr(g) + skip) + off, static_cast<const char *>(b.p) + off, c,
                                       hipMemcpyHostToDevice, s->st[g]));
            }
        } else {
            PCX_HIP(hipMemcpyAsync(s->in_ptr(g) + skip, src, bytes, hipMemcpyHostToDevice, s->st[g]));
        }
    }
    return PCX_OK;
}

int pcx_shard_gather(pcx_shard *s, void *host_out, size_t elems)
{
    PCX_TRACE();
    PCX_CHECK_ARG(s && host_out, "null argument");
    PCX_CHECK_ARG(s->C && elems == (size_t)s->G * s->C, "pcx_shard_gather: %zu elements, expected shards*C = %zu", elems, (size_t)s->G * s->C);
    DeviceGuard guard;
    float2 *y = static_cast<float2 *>(host_out);
    const bool locked = device_alias(host_out) != nullptr;
    const size_t bytes = s->C * sizeof(float2);
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
        void *dst = locked ? static_cast<void *>(y + (size_t)g * s->C) : s->bounce_out[g]->p;
        PCX_HIP(hipMemcpyAsync(dst, s->out[g], bytes, hipMemcpyDeviceToHost, s->st[g]));
    }
    for (int g = 0; g < s->G; g++) {
        PCX_HIP(hipSetDevice(s->dev[g]));
        PCX_HIP(hipStreamSynchronize(s->st[g]));
        if (!locked) std::memcpy(y + (size_t)g * s->C, s->bounce_out[g]->p, bytes);
    }
    return PCX_OK;
}

int pcx_shard_sync(pcx_shard *s)
{
    PCX_CHECK_ARG(s, "null handle");
    DeviceGuard guard;
    for (int g = 0; g < s->G; g++) {
        PCX_HIP(hipSetDevice(s->dev[g]));
        PCX_HIP(hipStreamSynchronize(s->st[g]));
        PCX_HIP(hipStreamSynchronize(s->hst[g]));
    }
    return PCX_OK;
}

static int shard_run(pcx_shard *s, int g, size_t first_out, size_t n_out)
{
    // outputs [first_out, first_out + n_out) read in[first_out : first_out + n_out + K - 1]
    size_t c = 0, p = 0;
    PCX_TRY(pcx_fir_process_dev(s->fir[g], s->in_ptr(g) + first_out, n_out + s->K - 1, static_cast<float2 *>(s->out[g]) + first_out, n_out, &c, &p,
                                s->st[g]));
    if (c != n_out || p != n_out) { set_error("pcx_shard: shard %d produced %zu of %zu outputs", g, p, n_out); return PCX_ERR_STATE; }
    return PCX_OK;
}

int pcx_shard_step(pcx_shard *s)
{
    PCX_TRACE();
    PCX_CHECK_ARG(s, "null handle");
    PCX_CHECK_ARG(s->C, "pcx_shard_step: call pcx_shard_configure first");
    DeviceGuard guard;
    const int G = s->G;
    const size_t halo = s->K - 1, hbytes = halo * sizeof(float2);
    // every shard's tables are on its device BEFORE anything of the pass is queued: uploading them lazily -- allocations and
    // transfers of the control plane -- between other shards' queued work lost one shard's pass (AMD_DIRECT_DISPATCH=0,
    // tests/test_shard_gpu.py retap test; the rule of pcx_shard_scatter)
    for (int g = 0; g < G; g++) {
        PCX_HIP(hipSetDevice(s->dev[g]));
        PCX_TRY(fir_prepare(s->fir[g]));
    }
    if (G == 1 || halo == 0) {
        // nothing to exchange: each shard is one plain call (with one device, exactly pcx_fir_process_dev on the whole stream)
        for (int g = 0; g < G; g++) {
            PCX_HIP(hipSetDevice(s->dev[g]));
            PCX_TRY(shard_run(s, g, 0, s->C));
        }
        s->steps++;
        return PCX_OK;
    }
    // 1. inputs of this pass are in place once everything queued on the compute streams so far has run (the caller's
    //    fill / scatter, and the previous pass's head kernel, which READ the halo slot this pass overwrites)
    for (int g = 0; g < G; g++) {
        PCX_HIP(hipSetDevice(s->dev[g]));
        PCX_HIP(hipEventRecord(s->in_ready[g], s->st[g]));
    }
    // 2. the exchange, on the halo streams: tail of shard g -> halo slot of shard g+1, in place
    for (int g = 0; g < G; g++) {
        PCX_HIP(hipSetDevice(s->dev[g]));
        PCX_HIP(hipStreamWaitEvent(s->hst[g], s->in_ready[g], 0));
        if (s->transport == PCX_SHARD_PEER_COPY && g > 0) PCX_HIP(hipStreamWaitEvent(s->hst[g], s->in_ready[g - 1], 0));   // the source shard's samples
    }
    if (s->transport == PCX_SHARD_RCCL) {
        PCX_RCCL(s->rccl, s->rccl->GroupStart());
        for (int g = 0; g < G; g++) {
            if (g + 1 < G) PCX_RCCL(s->rccl, s->rccl->Send(s->in_ptr(g) + s->C, hbytes, ncclChar, g + 1, s->comm[g], s->hst[g]));   // last K-1 samples
            if (g > 0) PCX_RCCL(s->rccl, s->rccl->Recv(s->in_ptr(g), hbytes, ncclChar, g - 1, s->comm[g], s->hst[g]));
        }
        PCX_RCCL(s->rccl, s->rccl->GroupEnd());
    } else {
        for (int g = 1; g < G; g++) {
            PCX_HIP(hipSetDevice(s->dev[g]));
            PCX_HIP(hipMemcpyPeerAsync(s->in_ptr(g), s->dev[g], s->in_ptr(g - 1) + s->C, s->dev[g - 1], hbytes, s->hst[g]));
        }
    }
    for (int g = 1; g < G; g++) {
        PCX_HIP(hipSetDevice(s->dev[g]));
        PCX_HIP(hipEventRecord(s->halo_ready[g], s->hst[g]));
    }
    // 3. the body of every shard while the halos are in flight (it does not touch the halo slot) ...
    if (s->C > s->head)
        for (int g = 0; g < G; g++) {
            PCX_HIP(hipSetDevice(s->dev[g]));
            PCX_TRY(shard_run(s, g, s->head, s->C - s->head));
        }
    // 4. ... then the head, behind the halo
    for (int g = 0; g < G; g++) {
        PCX_HIP(hipSetDevice(s->dev[g]));
        if (g > 0) PCX_HIP(hipStreamWaitEvent(s->st[g], s->halo_ready[g], 0));
        PCX_TRY(shard_run(s, g, 0, s->head));
    }
    s->steps++;
    return PCX_OK;
}
