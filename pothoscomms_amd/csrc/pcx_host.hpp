// pcx_host.hpp -- what the translation units behind include/pcx.h share (pcx_api.hip implements it; pcx_fir_api.hip and
// pcx_fft_api.hip hold the handles): argument checks, control-plane uploads, a handle's execution context, the staging of pageable
// host buffers and the launch shape of calls on page-locked ones.  Not installed.
#pragma once
#include <algorithm>
#include <chrono>
#include <complex>
#include <cstdio>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>

#include "pcx_internal.hpp"

#define PCX_CHECK_ARG(cond, ...)        \
    do {                                \
        if (!(cond)) {                  \
            ::pcx::set_error(__VA_ARGS__);     \
            return PCX_ERR_ARG;         \
        }                               \
    } while (0)

namespace pcx {

// upload host bytes into a DevBuf (control plane: COMPLETE on the device on return; the recipe in pcx_api.hip)
int upload_bytes(DevBuf &b, const void *src, size_t bytes);
template <typename T>
inline int upload(DevBuf &b, const std::vector<T> &v) { return upload_bytes(b, v.data(), v.size() * sizeof(T)); }

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
int ctx_own_stream(ExecCtx &c, hipStream_t *out);
int ctx_enter(ExecCtx &c, hipStream_t st);
int ctx_quiesce(ExecCtx &c);
size_t stage_piece(size_t bytes);
void stage_copy(void *dst, const void *src, size_t bytes);
int stage_reserve(const void *host, size_t bytes, StageBuf &ws);
int stage_in(const void *host, size_t bytes, StageBuf &ws, hipStream_t st, const void **dev);
int stage_out_begin(void *host, size_t bytes, StageBuf &ws, void **dev, bool *staged);
int stage_out_first(StageBuf &ws, size_t bytes, bool staged, hipStream_t st);
int stage_out_rest(void *host, StageBuf &ws, size_t bytes, bool staged, hipStream_t st);
int stage_out_end(void *host, size_t bytes, StageBuf &ws, bool staged, hipStream_t st);
bool host_page_locked(const void *p);
int drain_chunks(size_t bytes);
int drain_setup(ExecCtx &c, int nchunks);
int drain_chunk(ExecCtx &c, int i, hipStream_t compute, void *host_dst, const void *dev_src, size_t bytes);
int drain_finish(ExecCtx &c, hipStream_t compute);
size_t drain_from();
size_t drain_chunk_bytes();
unsigned host_grid();
unsigned host_map_grid(unsigned dflt = 32);
// (grid, map grid) for a launch whose input or output is page-locked host memory the kernel addresses in place, else (0, 0)
struct LinkBound : LinkBoundScope {
    static bool any(const void *a, const void *b, const void *c) { return host_page_locked(a) || host_page_locked(b) || (c && host_page_locked(c)); }
    LinkBound(const void *a, const void *b, const void *c = nullptr, unsigned map_blocks = 32) : LinkBound(any(a, b, c), map_blocks) {}

private:
    // (the pointers are looked up ONCE per call: each lookup is a hipPointerGetAttributes)
    LinkBound(bool link, unsigned map_blocks) : LinkBoundScope(link ? host_grid() : 0, link ? host_map_grid(map_blocks) : 0) {}
};

}  // namespace pcx
