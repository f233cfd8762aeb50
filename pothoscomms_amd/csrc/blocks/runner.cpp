// runner.cpp -- extern "C" driver over the bundled block runtime (include/pcx_blocks.h).
// Plays the part of the Pothos scheduler for one block at a time: plants port buffers and
// labels, sets workInfo, calls work() and propagateLabels(), reports consume/produce/reserve.
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <string>

#include <sys/mman.h>
#include <unistd.h>

#include "pcx.h"
#include "pcx_blocks.h"
#include "pcx_framework.hpp"

#ifdef PCX_WITH_POTHOS
#error "the runner drives the bundled runtime; inside Pothos the framework does this"
#endif

using namespace pcxfw;

struct pcxb_block {
    // managers first: destroyed after the block (members are destroyed in reverse order), and a pinned slab may still
    // be in use by the block's last call until its handle is gone
    pcxfw::BufferManager::Sptr inManager, outManager;   // what the block handed the scheduler (pcxb_acquire_buffer)
    pcxfw::BufferManager::Sptr linkIn;                  // the manager negotiated for the edge INTO this block (pcxb_link_buffer)
    std::unique_ptr<Block> blk;
    size_t initialReserve = std::numeric_limits<size_t>::max();
};

static thread_local std::string g_err;

template <typename F>
static int guarded(F &&f)
{
    try {
        f();
        return PCX_OK;
    } catch (const InvalidArgumentException &e) {
        g_err = e.what();
        return PCX_ERR_ARG;
    } catch (const BlockCallNotFound &e) {
        g_err = e.what();
        return PCX_ERR_STATE;
    } catch (const std::exception &e) {
        g_err = e.what();
        // device-side failures carry the ABI's own message; map "unsupported" back
        return g_err.find("not implemented") != std::string::npos || g_err.find("only power-of-two") != std::string::npos
                   ? PCX_ERR_UNSUPPORTED
                   : PCX_ERR_HIP;
    }
}

static Label toLabel(const pcxb_label &l)
{
    Object data;
    if (l.kind == PCXB_SIZE) data = Object((unsigned long)l.uval);
    else if (l.kind == PCXB_DOUBLE) data = Object(l.dval);
    else if (l.kind == PCXB_STRING) data = Object(std::string(l.sval));
    return Label(l.id, data, (size_t)l.index, (size_t)l.width);
}
static void fromLabel(const Label &l, pcxb_label &o)
{
    std::memset(&o, 0, sizeof(o));
    std::snprintf(o.id, sizeof(o.id), "%s", l.id.c_str());
    o.index = l.index;
    o.width = l.width;
    if (l.data.type() == typeid(double) || l.data.type() == typeid(float)) { o.kind = PCXB_DOUBLE; o.dval = l.data.convert<double>(); }
    else if (l.data.isNumber()) { o.kind = PCXB_SIZE; o.uval = l.data.convert<unsigned long>(); }
    else if (l.data.type() == typeid(std::string)) { o.kind = PCXB_STRING; std::snprintf(o.sval, sizeof(o.sval), "%s", l.data.convert<std::string>().c_str()); }
}

extern "C" {

const char *pcxb_last_error(void) { return g_err.c_str(); }
int pcxb_registry_has(const char *path) { return BlockRegistry::doesBlockExist(path) ? 1 : 0; }
size_t pcxb_registry_count(void) { return BlockRegistry::paths().size(); }
const char *pcxb_registry_path(size_t i)
{
    static thread_local std::string s;
    const auto p = BlockRegistry::paths();
    if (i >= p.size()) return "";
    s = p[i];
    return s.c_str();
}

int pcxb_make(const char *path, const char *dtype, size_t dimension, const char *sarg, size_t num_bins, int inverse, pcxb_block **out)
{
    return guarded([&] {
        const std::string p(path);
        std::vector<Object> args;
        const bool no_args = p == "/comms/fir_designer" || p == "/blocks/fir_designer";   // FIRDesigner::make(void)
        if (!no_args) args.push_back(Object(DType(std::string(dtype), dimension ? dimension : 1)));
        if (no_args) {
        } else if (p == "/comms/fir_filter" || p == "/blocks/fir_filter" || p == "/comms/arithmetic" || p == "/blocks/arithmetic" ||
                   p == "/comms/fm_demod_chain")
            args.push_back(Object(std::string(sarg ? sarg : "")));
        else if (p == "/comms/fft") { args.push_back(Object((unsigned long)num_bins)); args.push_back(Object(inverse != 0)); }
        std::unique_ptr<pcxb_block> b(new pcxb_block());
        b->blk.reset(BlockRegistry::make(p, args));
        if (!b->blk->inputs().empty() && b->blk->input(0)->_reserveSet) b->initialReserve = b->blk->input(0)->_reserve;
        *out = b.release();
    });
}
int pcxb_destroy(pcxb_block *b) { delete b; return PCX_OK; }
long pcxb_registry_arity(const char *path) { return path ? BlockRegistry::arity(path) : -1L; }
int pcxb_call_count(pcxb_block *b, size_t *count) { return guarded([&] { *count = b->blk->callArities().size(); }); }
int pcxb_call_name(pcxb_block *b, size_t i, char *out, size_t cap)
{
    return guarded([&] {
        const auto &m = b->blk->callArities();
        if (i >= m.size()) throw pcxfw::Exception("pcxb_call_name()", "index out of range");
        auto it = m.begin();
        std::advance(it, (long)i);
        std::snprintf(out, cap, "%s", it->first.c_str());
    });
}
long pcxb_call_arity(pcxb_block *b, const char *name)
{
    if (!b || !name) return -1L;
    const auto &m = b->blk->callArities();
    auto it = m.find(name);
    return it == m.end() ? -1L : (long)it->second;
}

int pcxb_call_double(pcxb_block *b, const char *name, double v) { return guarded([&] { b->blk->call(name, {Object(v)}); }); }
int pcxb_call_size(pcxb_block *b, const char *name, size_t v) { return guarded([&] { b->blk->call(name, {Object((unsigned long)v)}); }); }
int pcxb_call_bool(pcxb_block *b, const char *name, int v) { return guarded([&] { b->blk->call(name, {Object(v != 0)}); }); }
int pcxb_call_string(pcxb_block *b, const char *name, const char *v) { return guarded([&] { b->blk->call(name, {Object(std::string(v))}); }); }
int pcxb_call_taps(pcxb_block *b, const char *name, const double *taps, size_t n, int is_complex)
{
    return guarded([&] {
        if (is_complex) {
            std::vector<std::complex<double>> t(n);
            for (size_t i = 0; i < n; i++) t[i] = std::complex<double>(taps[2 * i], taps[2 * i + 1]);
            b->blk->call(name, {Object(t)});
        } else {
            b->blk->call(name, {Object(std::vector<double>(taps, taps + n))});
        }
    });
}
int pcxb_call_sizes(pcxb_block *b, const char *name, const size_t *v, size_t n)
{
    return guarded([&] { b->blk->call(name, {Object(std::vector<size_t>(v, v + n))}); });
}
int pcxb_get_sizes(pcxb_block *b, const char *name, size_t *out, size_t cap, size_t *n)
{
    return guarded([&] {
        const auto v = b->blk->call(name).convert<std::vector<size_t>>();
        *n = v.size();
        for (size_t i = 0; i < v.size() && i < cap; i++) out[i] = v[i];
    });
}
int pcxb_get_double(pcxb_block *b, const char *name, double *out) { return guarded([&] { *out = b->blk->call(name).convert<double>(); }); }
int pcxb_get_size(pcxb_block *b, const char *name, size_t *out) { return guarded([&] { *out = b->blk->call(name).convert<unsigned long>(); }); }
int pcxb_get_bool(pcxb_block *b, const char *name, int *out) { return guarded([&] { *out = b->blk->call(name).convert<bool>() ? 1 : 0; }); }
int pcxb_get_string(pcxb_block *b, const char *name, char *out, size_t cap)
{
    return guarded([&] { std::snprintf(out, cap, "%s", b->blk->call(name).convert<std::string>().c_str()); });
}
int pcxb_get_taps(pcxb_block *b, const char *name, double *out, size_t cap_doubles, size_t *n, int is_complex)
{
    return guarded([&] {
        const auto t = b->blk->call(name).convert<std::vector<std::complex<double>>>();
        *n = t.size();
        for (size_t i = 0; i < t.size(); i++) {
            if (is_complex) { if (2 * i + 1 < cap_doubles) { out[2 * i] = t[i].real(); out[2 * i + 1] = t[i].imag(); } }
            else if (i < cap_doubles) out[i] = t[i].real();
        }
    });
}

int pcxb_activate(pcxb_block *b) { return guarded([&] { b->blk->setActiveState(true); b->blk->activate(); }); }
int pcxb_deactivate(pcxb_block *b) { return guarded([&] { b->blk->deactivate(); b->blk->setActiveState(false); }); }
int pcxb_connect_signal(pcxb_block *src, const char *signal, pcxb_block *dst, const char *slot)
{
    return guarded([&] { src->blk->connectSignal(signal, dst->blk.get(), slot); });
}

int pcxb_port_dtype(pcxb_block *b, int is_output, char *name, size_t cap, size_t *dimension, size_t *bytes)
{
    return guarded([&] {
        const DType &dt = is_output ? b->blk->allOutputs().at(0)->dtype() : b->blk->allInputs().at(0)->dtype();   // first port, indexed or named
        std::snprintf(name, cap, "%s", dt.name().c_str());
        if (dimension) *dimension = dt.dimension();
        if (bytes) *bytes = dt.size();
    });
}
int pcxb_buffer_manager(pcxb_block *b, int is_output, char *name, size_t cap, size_t *buffer_size)
{
    return guarded([&] {
        auto m = is_output ? b->blk->getOutputBufferManager("", "") : b->blk->getInputBufferManager("", "");
        std::snprintf(name, cap, "%s", m ? m->name.c_str() : "");
        if (buffer_size) *buffer_size = m ? m->args.bufferSize : 0;
    });
}
int pcxb_acquire_buffer(pcxb_block *b, int is_output, size_t min_bytes, void **ptr, size_t *bytes, int *pinned)
{
    return guarded([&] {
        auto &slot = is_output ? b->outManager : b->inManager;
        if (!slot) slot = is_output ? b->blk->getOutputBufferManager("", "") : b->blk->getInputBufferManager("", "");
        if (!slot) slot = pcxfw::BufferManager::make("generic");     // a block without a preference: the scheduler's default slabs
        *ptr = slot->acquire(min_bytes, bytes);
        if (pinned) *pinned = slot->args.pinned ? 1 : 0;
    });
}
int pcxb_link_buffer(pcxb_block *src, pcxb_block *dst, size_t min_bytes, void **ptr, size_t *bytes, int *kind)
{
    // What the scheduler does for the edge src.output(0) -> dst.input(0): the downstream block is asked first, with the
    // upstream port's domain; if it has no preference the upstream block is asked, with the downstream port's domain [ext].
    return guarded([&] {
        const std::string srcDomain = src->blk->allOutputs().at(0)->domain(), dstDomain = dst->blk->allInputs().at(0)->domain();
        if (!dst->linkIn) {
            dst->linkIn = dst->blk->getInputBufferManager("", srcDomain);
            if (!dst->linkIn) dst->linkIn = src->blk->getOutputBufferManager("", dstDomain);
            if (!dst->linkIn) dst->linkIn = pcxfw::BufferManager::make("generic");
        }
        *ptr = dst->linkIn->acquire(min_bytes, bytes);
        if (kind) *kind = dst->linkIn->args.device ? 2 : dst->linkIn->args.pinned ? 1 : 0;
    });
}
int pcxb_initial_reserve(pcxb_block *b, size_t *reserve) { *reserve = b->initialReserve; return PCX_OK; }

// ---- the framework's circular buffer: one pageable shared-memory object mapped twice back to back ----
static std::mutex g_circ_mutex;
static std::map<void *, size_t> g_circ;      // base -> bytes of ONE mapping
int pcxb_circular_create(size_t bytes, void **base, size_t *actual)
{
    return guarded([&] {
        if (!base || !bytes) throw pcxfw::Exception("pcxb_circular_create()", "null argument");
        const size_t page = (size_t)sysconf(_SC_PAGESIZE);
        const size_t len = (bytes + page - 1) / page * page;
        const int fd = memfd_create("pcx-circular", MFD_CLOEXEC);
        if (fd < 0) throw pcxfw::Exception("pcxb_circular_create()", std::string("memfd_create: ") + std::strerror(errno));
        if (ftruncate(fd, (off_t)len) != 0) { const int e = errno; close(fd); throw pcxfw::Exception("pcxb_circular_create()", std::string("ftruncate: ") + std::strerror(e)); }
        // 2 x len of address space first, then the object twice on top of it
        void *span = mmap(nullptr, 2 * len, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (span == MAP_FAILED) { const int e = errno; close(fd); throw pcxfw::Exception("pcxb_circular_create()", std::string("mmap: ") + std::strerror(e)); }
        void *a = mmap(span, len, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_FIXED, fd, 0);
        void *b = mmap(static_cast<char *>(span) + len, len, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_FIXED, fd, 0);
        const int e = errno;
        close(fd);                                   // the mappings keep the object alive
        if (a == MAP_FAILED || b == MAP_FAILED) { munmap(span, 2 * len); throw pcxfw::Exception("pcxb_circular_create()", std::string("mmap of the object: ") + std::strerror(e)); }
        std::lock_guard<std::mutex> lk(g_circ_mutex);
        g_circ[span] = len;
        *base = span;
        if (actual) *actual = len;
    });
}
int pcxb_circular_destroy(void *base)
{
    return guarded([&] {
        std::lock_guard<std::mutex> lk(g_circ_mutex);
        auto it = g_circ.find(base);
        if (it == g_circ.end()) throw pcxfw::Exception("pcxb_circular_destroy()", "not a circular buffer of this runner");
        // the owner of the memory in front of munmap: whatever a block page-locked of it is let go of first (a registration
        // outliving its mapping would describe whatever is mapped there next)
        (void)pcx_host_release_range(base, 2 * it->second);
        munmap(base, 2 * it->second);
        g_circ.erase(it);
    });
}

int pcxb_work(pcxb_block *b, const void *in, size_t in_elems, const pcxb_label *labels, size_t nlabels, void *out,
              size_t out_elems, size_t *consumed, size_t *produced, size_t *reserve, pcxb_label *posted, size_t cap,
              size_t *nposted)
{
    return guarded([&] {
        InputPort *ip = b->blk->input(0);
        OutputPort *op = b->blk->output(0);
        ip->_buffer = BufferChunk::view(const_cast<void *>(in), in_elems * ip->dtype().size(), ip->dtype());
        op->_buffer = BufferChunk::view(out, out_elems * op->dtype().size(), op->dtype());
        ip->_labels.clear();
        for (size_t i = 0; i < nlabels; i++) ip->_labels.push_back(toLabel(labels[i]));
        ip->_consumed = 0; ip->_reserveSet = false; ip->_reserve = 0;
        op->_produced = 0; op->_posted.clear();
        WorkInfo &wi = b->blk->workInfoMutable();
        wi.minInElements = in_elems; wi.minOutElements = out_elems;
        wi.minElements = in_elems < out_elems ? in_elems : out_elems;
        b->blk->work();
        *consumed = ip->_consumed;
        *produced = op->_produced;
        *reserve = ip->_reserveSet ? ip->_reserve : std::numeric_limits<size_t>::max();
        // the framework forwards the labels that fell inside the consumed region [ext]
        std::vector<Label> inside;
        for (const auto &l : ip->_labels) if (l.index < ip->_consumed) inside.push_back(l);
        ip->_labels.swap(inside);
        if (!ip->_labels.empty()) b->blk->propagateLabels(ip);
        size_t n = 0;
        for (const auto &l : op->_posted) { if (n < cap && posted) fromLabel(l, posted[n]); n++; }
        if (nposted) *nposted = n;
    });
}

int pcxb_work_loop(pcxb_block *b, const void *in, size_t in_elems, void *out, size_t out_elems, size_t reps, double *seconds,
                   size_t *consumed, size_t *produced)
{
    return guarded([&] {
        InputPort *ip = b->blk->input(0);
        OutputPort *op = b->blk->output(0);
        ip->_labels.clear();
        WorkInfo &wi = b->blk->workInfoMutable();
        wi.minInElements = in_elems; wi.minOutElements = out_elems;
        wi.minElements = in_elems < out_elems ? in_elems : out_elems;
        const auto t0 = std::chrono::steady_clock::now();
        for (size_t r = 0; r < reps; r++) {
            ip->_buffer = BufferChunk::view(const_cast<void *>(in), in_elems * ip->dtype().size(), ip->dtype());
            op->_buffer = BufferChunk::view(out, out_elems * op->dtype().size(), op->dtype());
            ip->_consumed = 0; ip->_reserveSet = false; ip->_reserve = 0;
            op->_produced = 0; op->_posted.clear();
            b->blk->work();
        }
        if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (consumed) *consumed = ip->_consumed;
        if (produced) *produced = op->_produced;
    });
}

int pcxb_num_ports(pcxb_block *b, int is_output, size_t *count)
{
    return guarded([&] { *count = is_output ? b->blk->allOutputs().size() : b->blk->allInputs().size(); });
}
int pcxb_port_info(pcxb_block *b, int is_output, size_t i, char *name, size_t name_cap, char *dtype, size_t dtype_cap,
                   size_t *dimension, size_t *bytes, size_t *preloaded)
{
    return guarded([&] {
        DType dt;
        std::string nm;
        size_t pre = 0;
        if (is_output) {
            OutputPort *p = b->blk->allOutputs().at(i);
            dt = p->dtype(); nm = p->name();
        } else {
            InputPort *p = b->blk->allInputs().at(i);
            dt = p->dtype(); nm = p->name();
            for (const auto &c : p->_pushed) pre += c.length / dt.size();
        }
        if (name) std::snprintf(name, name_cap, "%s", nm.c_str());
        if (dtype) std::snprintf(dtype, dtype_cap, "%s", dt.name().c_str());
        if (dimension) *dimension = dt.dimension();
        if (bytes) *bytes = dt.size();
        if (preloaded) *preloaded = pre;
    });
}
int pcxb_work_ports(pcxb_block *b, size_t nin, const void *const *ins, const size_t *in_elems, size_t nout, void *const *outs,
                    const size_t *out_elems, size_t *consumed, size_t *produced)
{
    return guarded([&] {
        const auto ips = b->blk->allInputs();
        const auto ops = b->blk->allOutputs();
        if (ips.size() != nin || ops.size() != nout) throw pcxfw::Exception("pcxb_work_ports()", "port count mismatch");
        const size_t none = std::numeric_limits<size_t>::max();
        size_t minIn = none, minOut = none, minAll = none;
        for (size_t i = 0; i < nin; i++) {
            InputPort *ip = ips[i];
            ip->_buffer = BufferChunk::view(const_cast<void *>(ins[i]), in_elems[i] * ip->dtype().size(), ip->dtype());
            ip->_labels.clear();
            ip->_consumed = 0; ip->_reserveSet = false; ip->_reserve = 0;
            if (ip->index() >= 0) minIn = std::min(minIn, in_elems[i]);
            minAll = std::min(minAll, in_elems[i]);
        }
        for (size_t i = 0; i < nout; i++) {
            OutputPort *op = ops[i];
            op->_buffer = BufferChunk::view(outs[i], out_elems[i] * op->dtype().size(), op->dtype());
            op->_produced = 0; op->_posted.clear();
            if (op->index() >= 0) minOut = std::min(minOut, out_elems[i]);
            minAll = std::min(minAll, out_elems[i]);
        }
        WorkInfo &wi = b->blk->workInfoMutable();
        wi.minInElements = minIn == none ? 0 : minIn;
        wi.minOutElements = minOut == none ? 0 : minOut;
        // minElements spans the indexed ports only; a side without indexed ports does not bound it
        const size_t me = std::min(minIn, minOut);
        wi.minElements = me == none ? 0 : me;
        wi.minAllElements = minAll == none ? 0 : minAll;
        b->blk->work();
        for (size_t i = 0; i < nin; i++) consumed[i] = ips[i]->_consumed;
        for (size_t i = 0; i < nout; i++) produced[i] = ops[i]->_produced;
    });
}

}  // extern "C"
