// comms_blocks.cpp -- the MI355X-backed /comms blocks: same registry paths, factory
// arguments, setters/getters, port types, buffer-manager requests, label behaviour and
// consume/produce accounting as the reference blocks; the arithmetic runs in libpcx_hip.so
// through the C ABI (include/pcx.h).
//
//   /comms/fir_filter (+ /blocks/fir_filter)   filter/FIRFilter.cpp:98-389
//   /comms/fft                                 fft/FFT.cpp:39-95
//   /comms/freq_demod                          demod/FreqDemod.cpp:33-95
//   /comms/rotate                              math/Rotate.cpp:47-160
//   /comms/scale                               math/Scale.cpp:46-160
//   /comms/abs                                 math/Abs.cpp:66-125
//   /comms/conjugate                           math/Conjugate.cpp:61-119
//   /comms/angle  (SURVEY 8f "next")           math/Angle.cpp:50-110
//   /comms/arithmetic (+ /blocks/arithmetic)   math/Arithmetic.cpp:150-305      (8f "next")
//   /comms/split_complex, /comms/combine_complex   utility/{Split,Combine}Complex.cpp  (8f "next")
//   (/comms/fir_designer, host-side only, lives in fir_designer.cpp)
//
// The reference instantiates one C++ template per element type; here a block carries a
// pcx_scalar code instead and the type dispatch happens behind the ABI, so one class per
// block serves the whole factory matrix.  There is no CPU path: a (type, size) the device
// library does not implement surfaces as an exception from the factory or from work().
//
// Built against pcx_framework.hpp: PothosCore when -DPCX_WITH_POTHOS, the bundled runtime
// otherwise (tests, runner ABI).
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstring>
#include <deque>
#include <string>
#include <vector>

#include "pcx.h"
#include "pcx_framework.hpp"

using pcxfw::Block;
using pcxfw::BufferChunk;
using pcxfw::DType;
using pcxfw::InvalidArgumentException;
using pcxfw::Label;

namespace {

// DType element name -> (pcx_scalar, complex?)
bool parseElemType(const DType &dt, int &scalar, bool &cplx)
{
    std::string n = DType::fromDType(dt, 1).name();
    cplx = n.compare(0, 8, "complex_") == 0;
    if (cplx) n = n.substr(8);
    if (n == "float64") scalar = PCX_F64;
    else if (n == "float32") scalar = PCX_F32;
    else if (n == "int64") scalar = PCX_I64;
    else if (n == "int32") scalar = PCX_I32;
    else if (n == "int16") scalar = PCX_I16;
    else if (n == "int8") scalar = PCX_I8;
    else return false;
    return true;
}
// the arithmetic factory also takes the unsigned types (Arithmetic.cpp:288-291)
bool parseArithType(const DType &dt, int &scalar, bool &cplx)
{
    if (parseElemType(dt, scalar, cplx)) return true;
    std::string n = DType::fromDType(dt, 1).name();
    if (cplx) n = n.substr(8);
    if (n == "uint64") scalar = PCX_U64;
    else if (n == "uint32") scalar = PCX_U32;
    else if (n == "uint16") scalar = PCX_U16;
    else if (n == "uint8") scalar = PCX_U8;
    else return false;
    return true;
}
DType complexOf(const DType &dt)
{
    return DType("complex_" + DType::fromDType(dt, 1).name(), dt.dimension());
}
DType realOf(const DType &dt)
{
    std::string n = DType::fromDType(dt, 1).name();
    if (n.compare(0, 8, "complex_") == 0) n = n.substr(8);
    return DType(n, dt.dimension());
}
// ABI status -> the exception type the reference would throw at that point
void check(int rc, const std::string &where)
{
    if (rc == PCX_OK) return;
    const std::string msg = pcx_last_error();
    if (rc == PCX_ERR_ARG) throw InvalidArgumentException(where, msg);
    throw pcxfw::Exception(where, msg);
}

// EXTENSION (not in the reference) of the integer FIR / Rotate / Scale blocks: setQFormat("HALF_Q,TRUNCATE,FLOOR") names the reading of
// Pothos::Util::floatToQ / fromQ the block computes with (include/pcx.h, pcx_qformat: PothosCore's QFormat.hpp is not part of the
// reference tree and the reference's tests leave twelve readings standing).  Three comma-separated words -- fractional bits HALF_Q |
// HALF_ELEM; floatToQ TRUNCATE | NEAREST; fromQ FLOOR | TOWARD_ZERO | ROUND -- or "DEFAULT" for the process-wide reading.
struct BlockQFormat {
    bool set = false;          // false: the process-wide reading (a NULL pcx_qformat)
    pcx_qformat q{0, 0, 0};
    std::string spec = "DEFAULT";
    const pcx_qformat *ptr() const { return set ? &q : nullptr; }
    void parse(const std::string &text, const std::string &where)
    {
        if (text == "DEFAULT" || text.empty()) { set = false; spec = "DEFAULT"; return; }
        std::vector<std::string> w;
        size_t a = 0;
        for (;;) {
            const size_t b = text.find(',', a);
            std::string t = text.substr(a, b == std::string::npos ? std::string::npos : b - a);
            while (!t.empty() && t.front() == ' ') t.erase(t.begin());
            while (!t.empty() && t.back() == ' ') t.pop_back();
            w.push_back(t);
            if (b == std::string::npos) break;
            a = b + 1;
        }
        pcx_qformat v{0, 0, 0};
        bool ok = w.size() == 3;
        if (ok) {
            if (w[0] == "HALF_Q") v.frac = PCX_Q_FRAC_HALF_Q; else if (w[0] == "HALF_ELEM") v.frac = PCX_Q_FRAC_HALF_ELEM; else ok = false;
            if (w[1] == "TRUNCATE") v.float_to_q = PCX_Q_TRUNCATE; else if (w[1] == "NEAREST") v.float_to_q = PCX_Q_NEAREST; else ok = false;
            if (w[2] == "FLOOR") v.from_q = PCX_Q_FLOOR; else if (w[2] == "TOWARD_ZERO") v.from_q = PCX_Q_TOWARD_ZERO;
            else if (w[2] == "ROUND") v.from_q = PCX_Q_ROUND; else ok = false;
        }
        if (!ok) throw InvalidArgumentException(where + "(" + text + ")", "expected HALF_Q|HALF_ELEM,TRUNCATE|NEAREST,FLOOR|TOWARD_ZERO|ROUND or DEFAULT");
        q = v; set = true;
        spec = w[0] + "," + w[1] + "," + w[2];
    }
};

/***********************************************************************
 * Port buffers of a device-backed block: page-locked slabs
 *
 * A Pothos block may hand the scheduler its own buffer managers (the reference does: FIRFilter.cpp:196-199 asks for a
 * circular input buffer, FFT.cpp:54-59 for frame-sized output slabs).  Every block of this module asks for PINNED slabs
 * on both sides: the C ABI then runs its kernels directly on the port buffers over PCIe instead of staging them
 * (include/pcx.h; 43 GB/s each way against 28 staged, tools/pcie_lab.hip) -- a plain work() loop gets that without
 * the topology doing anything.
 *
 * How large: every work() call costs about 60 us of launch, synchronisation and link ramp-up whatever it carries, so the
 * slab size decides what share of the link a block reaches (complex_float32 FIR, 255 taps, examples/c_block_path.c and
 * bench.py secondary.host_path): 8 MiB slabs (1 Mi samples per call) 4.1 Gsamples/s = 0.57 of the PCIe roof at 0.26 ms per
 * call; 32 MiB (4 Mi samples) 5.0 = 0.70 at 0.84 ms; 64 MiB (8 Mi) 5.3 = 0.74 at 1.6 ms; 128 MiB (16 Mi) 5.5 = 0.76 at 3.1 ms.  The
 * default is 64 MiB: the fixed cost is 6 % of such a call, and the curve is flat from there.  It is a SETTING of every block (setPortSlabBytes, an initializer: the framework asks for the managers when
 * the topology is committed) -- a latency-bound topology takes 1-8 MiB, a throughput-bound one 128.  Page-locked memory per
 * block: slab size x the framework's buffers per port (4 by default) x the ports that bring their own manager.
 **********************************************************************/
constexpr size_t kPortSlabBytes = 64u << 20;
constexpr size_t kPortSlabMin = 64u << 10, kPortSlabMax = 1u << 30;
#ifdef PCX_WITH_POTHOS
// Pothos build: a pool of page-locked slabs behind Pothos::BufferManager's PUBLIC interface (init / empty / pop / push over
// setFrontBuffer) -- the pool logic of the framework's own "generic" manager, which lives in PothosCore's library
// (lib/Framework/Builtin/GenericBufferManager.cpp) and is not in the installed headers, so it cannot be a base class here
// (ADVICE r2).  Slabs come from pcx_host_alloc: the C ABI's host-pointer entry points then run their kernels on the port
// buffers in place over PCIe instead of staging them.
// [PothosCore is not installable in this image: this branch is parsed and type-checked against a declaration-only header set
// (tests/pothos_decl, tests/test_pothos_syntax_cpu.py) and has never been linked or run -- INTEGRATION.md 2.]
class PinnedBufferManager : public Pothos::BufferManager, public std::enable_shared_from_this<PinnedBufferManager> {
public:
    explicit PinnedBufferManager(size_t slabBytes) : _slabBytes(slabBytes) {}
    void init(const Pothos::BufferManagerArgs &args) override
    {
        Pothos::BufferManager::init(args);
        const size_t bytes = std::max(args.bufferSize, _slabBytes);
        for (size_t i = 0; i < args.numBuffers; i++) {
            void *p = nullptr;
            if (pcx_host_alloc(&p, bytes) != PCX_OK) throw Pothos::Exception("PinnedBufferManager::init()", pcx_last_error());
            auto keep = std::shared_ptr<void>(p, [](void *q) { (void)pcx_host_free(q); });
            Pothos::SharedBuffer sb(size_t(p), bytes, keep);
            Pothos::ManagedBuffer mb;
            mb.reset(this->shared_from_this(), sb, i);
            this->push(mb);
        }
    }
    bool empty() const override { return _ready.empty(); }
    void pop(const size_t) override
    {
        _ready.pop_front();     // the consumer holds the chunk; the slab comes back through push() when its last reference goes
        if (_ready.empty()) this->setFrontBuffer(Pothos::BufferChunk::null());
        else this->setFrontBuffer(Pothos::BufferChunk(_ready.front()));
    }
    void push(const Pothos::ManagedBuffer &buff) override
    {
        if (_ready.empty()) this->setFrontBuffer(Pothos::BufferChunk(buff));
        _ready.push_back(buff);
    }

private:
    size_t _slabBytes;
    std::deque<Pothos::ManagedBuffer> _ready;
};
// "circular": the FIR's sliding window needs its K-1 history contiguous in front of new samples, which the framework's own
// circular manager provides by mapping its memory twice (FIRFilter.cpp:196-199 asks for exactly this).  That memory is pageable
// when the framework hands it over; the FIR block page-locks it where it lies the first time work() sees it (FIRFilter::pageLock
// below, pcx_host_register_mapping), after which the kernel reads it in place like any pinned slab.  Anything else: page-locked slabs.
static pcxfw::BufferManager::Sptr pinnedManager(const std::string &name, size_t slabBytes)
{
    if (name == "circular") return Pothos::BufferManager::make("circular");
    return pcxfw::BufferManager::Sptr(new PinnedBufferManager(slabBytes));   // the scheduler calls init() with its own args
}
#else
static pcxfw::BufferManager::Sptr pinnedManager(const std::string &name, size_t slabBytes)
{
    pcxfw::BufferManagerArgs args;
    args.bufferSize = slabBytes;
    args.numBuffers = 4;
    args.pinned = true;
    return pcxfw::BufferManager::make(name, args);
}
// slabs in device memory, for an edge whose other end is a block of this module too
static pcxfw::BufferManager::Sptr deviceManager(const std::string &name, size_t slabBytes)
{
    pcxfw::BufferManagerArgs args;
    args.bufferSize = slabBytes;
    args.numBuffers = 4;
    args.device = true;
    return pcxfw::BufferManager::make(name, args);
}
#endif
// The port domain of this module's blocks (Pothos: the third argument of setupInput / setupOutput, handed to the OTHER end's
// get{Input,Output}BufferManager [ext]).  When both ends of an edge carry it, the edge's buffers live in device memory: the
// upstream block's output manager hands out HBM slabs, the downstream block asks for nothing -- or, the FIR, for its circular
// buffer in HBM -- and the host-pointer entry points of the C ABI run in place on them (they recognise device pointers exactly
// as they recognise page-locked ones).  A Rotate -> FIR -> FreqDemod topology of three separate blocks then crosses PCIe
// once in and once out instead of three times each way.  Any other domain ("" = a host block) gets page-locked host slabs.
// BUNDLED RUNTIME ONLY.  Under -DPCX_WITH_POTHOS every port gets page-locked HOST slabs, whatever the domain at the other end: a
// Pothos::BufferChunk that points into HBM would be dereferenced on the CPU by the framework itself -- the input port's accumulator
// memcpy's queued chunks together whenever a reserve exceeds the front buffer (the FIR's M + K - 1, the FFT's frame), the topology
// inserts a copier block where both ends of an edge bring a manager (any output meeting the FIR's circular input), and any foreign
// block connected to the same output reads the bytes -- and nothing short of changing PothosCore can tell those "this is device
// memory".  An unchanged three-block chain therefore pays PCIe per edge inside Pothos (1.1-1.6 Gsamples/s against 3.8-5.7 for the
// fused /comms/fm_demod_chain block, tools/chain_path.py): the fused block is the remedy, not device pointers in
// the framework's hands.
static const char *const kDomain = "pcx-hip";

// the calling thread's current device for the length of a scope (the C ABI binds a handle to the device current when it is
// CREATED and runs the stateless maps on the device current when they are CALLED, include/pcx.h)
class OnDevice {
public:
    explicit OnDevice(int device, const char *where = "DeviceBlock") : _prev(-1)
    {
        int cur = -1;
        if (device < 0 || pcx_get_device(&cur) != PCX_OK || cur == device) return;
        check(pcx_set_device(device), where);
        _prev = cur;
    }
    ~OnDevice() { if (_prev >= 0) (void)pcx_set_device(_prev); }
    OnDevice(const OnDevice &) = delete;
    OnDevice &operator=(const OnDevice &) = delete;

private:
    int _prev;
};

/***********************************************************************
 * What every block of this module has on top of its reference counterpart: the GPU it lives on and the size of its port slabs.
 *
 *   setDevice(device) / getDevice()   EXTENSION.  The ordinal of the GPU that carries the block (the reference has no devices;
 *       SURVEY.md 5 plans the knob).  A block is born on the device current on the thread that runs its factory -- inside a Pothos
 *       process that is device 0 -- and setDevice moves it: device handles are created again on the named device (also when that is
 *       where the block is) and taps, phase, decimation ... pushed again; CARRIED state starts over as after activate() (FreqDemod's
 *       previous sample, a FIR's position inside a burst is kept by the block itself).  The stateless
 *       maps simply make the device current around their call.  work() never changes the calling thread's current device for longer
 *       than the call.  Eight independent chains of one Pothos process on the eight GPUs of a node: setDevice(0 .. 7), nothing else
 *       (the streams are independent: "split on frame / element boundaries, no halo").
 *   setPortSlabBytes(bytes) / getPortSlabBytes()   EXTENSION.  The size of the page-locked slabs the block's buffer managers hand
 *       out (kPortSlabBytes above); takes effect when the topology asks for the managers, i.e. it is an initializer.
 **********************************************************************/
class DeviceBlock : public Block {
public:
    DeviceBlock() : _device(-1), _slabBytes(kPortSlabBytes)
    {
        int cur = -1;
        if (pcx_get_device(&cur) == PCX_OK) _device = cur;     // (no device in the process: the first handle's create says so)
        this->registerCall(this, "setDevice", &DeviceBlock::setDevice);
        this->registerCall(this, "getDevice", &DeviceBlock::getDevice);
        this->registerCall(this, "setPortSlabBytes", &DeviceBlock::setPortSlabBytes);
        this->registerCall(this, "getPortSlabBytes", &DeviceBlock::getPortSlabBytes);
    }
    virtual ~DeviceBlock() {}

    void setDevice(const size_t device)
    {
        int n = 0;
        check(pcx_device_count(&n), "DeviceBlock::setDevice()");
        if (device >= (size_t)n)
            throw InvalidArgumentException("DeviceBlock::setDevice(" + std::to_string(device) + ")", "the process sees " + std::to_string(n) + " device(s)");
        // (also for the device the block is on already: "create the handles again here" is what the call means, whatever here is --
        // which is how one GPU exercises the path)
        const int from = _device;
        _device = (int)device;
        try {
            OnDevice on(_device, "DeviceBlock::setDevice()");
            this->rebind();
        } catch (...) {
            // getDevice() reports where the block IS: back to where it was (its handles there are still alive when rebind threw early)
            _device = from;
            try { OnDevice on(_device, "DeviceBlock::setDevice()"); this->rebind(); } catch (...) {}
            throw;
        }
    }
    size_t getDevice() const { return _device < 0 ? 0 : (size_t)_device; }
    void setPortSlabBytes(const size_t bytes)
    {
        if (bytes < kPortSlabMin || bytes > kPortSlabMax)
            throw InvalidArgumentException("DeviceBlock::setPortSlabBytes(" + std::to_string(bytes) + ")", "64 KiB ... 1 GiB");
        _slabBytes = bytes;
    }
    size_t getPortSlabBytes() const { return _slabBytes; }

    // every port of a device block carries the module's domain
    pcxfw::InputPort *setupInput(size_t i, const DType &dt = DType()) { return Block::setupInput(i, dt, kDomain); }
    pcxfw::OutputPort *setupOutput(size_t i, const DType &dt = DType()) { return Block::setupOutput(i, dt, kDomain); }
    // an explicit domain stays what the block asked for (Arithmetic's unique one, Arithmetic.cpp:136: buffer forwarding)
    pcxfw::OutputPort *setupOutput(size_t i, const DType &dt, const std::string &domain) { return Block::setupOutput(i, dt, domain); }
    pcxfw::InputPort *setupInput(const std::string &name, const DType &dt = DType()) { return Block::setupInput(name, dt, kDomain); }
    pcxfw::OutputPort *setupOutput(const std::string &name, const DType &dt = DType()) { return Block::setupOutput(name, dt, kDomain); }
#ifndef PCX_WITH_POTHOS
    pcxfw::BufferManager::Sptr getInputBufferManager(const std::string &, const std::string &domain)
    {
        if (domain == kDomain) return pcxfw::BufferManager::Sptr();     // the upstream block of this module provides device slabs
        return pinnedManager("generic", _slabBytes);
    }
    pcxfw::BufferManager::Sptr getOutputBufferManager(const std::string &, const std::string &domain)
    {
        OnDevice on(_device);       // (device slabs are allocated on the block's device)
        return domain == kDomain ? deviceManager("generic", _slabBytes) : pinnedManager("generic", _slabBytes);
    }
#else
    // Inside Pothos every edge is page-locked HOST memory.  An input port whose upstream block is one of this module's (its port domain
    // is ours) ABDICATES -- a null manager: that block's output manager already hands out page-locked slabs, and two custom managers on
    // one edge make the topology insert a copier block, a CPU memcpy per buffer [ext: Topology commit, domain / manager rectification].
    pcxfw::BufferManager::Sptr getInputBufferManager(const std::string &, const std::string &domain)
    {
        return domain == kDomain ? pcxfw::BufferManager::Sptr() : pinnedManager("generic", _slabBytes);
    }
    pcxfw::BufferManager::Sptr getOutputBufferManager(const std::string &, const std::string &) { return pinnedManager("generic", _slabBytes); }
#endif

protected:
    // create the block's device handles again on the (now current) device and push its settings; the stateless maps have none
    virtual void rebind() {}
    int _device;            // the ordinal the block lives on
    size_t _slabBytes;
};

/***********************************************************************
 * |PothosDoc FIR Filter
 *
 * A finite-impulse-response filter on the GPU: the element stream on input port 0 is convolved with the taps
 * and leaves on output port 0, optionally resampled by interpolation / decimation (a polyphase filter: only the
 * outputs that survive the decimator are computed).  Same results as the CPU block of PothosComms: floating-point
 * streams within 1e-5 of the output scale, integer streams bit for bit.
 *
 * <h2>Bursts</h2>
 * A burst that is marked inside the stream is filtered as a unit: its tail is flushed with zeros and no sample of the
 * following burst enters the sums.  Mark a burst either with a label on its first element whose data is the burst's
 * length in elements, or with a label on its last element.
 *
 * <h2>Device</h2>
 * The filter runs as a frequency-domain overlap-save kernel from a few dozen taps up and as a time-domain kernel below;
 * "Kernel" overrides the choice.  The block page-locks its input buffer where the framework allocated it and the kernels
 * read and write the port buffers over PCIe in place.
 *
 * |category /Filter
 * |keywords fir filter taps highpass lowpass bandpass gpu hip
 * |alias /blocks/fir_filter
 *
 * |param dtype[Data Type] Element type of the input and of the output stream.
 * |widget DTypeChooser(float=1,cfloat=1,int=1,cint=1)
 * |default "complex_float32"
 * |preview disable
 *
 * |param tapsType[Taps Type] Real taps, or complex taps (complex streams only).
 * |option [Real] "REAL"
 * |option [Complex] "COMPLEX"
 *
 * |param decim[Decimation] Keep one output in this many.
 * |default 1
 * |widget SpinBox(minimum=1)
 *
 * |param interp[Interpolation] Outputs per input element.
 * |default 1
 * |widget SpinBox(minimum=1)
 *
 * |param taps The filter taps.
 * Type or paste them here, or leave the default and connect a FIR Designer's "tapsChanged" signal to the setTaps slot.
 * |default [1.0]
 *
 * |param waitTaps[Wait Taps] Hold the stream back until taps have arrived through setTaps().
 * For filters whose taps only ever come from a designer block at run time.
 * |default false
 * |preview valid
 * |option [Enabled] true
 * |option [Disabled] false
 *
 * |param frameStartId[Frame Start ID] ID of the label that marks the first element of a burst and carries its length.
 * Empty: no burst handling by start label.
 * |default ""
 * |widget StringEntry()
 * |preview valid
 * |tab Labels
 *
 * |param frameEndId[Frame End ID] ID of the label that marks the last element of a burst.
 * Empty: no burst handling by end label.
 * |default ""
 * |widget StringEntry()
 * |preview valid
 * |tab Labels
 *
 * |param kernel[Kernel] Which device kernel family serves the filter.
 * AUTO picks by tap count and type; EXACT is the time-domain sum in the CPU block's operation order (bit-identical floats).
 * |default "AUTO"
 * |option [Auto] "AUTO"
 * |option [Overlap-save FFT] "OLS_FFT"
 * |option [Time domain] "DIRECT"
 * |option [Time domain, CPU order] "EXACT"
 * |preview disable
 * |tab Device
 *
 * |param qformat[Q Format] Fixed-point reading of an integer filter: fractional bits, tap rounding, output rounding.
 * "DEFAULT", or three words such as "HALF_Q,TRUNCATE,FLOOR".
 * |default "DEFAULT"
 * |widget StringEntry()
 * |preview disable
 * |tab Device
 *
 * |param device[Device] Ordinal of the GPU that carries the block.
 * |default 0
 * |widget SpinBox(minimum=0)
 * |preview disable
 * |tab Device
 *
 * |param devices[Shard Devices] GPUs that share ONE stream, in stream order: each call is cut into as many contiguous shards,
 * the tap-length halo travels between neighbours.  Empty: a single device (the one above); one ordinal: that device.
 * complex_float32 without resampling.
 * |default []
 * |preview disable
 * |tab Device
 *
 * |param portSlabBytes[Port Slab Bytes] Size of the page-locked port buffers the block asks the framework for.
 * Larger slabs carry more samples per call (throughput), smaller ones return sooner (latency).
 * |default 67108864
 * |units bytes
 * |preview disable
 * |tab Device
 *
 * |factory /comms/fir_filter(dtype, tapsType)
 * |initializer setPortSlabBytes(portSlabBytes)
 * |initializer setDevice(device)
 * |setter setTaps(taps)
 * |setter setDecimation(decim)
 * |setter setInterpolation(interp)
 * |setter setWaitTaps(waitTaps)
 * |setter setFrameStartId(frameStartId)
 * |setter setFrameEndId(frameEndId)
 * |setter setKernel(kernel)
 * |setter setQFormat(qformat)
 * |setter setDevices(devices)
 **********************************************************************/
class FIRFilter : public DeviceBlock {
public:
    FIRFilter(const DType &dtype, int scalar, bool cplx, bool complexTaps)
        : _scalar(scalar), _cplx(cplx), _complexTaps(complexTaps), _elemBytes(dtype.size()), M(1), L(1), K(1), _inputRequire(1),
          _waitTapsMode(false), _waitTapsArmed(false), _eobSampsLeft(0), _dtype(dtype), _h(nullptr)
    {
        check(pcx_fir_create(scalar, cplx ? 1 : 0, complexTaps ? 1 : 0, &_h), "FIRFilterFactory(" + dtype.toString() + ")");
        this->setupInput(0, dtype);
        this->setupOutput(0, dtype);
        if (complexTaps) {
            this->registerCall(this, "setTaps", &FIRFilter::setTapsComplex);
            this->registerCall(this, "getTaps", &FIRFilter::getTapsComplex);
        } else {
            this->registerCall(this, "setTaps", &FIRFilter::setTapsReal);
            this->registerCall(this, "getTaps", &FIRFilter::getTapsReal);
        }
        this->registerCall(this, PCX_FCN_TUPLE(FIRFilter, setDecimation));
        this->registerCall(this, PCX_FCN_TUPLE(FIRFilter, getDecimation));
        this->registerCall(this, PCX_FCN_TUPLE(FIRFilter, setInterpolation));
        this->registerCall(this, PCX_FCN_TUPLE(FIRFilter, getInterpolation));
        this->registerCall(this, PCX_FCN_TUPLE(FIRFilter, setWaitTaps));
        this->registerCall(this, PCX_FCN_TUPLE(FIRFilter, getWaitTaps));
        this->registerCall(this, PCX_FCN_TUPLE(FIRFilter, setFrameStartId));
        this->registerCall(this, PCX_FCN_TUPLE(FIRFilter, getFrameStartId));
        this->registerCall(this, PCX_FCN_TUPLE(FIRFilter, setFrameEndId));
        this->registerCall(this, PCX_FCN_TUPLE(FIRFilter, getFrameEndId));
        // extension (not in the reference): which device kernel family serves the filter
        this->registerCall(this, PCX_FCN_TUPLE(FIRFilter, setKernel));
        this->registerCall(this, PCX_FCN_TUPLE(FIRFilter, getKernel));
        this->registerCall(this, PCX_FCN_TUPLE(FIRFilter, setQFormat));
        this->registerCall(this, PCX_FCN_TUPLE(FIRFilter, getQFormat));
        // extension: the block's stream spread over several devices of the node from inside work() (pcx_shard_*, pcx.h)
        this->registerCall(this, PCX_FCN_TUPLE(FIRFilter, setDevices));
        this->registerCall(this, PCX_FCN_TUPLE(FIRFilter, getDevices));
        this->registerCall(this, PCX_FCN_TUPLE(FIRFilter, getShardPasses));
        // initial update: a single unit tap (reference ctor)
        _taps.assign(1, std::complex<double>(1.0, 0.0));
        this->pushTaps();
    }
    ~FIRFilter()
    {
        if (_sh) pcx_shard_destroy(_sh);
        pcx_fir_destroy(_h);
        this->unlockAll();
    }

    void setTapsReal(const std::vector<double> &taps)
    {
        if (taps.empty()) throw InvalidArgumentException("FIRFilter::setTaps()", "taps cannot be empty");
        _taps.assign(taps.begin(), taps.end());
        _waitTapsArmed = false;  // got taps
        this->pushTaps();
    }
    void setTapsComplex(const std::vector<std::complex<double>> &taps)
    {
        if (taps.empty()) throw InvalidArgumentException("FIRFilter::setTaps()", "taps cannot be empty");
        _taps = taps;
        _waitTapsArmed = false;
        this->pushTaps();
    }
    std::vector<double> getTapsReal() const
    {
        std::vector<double> t(_taps.size());
        for (size_t i = 0; i < t.size(); i++) t[i] = _taps[i].real();
        return t;
    }
    std::vector<std::complex<double>> getTapsComplex() const { return _taps; }

    void setDecimation(const size_t decim)
    {
        if (decim == 0) throw InvalidArgumentException("FIRFilter::setDecimation()", "decimation cannot be 0");
        M = decim;
        check(pcx_fir_set_decimation(_h, decim), "FIRFilter::setDecimation()");
        this->refreshGeometry();
    }
    size_t getDecimation() const { return M; }
    void setInterpolation(const size_t interp)
    {
        if (interp == 0) throw InvalidArgumentException("FIRFilter::setInterpolation()", "interpolation cannot be 0");
        L = interp;
        check(pcx_fir_set_interpolation(_h, interp), "FIRFilter::setInterpolation()");
        this->refreshGeometry();
    }
    size_t getInterpolation() const { return L; }
    // "AUTO" (default), "OLS_FFT" (frequency domain), "DIRECT" (time domain, FMA), "EXACT" (time domain in the
    // reference's operation order: bit-identical floats, and the reference's locality for Inf/NaN samples)
    void setKernel(const std::string &name)
    {
        check(pcx_fir_set_algo(_h, kernelCode(name)), "FIRFilter::setKernel(" + name + ")");
        _kernel = name;
    }
    std::string getKernel() const { return _kernel; }
    // the floatToQ<QTapsType> / fromQ<OutType> reading of an integer filter (FIRFilter.cpp:300,348; BlockQFormat above)
    void setQFormat(const std::string &spec)
    {
        _qformat.parse(spec, "FIRFilter::setQFormat");
        check(pcx_fir_set_qformat(_h, _qformat.ptr()), "FIRFilter::setQFormat(" + spec + ")");
    }
    std::string getQFormat() const { return _qformat.spec; }
    void setWaitTaps(const bool waitTaps) { _waitTapsMode = waitTaps; }
    bool getWaitTaps() const { return _waitTapsMode; }
    void setFrameStartId(std::string id) { _frameStartId = id; }
    std::string getFrameStartId() const { return _frameStartId; }
    void setFrameEndId(std::string id) { _frameEndId = id; }
    std::string getFrameEndId() const { return _frameEndId; }

    // EXTENSION (not in the reference): one stream over several GPUs from this one block.  devices = the ordinals that carry a
    // shard each, in stream order.  An EMPTY list (the default) is the single-device filter on the block's device (setDevice); ONE
    // ordinal is setDevice(that ordinal) -- the block moves there; two or more: each work() call splits what the port holds into
    // devices.size() contiguous shards, moves the K-1-sample halo between neighbouring devices (RCCL send/recv; peer copies when an
    // ordinal repeats, i.e. several shards on one device -- a rehearsal) and filters every shard in ONE gated launch (pcx_shard_step,
    // DESIGN.md 6).  complex_float32 with M = L = 1 outside burst mode; anything else, and calls that bring less than a shard's worth
    // per device, run on the single-device handle: the totals of consume / produce are the reference's either way.
    void setDevices(const std::vector<size_t> &devices)
    {
        // validate first: getDevices() reports the layout that is IN EFFECT, also after a call that threw
        if (devices.size() >= 2 && !(_dtype == DType("complex_float32")))
            throw InvalidArgumentException("FIRFilter::setDevices()", "a sharded stream is complex_float32");
        if (devices.size() == 1) this->setDevice(devices[0]);      // (throws for an ordinal the process does not have)
        pcx_shard *sh = nullptr;
        if (devices.size() >= 2) {
            std::vector<int> d(devices.begin(), devices.end());
            std::vector<int> sorted(d);
            std::sort(sorted.begin(), sorted.end());
            const bool distinct = std::adjacent_find(sorted.begin(), sorted.end()) == sorted.end();
            check(pcx_shard_create((int)d.size(), d.data(), distinct ? PCX_SHARD_RCCL : PCX_SHARD_PEER_COPY, &sh), "FIRFilter::setDevices()");
        }
        if (_sh) { const int rc = pcx_shard_destroy(_sh); _sh = nullptr; if (rc != PCX_OK) { (void)pcx_shard_destroy(sh); check(rc, "FIRFilter::setDevices()"); } }
        _sh = sh;
        _devices = devices;
        _shardC = 0;
        _shardShort = 0;
        if (_sh) this->pushTaps();
    }
    // the layout in effect: the shard devices, or the one device of a filter that was placed with setDevices({d})
    std::vector<size_t> getDevices() const { return _devices.size() == 1 ? std::vector<size_t>(1, this->getDevice()) : _devices; }
    size_t getShardPasses() const { return _shardPasses; }

    // the sliding window needs its K-1 history contiguous in front of new samples -- in HBM when the upstream block is one of
    // this module's (kDomain), in page-locked host memory otherwise
    pcxfw::BufferManager::Sptr getInputBufferManager(const std::string &, const std::string &domain)
    {
#ifndef PCX_WITH_POTHOS
        if (domain == kDomain) { OnDevice on(_device); return deviceManager("circular", _slabBytes); }
#else
        (void)domain;
#endif
        return pinnedManager("circular", _slabBytes);
    }

    void activate()
    {
        _waitTapsArmed = _waitTapsMode;
        _eobSampsLeft = 0;
        // what this block page-locked belongs to MAPPINGS, and a topology that was re-committed has re-allocated its buffers since: a
        // range whose mapping is gone is let go of (a new buffer at the same address is pageable memory, whatever the old entry said)
        for (size_t i = 0; i < _locked.size();) {
            int alive = 0;
            if (pcx_host_mapping_alive(_locked[i].first, &alive) == PCX_OK && alive) { i++; continue; }
            (void)pcx_host_release_range(_locked[i].first, _locked[i].second);
            _locked.erase(_locked.begin() + i);
        }
        _unlockable.clear();
    }
    // the framework may unmap the port buffers once the block is inactive: nothing stays page-locked behind its back.  (Counted:
    // another block of the module that runs on the same buffer keeps it locked, pcx_host_unregister.)
    void deactivate() { this->unlockAll(); }

    void work()
    {
        if (_waitTapsArmed) return;
        auto inPort = this->input(0);
        auto outPort = this->output(0);
        size_t avail = inPort->elements();
        if (avail == 0) return;

        // burst bookkeeping: where does the current frame end?
        if (_eobSampsLeft == 0) {
            for (const auto &label : inPort->labels()) {
                if (!_frameStartId.empty() && label.id == _frameStartId && label.data.canConvert(typeid(size_t))) {
                    _eobSampsLeft = label.index + label.data.template convert<size_t>() * label.width;
                    break;
                }
                if (!_frameEndId.empty() && label.id == _frameEndId) {
                    _eobSampsLeft = label.index + label.width;
                    break;
                }
            }
        }
        if (_eobSampsLeft != 0) {
            if (_eobSampsLeft > avail) { inPort->setReserve(_eobSampsLeft); return; }  // wait for the whole frame
            avail = _eobSampsLeft;
        } else if (avail < _inputRequire) {
            inPort->setReserve(_inputRequire);
            return;
        }
        inPort->setReserve(0);

        // device call on [history | samples]; a burst tail shorter than M+K-1 is flushed with K-1 zeros
        const void *src = inPort->buffer().template as<const void *>();
        size_t srcElems = avail;
        this->pageLock(src, avail * _elemBytes);
        if (_eobSampsLeft != 0 && _eobSampsLeft < _inputRequire) {
            _flush.assign((_eobSampsLeft + K - 1) * _elemBytes, 0);
            // the port buffer may be a DEVICE slab (the upstream block is one of this module's, getInputBufferManager): the CPU must
            // not read it (this was a std::memcpy)
            check(pcx_memcpy_to_host(_flush.data(), src, _eobSampsLeft * _elemBytes), "FIRFilter::work()");
            src = _flush.data();
            srcElems = _eobSampsLeft + K - 1;
        }
        size_t consumed = 0, produced = 0;
        if (_sh && M == 1 && L == 1 && _eobSampsLeft == 0 && srcElems >= K) {
            // the sharded pass: G shards of C samples each out of what the port holds.  C follows what the calls bring: it is laid out
            // by the first call that brings a shard set worth the G launches and the exchange (kMinShard samples per device), laid
            // out again when a call brings twice as much (a small first call must not pin tiny shards for good) or when two calls in
            // a row bring less than one set (the port slabs shrank); a single shorter call -- the tail of a stream -- and anything
            // below kMinShard per device go to the single-device handle below
            const size_t G = _devices.size();
            const size_t N = std::min(srcElems - (K - 1), outPort->elements());
            const size_t want = N / G;
            const bool worth = want >= std::max(K, kMinShard);
            if (_shardC != 0 && N < G * _shardC) _shardShort++; else _shardShort = 0;
            if (worth && (_shardC == 0 || want >= 2 * _shardC || _shardShort >= 2)) {
                _shardC = want;
                _shardShort = 0;
                check(pcx_shard_configure(_sh, _shardC), "FIRFilter::work()");
            }
            if (_shardC != 0 && N >= G * _shardC) {
                check(pcx_shard_scatter(_sh, src, K - 1 + G * _shardC), "FIRFilter::work()");
                check(pcx_shard_step(_sh), "FIRFilter::work()");
                check(pcx_shard_gather(_sh, outPort->buffer().template as<void *>(), G * _shardC), "FIRFilter::work()");
                _shardPasses++;
                inPort->consume(G * _shardC);
                outPort->produce(G * _shardC);
                return;
            }
        }
        check(pcx_fir_process(_h, src, srcElems, outPort->buffer().template as<void *>(), outPort->elements(), &consumed, &produced),
              "FIRFilter::work()");      // (the handle carries its device: no switch needed around the call)

        // K-1 elements stay in the input buffer as filter history
        if (_eobSampsLeft != 0) _eobSampsLeft -= consumed;
        inPort->consume(consumed);
        outPort->produce(produced);
    }

    void propagateLabels(const pcxfw::InputPort *port)
    {
        auto outputPort = this->output(0);
        for (const auto &label : port->labels()) {
            auto newLabel = label.toAdjusted(L, M);
            if (label.id == "rxRate" && label.data.type() == typeid(double)) {
                newLabel.data = pcxfw::Object((double(label.data) * L) / M);
            }
            outputPort->postLabel(newLabel);
        }
    }

private:
    // The input buffer inside Pothos is the FRAMEWORK's circular buffer (getInputBufferManager below asks for it, as the reference
    // does, FIRFilter.cpp:196-199): pageable memory, mapped twice back to back.  Left as it is, every call would be staged through
    // the C ABI's bounce buffer by the CPU (0.9-1.2 Gsamples/s); page-locked where it lies, the kernel reads it in place over PCIe.
    // So: the first time a pageable buffer shows up -- and again whenever the port's address leaves what has been locked (the
    // framework re-allocated) -- the mapping that holds it is page-locked, both halves of the double mapping (pcx.h
    // pcx_host_register_mapping: shared file mappings only, never a heap arena; counted, so two blocks on one buffer share one
    // lock).  deactivate() and the destructor let go; activate() checks that what is still held is still mapped.  Memory that
    // cannot be locked (not a shared mapping; the runtime refuses) is remembered and staged as before.
    void pageLock(const void *p, size_t bytes)
    {
        if (bytes < kLockFrom) return;
        const uintptr_t lo = (uintptr_t)p, hi = lo + bytes;
        for (const auto &r : _locked) if ((uintptr_t)r.first <= lo && hi <= (uintptr_t)r.first + r.second) return;
        for (const auto &r : _unlockable) if (r.first <= lo && hi <= r.first + r.second) return;
        // (a range ANOTHER block of the module locked reads as page-locked here: pcx_host_register_mapping makes this block a holder
        // of it too, so that the other block's destructor cannot unlock it under this one; a pinned or device slab of somebody else's
        // comes back with *base NULL, nothing to hold)
        void *base = nullptr;
        size_t len = 0;
        if (pcx_host_register_mapping(p, bytes, 0, &base, &len) == PCX_OK && base) {
            _locked.emplace_back(base, len);
            if (_locked.size() > 4) {                   // the framework keeps re-allocating: let go of the oldest range
                // (the host-pointer entry points return with their result in place: no call of THIS block is reading the range, and
                // the count inside the library keeps it locked for any other block that holds it)
                (void)pcx_host_unregister(_locked.front().first);
                _locked.erase(_locked.begin());
            }
            return;
        }
        int kind = PCX_PTR_PAGEABLE;
        if (pcx_pointer_kind(p, &kind) != PCX_OK || kind != PCX_PTR_PAGEABLE) return;     // a pinned slab, a device slab: in place as it is
        {
            // a window that slides through one unlockable buffer must not cost a look at /proc/self/maps per call: a window that overlaps
            // or touches a range already known to be unlockable grows that range
            for (auto &r : _unlockable)
                if (lo <= r.first + r.second && r.first <= hi) {
                    const uintptr_t nlo = std::min(lo, r.first), nhi = std::max(hi, r.first + r.second);
                    r = std::make_pair(nlo, (size_t)(nhi - nlo));
                    return;
                }
            if (_unlockable.size() >= 16) _unlockable.erase(_unlockable.begin());
            _unlockable.emplace_back(lo, bytes);
        }
    }
    void unlockAll()
    {
        for (const auto &r : _locked) (void)pcx_host_unregister(r.first);
        _locked.clear();
        _unlockable.clear();
    }
    std::vector<std::pair<void *, size_t>> _locked;          // ranges this block holds page-locked (base, bytes)
    std::vector<std::pair<uintptr_t, size_t>> _unlockable;   // windows that could not be locked: not asked about again
    static constexpr size_t kLockFrom = 65536;               // bytes per call below which staging is as good

    static int kernelCode(const std::string &name)
    {
        const int algo = name == "AUTO" ? PCX_FIR_AUTO : name == "DIRECT" ? PCX_FIR_DIRECT : name == "OLS_FFT" ? PCX_FIR_OLS_FFT
                       : name == "EXACT" ? PCX_FIR_EXACT : -1;
        if (algo < 0) throw InvalidArgumentException("FIRFilter::setKernel(" + name + ")", "unknown kernel");
        return algo;
    }
    // setDevice: the single-device filter again on the device that is current now, with everything the setters have told it (the
    // shard set keeps its own devices)
    void rebind() override
    {
        pcx_fir *fresh = nullptr;
        check(pcx_fir_create(_scalar, _cplx ? 1 : 0, _complexTaps ? 1 : 0, &fresh), "FIRFilter::setDevice()");
        pcx_fir *old = _h;
        _h = fresh;
        try {
            check(pcx_fir_set_qformat(_h, _qformat.ptr()), "FIRFilter::setDevice()");
            check(pcx_fir_set_decimation(_h, M), "FIRFilter::setDevice()");
            check(pcx_fir_set_interpolation(_h, L), "FIRFilter::setDevice()");
            this->pushTaps();
            check(pcx_fir_set_algo(_h, kernelCode(_kernel)), "FIRFilter::setDevice()");
        } catch (...) {
            _h = old;
            pcx_fir_destroy(fresh);
            throw;
        }
        pcx_fir_destroy(old);
    }
    void pushTaps()
    {
        std::vector<double> flat;
        if (_complexTaps) {
            flat.resize(2 * _taps.size());
            for (size_t i = 0; i < _taps.size(); i++) { flat[2 * i] = _taps[i].real(); flat[2 * i + 1] = _taps[i].imag(); }
        } else {
            flat.resize(_taps.size());
            for (size_t i = 0; i < _taps.size(); i++) flat[i] = _taps[i].real();
        }
        check(pcx_fir_set_taps(_h, flat.data(), _taps.size()), "FIRFilter::setTaps()");
        if (_sh) { check(pcx_shard_set_taps(_sh, flat.data(), _taps.size(), _complexTaps ? 1 : 0), "FIRFilter::setTaps()"); _shardC = 0; }
        this->refreshGeometry();
    }
    void refreshGeometry() { check(pcx_fir_get_geometry(_h, &K, &_inputRequire), "FIRFilter::updateInternals()"); }

    std::vector<std::complex<double>> _taps;
    int _scalar;
    bool _cplx, _complexTaps;
    size_t _elemBytes;
    size_t M, L, K, _inputRequire;
    bool _waitTapsMode, _waitTapsArmed;
    std::string _frameStartId, _frameEndId;
    std::string _kernel = "AUTO";
    BlockQFormat _qformat;
    pcx_shard *_sh = nullptr;              // setDevices(): the stream over several devices
    std::vector<size_t> _devices;
    size_t _shardC = 0, _shardPasses = 0, _shardShort = 0;
    static constexpr size_t kMinShard = 32768;   // samples per device below which a sharded pass is not worth its G launches
    size_t _eobSampsLeft;
    DType _dtype;
    pcx_fir *_h;
    std::vector<char> _flush;
};

Block *FIRFilterFactory(const DType &dtype, const std::string &tapsType)
{
    int scalar;
    bool cplx;
    const bool known = parseElemType(dtype, scalar, cplx);
    const bool real = tapsType == "REAL", complexTaps = tapsType == "COMPLEX";
    if (known && dtype.dimension() == 1 && (real || (complexTaps && cplx))) return new FIRFilter(dtype, scalar, cplx, complexTaps);
    throw InvalidArgumentException("FIRFilterFactory(" + dtype.toString() + ")", "unsupported types");
}
pcxfw::BlockRegistry registerFIRFilter("/comms/fir_filter", &FIRFilterFactory);
pcxfw::BlockRegistry registerFIRFilterOldPath("/blocks/fir_filter", &FIRFilterFactory);

/***********************************************************************
 * |PothosDoc FFT
 *
 * Discrete Fourier transforms on the GPU: every numBins consecutive elements of input port 0 are one frame, its
 * transform leaves on output port 0.  All whole frames a call finds are transformed in one launch.  Unscaled in both
 * directions for the floating-point types; the fixed-point type divides by the radix at every stage, as the CPU block does.
 *
 * |category /FFT
 * |keywords dft fft fast fourier transform gpu hip
 *
 * |param dtype[Data Type] Element type of the input and of the output stream.
 * |widget DTypeChooser(cfloat=1, cint=1)
 * |default "complex_float32"
 * |preview disable
 *
 * |param numBins[Num FFT Bins] Frame length: the number of bins of one transform (any length, powers of two are fastest).
 * |default 1024
 * |option 512
 * |option 1024
 * |option 2048
 * |option 4096
 * |widget ComboBox(editable=true)
 *
 * |param inverse[Inverse FFT] Direction of the transform.
 * |option [Forward] false
 * |option [Inverse] true
 * |default false
 *
 * |param device[Device] Ordinal of the GPU that carries the block.
 * |default 0
 * |widget SpinBox(minimum=0)
 * |preview disable
 * |tab Device
 *
 * |param portSlabBytes[Port Slab Bytes] Size of the page-locked port buffers the block asks the framework for.
 * Larger slabs carry more samples per call (throughput), smaller ones return sooner (latency).
 * |default 67108864
 * |units bytes
 * |preview disable
 * |tab Device
 *
 * |factory /comms/fft(dtype, numBins, inverse)
 * |initializer setPortSlabBytes(portSlabBytes)
 * |initializer setDevice(device)
 **********************************************************************/
class FFT : public DeviceBlock {
public:
    FFT(const DType &dtype, int scalar, const size_t numBins, const bool inverse)
        : _scalar(scalar), _numBins(numBins), _inverse(inverse), _elemBytes(dtype.size()), _h(nullptr)
    {
        check(pcx_fft_create(scalar, numBins, inverse ? 1 : 0, &_h), "FFTFactory(" + dtype.toString() + ")");
        this->setupInput(0, dtype);
        this->setupOutput(0, dtype);
        this->input(0)->setReserve(_numBins);
    }
    ~FFT() { pcx_fft_destroy(_h); }

    pcxfw::BufferManager::Sptr getOutputBufferManager(const std::string &, const std::string &domain)
    {
        // a device launch per frame would be launch-bound: the block asks for output slabs of many frames and transforms every whole
        // frame present in one call -- as many as the slab setting holds, never less than one frame (the reference's own request,
        // FFT.cpp:54-59) and always a whole number of them.  Totals (consume == produce == frames * numBins) are those of the
        // reference's one-frame calls.
        const size_t frame = _numBins * _elemBytes;
        const size_t frames = std::max<size_t>(1, _slabBytes / frame);
#ifndef PCX_WITH_POTHOS
        if (domain == kDomain) { OnDevice on(_device); return deviceManager("generic", frame * frames); }
#else
        (void)domain;
#endif
        return pinnedManager("generic", frame * frames);
    }

    void work()
    {
        auto inPort = this->input(0);
        auto outPort = this->output(0);
        const size_t frames = std::min(inPort->elements(), outPort->elements()) / _numBins;
        if (frames == 0) return;
        check(pcx_fft_transform(_h, inPort->buffer().template as<const void *>(), outPort->buffer().template as<void *>(), frames),
              "FFT::work()");
        inPort->consume(frames * _numBins);
        outPort->produce(frames * _numBins);
    }

private:
    void rebind() override
    {
        pcx_fft *fresh = nullptr;
        check(pcx_fft_create(_scalar, _numBins, _inverse ? 1 : 0, &fresh), "FFT::setDevice()");
        pcx_fft_destroy(_h);
        _h = fresh;
    }
    const int _scalar;
    const size_t _numBins;
    const bool _inverse;
    const size_t _elemBytes;
    pcx_fft *_h;
};

Block *FFTFactory(const DType &dtype, const size_t numBins, const bool inverse)
{
    int scalar;
    bool cplx;
    if (parseElemType(dtype, scalar, cplx) && cplx && dtype.dimension() == 1 &&
        (scalar == PCX_F64 || scalar == PCX_F32 || scalar == PCX_I16))
        return new FFT(dtype, scalar, numBins, inverse);
    throw InvalidArgumentException("FFTFactory(" + dtype.toString() + ")", "unsupported type");
}
pcxfw::BlockRegistry registerFFT("/comms/fft", &FFTFactory);

/***********************************************************************
 * |PothosDoc Freq Demod
 *
 * FM demodulation on the GPU: output element n is the angle of in[n] times the conjugate of in[n-1] -- the phase
 * step from one complex sample to the next.  The last sample of a call is kept for the first of the following one;
 * activation starts from zero.
 *
 * |category /Demod
 * |keywords frequency modulation fm atan differential gpu hip
 *
 * |param dtype[Data Type] Element type of the complex input stream; the output stream is its real type.
 * Floating-point outputs are radians in [-pi, +pi]; fixed-point outputs map -pi ... +pi (exclusive) to the signed 16-bit range.
 * |widget DTypeChooser(cfloat=1,cint=1)
 * |default "complex_float32"
 * |preview disable
 *
 * |param device[Device] Ordinal of the GPU that carries the block.
 * |default 0
 * |widget SpinBox(minimum=0)
 * |preview disable
 * |tab Device
 *
 * |param portSlabBytes[Port Slab Bytes] Size of the page-locked port buffers the block asks the framework for.
 * Larger slabs carry more samples per call (throughput), smaller ones return sooner (latency).
 * |default 67108864
 * |units bytes
 * |preview disable
 * |tab Device
 *
 * |factory /comms/freq_demod(dtype)
 * |initializer setPortSlabBytes(portSlabBytes)
 * |initializer setDevice(device)
 **********************************************************************/
class FreqDemod : public DeviceBlock {
public:
    FreqDemod(const DType &dtype, int scalar) : _scalar(scalar), _h(nullptr)
    {
        check(pcx_freqdemod_create(scalar, &_h), "FreqDemodFactory(" + dtype.toString() + ")");
        this->setupInput(0, dtype);
        this->setupOutput(0, realOf(dtype));
    }
    ~FreqDemod() { pcx_freqdemod_destroy(_h); }
    void activate() { check(pcx_freqdemod_reset(_h), "FreqDemod::activate()"); }  // _prev = 0
    void work()
    {
        auto inPort = this->input(0);
        auto outPort = this->output(0);
        const size_t N = this->workInfo().minElements;
        if (N == 0) return;
        check(pcx_freqdemod_process(_h, inPort->buffer().template as<const void *>(), outPort->buffer().template as<void *>(), N),
              "FreqDemod::work()");
        inPort->consume(N);
        outPort->produce(N);
    }

private:
    void rebind() override      // (the carried sample starts over, as after activate())
    {
        pcx_freqdemod *fresh = nullptr;
        check(pcx_freqdemod_create(_scalar, &fresh), "FreqDemod::setDevice()");
        pcx_freqdemod_destroy(_h);
        _h = fresh;
    }
    const int _scalar;
    pcx_freqdemod *_h;
};
Block *FreqDemodFactory(const DType &dtype)
{
    int scalar;
    bool cplx;
    if (parseElemType(dtype, scalar, cplx) && cplx && dtype.dimension() == 1) return new FreqDemod(dtype, scalar);
    throw InvalidArgumentException("FreqDemodFactory(" + dtype.toString() + ")", "unsupported types");
}
pcxfw::BlockRegistry registerFreqDemod("/comms/freq_demod", &FreqDemodFactory);

/***********************************************************************
 * /comms/fm_demod_chain -- EXTENSION (no counterpart in the reference): /comms/rotate -> /comms/fir_filter -> /comms/freq_demod
 * as ONE block over the fused kernel (pcx_fmchain_*, BASELINE configs[4]).  A topology that wires those three blocks in a row
 * gets the same stream from this one with a single trip through the device: complex_float32 in, float32 out, the registered
 * calls of the three (setPhase / getPhase as Rotate.cpp:63-66; setTaps / getTaps as FIRFilter.cpp:113-124, REAL or COMPLEX by the
 * factory's tapsType); activate() resets the demodulator's carried sample as FreqDemod::activate does (FreqDemod.cpp:44-47).
 * Like the FIR it keeps K-1 samples of history at the front of its (circular) input buffer and produces one output per input.
 * Rotate's quirk is kept: until setPhase is called the phasor is zero and so is the output (Rotate.cpp:60-62).
 **********************************************************************/
/***********************************************************************
 * |PothosDoc FM Demod Chain
 *
 * Rotate, FIR filter and frequency demodulator as one GPU block: out[n] = angle(y[n] * conj(y[n-1])) with
 * y = taps convolved with in * exp(j*phase).  The same stream as the three separate blocks wired in a row, at one
 * trip through the device instead of three.
 *
 * |category /Demod
 * |category /Filter
 * |keywords fm demod rotate fir fused chain gpu hip
 *
 * |param dtype[Data Type] Element type of the input stream (the output is float32).
 * |widget DTypeChooser(cfloat=1)
 * |default "complex_float32"
 * |preview disable
 *
 * |param tapsType[Taps Type] Real or complex filter taps.
 * |option [Real] "REAL"
 * |option [Complex] "COMPLEX"
 *
 * |param phase[Phase] Rotation applied in front of the filter, in radians.
 * |units radians
 * |default 0.0
 *
 * |param taps The filter taps.
 * |default [1.0]
 *
 * |param device[Device] Ordinal of the GPU that carries the block.
 * |default 0
 * |widget SpinBox(minimum=0)
 * |preview disable
 * |tab Device
 *
 * |param portSlabBytes[Port Slab Bytes] Size of the page-locked port buffers the block asks the framework for.
 * Larger slabs carry more samples per call (throughput), smaller ones return sooner (latency).
 * |default 67108864
 * |units bytes
 * |preview disable
 * |tab Device
 *
 * |factory /comms/fm_demod_chain(dtype, tapsType)
 * |initializer setPortSlabBytes(portSlabBytes)
 * |initializer setDevice(device)
 * |setter setPhase(phase)
 * |setter setTaps(taps)
 **********************************************************************/
class FmDemodChain : public DeviceBlock {
public:
    FmDemodChain(const DType &dtype, bool complexTaps) : _complexTaps(complexTaps), _phase(0.0), _phaseSet(false), K(1), _h(nullptr)
    {
        check(pcx_fmchain_create(&_h), "fmDemodChainFactory(" + dtype.toString() + ")");
        this->setupInput(0, dtype);
        this->setupOutput(0, realOf(dtype));
        this->registerCall(this, PCX_FCN_TUPLE(FmDemodChain, setPhase));
        this->registerCall(this, PCX_FCN_TUPLE(FmDemodChain, getPhase));
        if (complexTaps) {
            this->registerCall(this, "setTaps", &FmDemodChain::setTapsComplex);
            this->registerCall(this, "getTaps", &FmDemodChain::getTapsComplex);
        } else {
            this->registerCall(this, "setTaps", &FmDemodChain::setTapsReal);
            this->registerCall(this, "getTaps", &FmDemodChain::getTapsReal);
        }
        _taps.assign(1, std::complex<double>(1.0, 0.0));
        this->pushTaps();
    }
    ~FmDemodChain() { pcx_fmchain_destroy(_h); }

    void setPhase(const double phase)
    {
        _phase = phase;
        _phaseSet = true;
        check(pcx_fmchain_set_phase(_h, phase), "FmDemodChain::setPhase()");
    }
    double getPhase() const { return _phase; }
    void setTapsReal(const std::vector<double> &taps)
    {
        if (taps.empty()) throw InvalidArgumentException("FmDemodChain::setTaps()", "taps cannot be empty");
        _taps.assign(taps.begin(), taps.end());
        this->pushTaps();
    }
    void setTapsComplex(const std::vector<std::complex<double>> &taps)
    {
        if (taps.empty()) throw InvalidArgumentException("FmDemodChain::setTaps()", "taps cannot be empty");
        _taps = taps;
        this->pushTaps();
    }
    std::vector<double> getTapsReal() const
    {
        std::vector<double> t(_taps.size());
        for (size_t i = 0; i < t.size(); i++) t[i] = _taps[i].real();
        return t;
    }
    std::vector<std::complex<double>> getTapsComplex() const { return _taps; }

    pcxfw::BufferManager::Sptr getInputBufferManager(const std::string &, const std::string &domain)
    {
#ifndef PCX_WITH_POTHOS
        if (domain == kDomain) { OnDevice on(_device); return deviceManager("circular", _slabBytes); }
#else
        (void)domain;
#endif
        return pinnedManager("circular", _slabBytes);
    }
    void activate() { check(pcx_fmchain_reset(_h), "FmDemodChain::activate()"); }

    void work()
    {
        auto inPort = this->input(0);
        auto outPort = this->output(0);
        const size_t avail = inPort->elements();
        if (avail == 0) return;
        if (avail < K) { inPort->setReserve(K); return; }          // one output needs K samples (K-1 of them history)
        inPort->setReserve(0);
        size_t consumed = 0, produced = 0;
        check(pcx_fmchain_process(_h, inPort->buffer().template as<const void *>(), avail, outPort->buffer().template as<void *>(), outPort->elements(),
                                  &consumed, &produced),
              "FmDemodChain::work()");
        inPort->consume(consumed);                                 // K-1 elements stay as history
        outPort->produce(produced);
    }

private:
    void rebind() override      // (the demodulator's carried sample starts over, as after activate(); the phasor stays unset if it was)
    {
        pcx_fmchain *fresh = nullptr;
        check(pcx_fmchain_create(&fresh), "FmDemodChain::setDevice()");
        pcx_fmchain *old = _h;
        _h = fresh;
        try {
            if (_phaseSet) check(pcx_fmchain_set_phase(_h, _phase), "FmDemodChain::setDevice()");
            this->pushTaps();
        } catch (...) {
            _h = old;
            pcx_fmchain_destroy(fresh);
            throw;
        }
        pcx_fmchain_destroy(old);
    }
    void pushTaps()
    {
        std::vector<double> flat;
        if (_complexTaps) {
            flat.resize(2 * _taps.size());
            for (size_t i = 0; i < _taps.size(); i++) { flat[2 * i] = _taps[i].real(); flat[2 * i + 1] = _taps[i].imag(); }
        } else {
            flat.resize(_taps.size());
            for (size_t i = 0; i < _taps.size(); i++) flat[i] = _taps[i].real();
        }
        check(pcx_fmchain_set_taps(_h, flat.data(), _taps.size(), _complexTaps ? 1 : 0), "FmDemodChain::setTaps()");
        K = _taps.size();
    }
    std::vector<std::complex<double>> _taps;
    bool _complexTaps;
    double _phase;
    bool _phaseSet;
    size_t K;
    pcx_fmchain *_h;
};
Block *fmDemodChainFactory(const DType &dtype, const std::string &tapsType)
{
    int scalar;
    bool cplx;
    if (!(parseElemType(dtype, scalar, cplx) && cplx && scalar == PCX_F32 && dtype.dimension() == 1))
        throw InvalidArgumentException("fmDemodChainFactory(" + dtype.toString() + ")", "unsupported types (complex_float32 only)");
    if (tapsType != "REAL" && tapsType != "COMPLEX")
        throw InvalidArgumentException("fmDemodChainFactory(" + dtype.toString() + ", " + tapsType + ")", "unsupported types");
    return new FmDemodChain(dtype, tapsType == "COMPLEX");
}
pcxfw::BlockRegistry registerFmDemodChain("/comms/fm_demod_chain", &fmDemodChainFactory);

/***********************************************************************
 * shared by Rotate and Scale: a coefficient that an upstream label may replace mid-stream
 **********************************************************************/
template <typename Derived>
class LabelDrivenMap : public DeviceBlock {
protected:
    // returns the number of elements to process this call, after applying a label that
    // sits at the front and cutting the call short before the next matching label
    size_t scanLabels(size_t elems)
    {
        if (_labelId.empty()) return elems;
        for (const auto &label : this->input(0)->labels()) {
            if (label.index >= elems) break;  // labels past the input bounds are not ours yet
            if (label.id != _labelId) continue;
            if (label.index == 0) static_cast<Derived *>(this)->applyLabel(label.data.template convert<double>());
            else { elems = label.index; break; }  // that label will be at index 0 next call
        }
        return elems;
    }
    std::string _labelId;
};

/***********************************************************************
 * |PothosDoc Rotate
 *
 * Turns every complex input element by a fixed phase on the GPU:
 *
 * out[n] = in[n] * exp(j*phase)
 *
 * |category /Math
 * |keywords math phase multiply rotate gpu hip
 *
 * |param dtype[Data Type] Element type of the stream (complex types; vectors allowed).
 * |widget DTypeChooser(cint=1, cfloat=1,dim=1)
 * |default "complex_float32"
 * |preview disable
 *
 * |param phase[Phase] The rotation, in radians.
 * |units radians
 * |default 0.0
 *
 * |param labelId[Label ID] ID of a label whose data replaces the phase from that element on.
 * An upstream block can steer the rotation along with the samples; empty: labels are ignored.
 * |preview valid
 * |default ""
 * |widget StringEntry()
 * |tab Labels
 *
 * |param qformat[Q Format] Fixed-point reading of the integer types: fractional bits, coefficient rounding, output rounding.
 * "DEFAULT", or three words such as "HALF_Q,TRUNCATE,FLOOR".
 * |default "DEFAULT"
 * |widget StringEntry()
 * |preview disable
 * |tab Device
 *
 * |param device[Device] Ordinal of the GPU that carries the block.
 * |default 0
 * |widget SpinBox(minimum=0)
 * |preview disable
 * |tab Device
 *
 * |param portSlabBytes[Port Slab Bytes] Size of the page-locked port buffers the block asks the framework for.
 * Larger slabs carry more samples per call (throughput), smaller ones return sooner (latency).
 * |default 67108864
 * |units bytes
 * |preview disable
 * |tab Device
 *
 * |factory /comms/rotate(dtype)
 * |initializer setPortSlabBytes(portSlabBytes)
 * |initializer setDevice(device)
 * |setter setPhase(phase)
 * |setter setLabelId(labelId)
 * |setter setQFormat(qformat)
 **********************************************************************/
class Rotate : public LabelDrivenMap<Rotate> {
public:
    Rotate(const DType &dtype, int scalar) : _scalar(scalar), _phase(0.0), _pr(0.0), _pi(0.0)
    {
        this->registerCall(this, PCX_FCN_TUPLE(Rotate, setPhase));
        this->registerCall(this, PCX_FCN_TUPLE(Rotate, getPhase));
        this->registerCall(this, PCX_FCN_TUPLE(Rotate, setLabelId));
        this->registerCall(this, PCX_FCN_TUPLE(Rotate, getLabelId));
        this->registerCall(this, PCX_FCN_TUPLE(Rotate, setQFormat));
        this->registerCall(this, PCX_FCN_TUPLE(Rotate, getQFormat));
        this->setupInput(0, dtype);
        this->setupOutput(0, dtype);
        // NB: like the reference, the phasor stays zero until setPhase() is called
    }
    void setPhase(const double phase)
    {
        _phase = phase;
        const std::complex<double> p = std::polar(1.0, phase);
        _pr = p.real();
        _pi = p.imag();
    }
    double getPhase() const { return _phase; }
    void setLabelId(const std::string &id) { _labelId = id; }
    std::string getLabelId() const { return _labelId; }
    void setQFormat(const std::string &spec) { _qformat.parse(spec, "Rotate::setQFormat"); }   // Rotate.cpp:21,74 (BlockQFormat above)
    std::string getQFormat() const { return _qformat.spec; }
    void applyLabel(double v) { this->setPhase(v); }
    void work()
    {
        auto elems = this->workInfo().minElements;
        if (elems == 0) return;
        auto inPort = this->input(0);
        auto outPort = this->output(0);
        elems = this->scanLabels(elems);
        const size_t N = elems * inPort->dtype().dimension();
        OnDevice on(_device);
        check(pcx_rotate_q(_scalar, _pr, _pi, _qformat.ptr(), inPort->buffer().template as<const void *>(),
                           outPort->buffer().template as<void *>(), N),
              "Rotate::work()");
        inPort->consume(elems);
        outPort->produce(elems);
    }

private:
    int _scalar;
    double _phase, _pr, _pi;
    BlockQFormat _qformat;
};
Block *rotateFactory(const DType &dtype)
{
    int scalar;
    bool cplx;
    if (parseElemType(dtype, scalar, cplx) && cplx) return new Rotate(dtype, scalar);
    throw InvalidArgumentException("rotateFactory(" + dtype.toString() + ")", "unsupported type");
}
pcxfw::BlockRegistry registerRotate("/comms/rotate", &rotateFactory);

/***********************************************************************
 * |PothosDoc Scale
 *
 * Multiplies every input element by a real factor on the GPU:
 *
 * out[n] = in[n] * factor
 *
 * |category /Math
 * |keywords math scale multiply factor gain gpu hip
 *
 * |param dtype[Data Type] Element type of the stream (real or complex; vectors allowed).
 * |widget DTypeChooser(float=1,cfloat=1,int=1,cint=1,dim=1)
 * |default "complex_float32"
 * |preview disable
 *
 * |param factor[Factor] The gain.
 * |default 0.0
 *
 * |param labelId[Label ID] ID of a label whose data replaces the factor from that element on.
 * An upstream block can steer the gain along with the samples; empty: labels are ignored.
 * |preview valid
 * |default ""
 * |widget StringEntry()
 * |tab Labels
 *
 * |param qformat[Q Format] Fixed-point reading of the integer types: fractional bits, coefficient rounding, output rounding.
 * "DEFAULT", or three words such as "HALF_Q,TRUNCATE,FLOOR".
 * |default "DEFAULT"
 * |widget StringEntry()
 * |preview disable
 * |tab Device
 *
 * |param device[Device] Ordinal of the GPU that carries the block.
 * |default 0
 * |widget SpinBox(minimum=0)
 * |preview disable
 * |tab Device
 *
 * |param portSlabBytes[Port Slab Bytes] Size of the page-locked port buffers the block asks the framework for.
 * Larger slabs carry more samples per call (throughput), smaller ones return sooner (latency).
 * |default 67108864
 * |units bytes
 * |preview disable
 * |tab Device
 *
 * |factory /comms/scale(dtype)
 * |initializer setPortSlabBytes(portSlabBytes)
 * |initializer setDevice(device)
 * |setter setFactor(factor)
 * |setter setLabelId(labelId)
 * |setter setQFormat(qformat)
 **********************************************************************/
class Scale : public LabelDrivenMap<Scale> {
public:
    Scale(const DType &dtype, int scalar, bool cplx) : _scalar(scalar), _cplx(cplx), _factor(0.0)
    {
        this->registerCall(this, PCX_FCN_TUPLE(Scale, setFactor));
        this->registerCall(this, PCX_FCN_TUPLE(Scale, getFactor));
        this->registerCall(this, PCX_FCN_TUPLE(Scale, setLabelId));
        this->registerCall(this, PCX_FCN_TUPLE(Scale, getLabelId));
        this->registerCall(this, PCX_FCN_TUPLE(Scale, setQFormat));
        this->registerCall(this, PCX_FCN_TUPLE(Scale, getQFormat));
        this->setupInput(0, dtype);
        this->setupOutput(0, dtype);
    }
    void setFactor(const double factor) { _factor = factor; }
    double getFactor() const { return _factor; }
    void setLabelId(const std::string &id) { _labelId = id; }
    std::string getLabelId() const { return _labelId; }
    void setQFormat(const std::string &spec) { _qformat.parse(spec, "Scale::setQFormat"); }   // Scale.cpp:21,73 (BlockQFormat above)
    std::string getQFormat() const { return _qformat.spec; }
    void applyLabel(double v) { this->setFactor(v); }
    void work()
    {
        auto elems = this->workInfo().minElements;
        if (elems == 0) return;
        auto inPort = this->input(0);
        auto outPort = this->output(0);
        elems = this->scanLabels(elems);
        const size_t N = elems * inPort->dtype().dimension();
        OnDevice on(_device);
        check(pcx_scale_q(_scalar, _cplx ? 1 : 0, _factor, _qformat.ptr(), inPort->buffer().template as<const void *>(),
                          outPort->buffer().template as<void *>(), N),
              "Scale::work()");
        inPort->consume(elems);
        outPort->produce(elems);
    }

private:
    int _scalar;
    bool _cplx;
    double _factor;
    BlockQFormat _qformat;
};
Block *scaleFactory(const DType &dtype)
{
    int scalar;
    bool cplx;
    if (parseElemType(dtype, scalar, cplx)) return new Scale(dtype, scalar, cplx);
    throw InvalidArgumentException("scaleFactory(" + dtype.toString() + ")", "unsupported type");
}
pcxfw::BlockRegistry registerScale("/comms/scale", &scaleFactory);

/***********************************************************************
 * |PothosDoc Abs
 *
 * Absolute value of every input element on the GPU: |x| for real streams, the magnitude for complex ones.
 *
 * out[n] = abs(in[n])
 *
 * |category /Math
 * |keywords math abs magnitude absolute gpu hip
 *
 * |param dtype[Data Type] Element type of the input stream; the output stream is its real type.
 * |widget DTypeChooser(float=1,cfloat=1,int=1,cint=1,dim=1)
 * |default "complex_float32"
 * |preview disable
 *
 * |param device[Device] Ordinal of the GPU that carries the block.
 * |default 0
 * |widget SpinBox(minimum=0)
 * |preview disable
 * |tab Device
 *
 * |param portSlabBytes[Port Slab Bytes] Size of the page-locked port buffers the block asks the framework for.
 * Larger slabs carry more samples per call (throughput), smaller ones return sooner (latency).
 * |default 67108864
 * |units bytes
 * |preview disable
 * |tab Device
 *
 * |factory /comms/abs(dtype)
 * |initializer setPortSlabBytes(portSlabBytes)
 * |initializer setDevice(device)
 **********************************************************************/
class Abs : public DeviceBlock {
public:
    Abs(const DType &dtype, int scalar, bool cplx) : _scalar(scalar), _cplx(cplx)
    {
        this->setupInput(0, dtype);
        this->setupOutput(0, realOf(dtype));
    }
    void work()
    {
        const auto elems = this->workInfo().minElements;
        if (elems == 0) return;
        auto inPort = this->input(0);
        auto outPort = this->output(0);
        const size_t N = elems * inPort->dtype().dimension();
        OnDevice on(_device);
        check(pcx_abs(_scalar, _cplx ? 1 : 0, inPort->buffer().template as<const void *>(), outPort->buffer().template as<void *>(), N),
              "Abs::work()");
        inPort->consume(elems);
        outPort->produce(elems);
    }

private:
    int _scalar;
    bool _cplx;
};
Block *absFactory(const DType &dtype)
{
    int scalar;
    bool cplx;
    if (parseElemType(dtype, scalar, cplx)) return new Abs(dtype, scalar, cplx);
    throw InvalidArgumentException("absFactory(" + dtype.toString() + ")", "unsupported type");
}
pcxfw::BlockRegistry registerAbs("/comms/abs", &absFactory);

/***********************************************************************
 * /comms/angle (math/Angle.cpp:50-110): the first "next" sibling, shares getAngle with FreqDemod
 **********************************************************************/
/***********************************************************************
 * |PothosDoc Angle
 *
 * The argument of every complex input element on the GPU:
 *
 * out[n] = atan2(Im{in[n]}, Re{in[n]})
 *
 * |category /Math
 * |keywords math angle complex arg atan gpu hip
 *
 * |param dtype[Data Type] Element type of the complex input stream; the output stream is its real type.
 * Floating-point outputs are radians in [-pi, +pi]; fixed-point outputs map -pi ... +pi (exclusive) to the signed 16-bit range.
 * |widget DTypeChooser(cfloat=1,cint=1,dim=1)
 * |default "complex_float32"
 * |preview disable
 *
 * |param device[Device] Ordinal of the GPU that carries the block.
 * |default 0
 * |widget SpinBox(minimum=0)
 * |preview disable
 * |tab Device
 *
 * |param portSlabBytes[Port Slab Bytes] Size of the page-locked port buffers the block asks the framework for.
 * Larger slabs carry more samples per call (throughput), smaller ones return sooner (latency).
 * |default 67108864
 * |units bytes
 * |preview disable
 * |tab Device
 *
 * |factory /comms/angle(dtype)
 * |initializer setPortSlabBytes(portSlabBytes)
 * |initializer setDevice(device)
 **********************************************************************/
class Angle : public DeviceBlock {
public:
    Angle(const DType &dtype, int scalar) : _scalar(scalar)
    {
        this->setupInput(0, dtype);
        this->setupOutput(0, realOf(dtype));
    }
    void work()
    {
        const auto elems = this->workInfo().minElements;
        if (elems == 0) return;
        auto inPort = this->input(0);
        auto outPort = this->output(0);
        const size_t N = elems * inPort->dtype().dimension();
        OnDevice on(_device);
        check(pcx_angle(_scalar, inPort->buffer().template as<const void *>(), outPort->buffer().template as<void *>(), N),
              "Angle::work()");
        inPort->consume(elems);
        outPort->produce(elems);
    }

private:
    int _scalar;
};
Block *angleFactory(const DType &dtype)
{
    int scalar;
    bool cplx;
    if (parseElemType(dtype, scalar, cplx) && cplx) return new Angle(dtype, scalar);
    throw InvalidArgumentException("angleFactory(" + dtype.toString() + ")", "unsupported type");
}
pcxfw::BlockRegistry registerAngle("/comms/angle", &angleFactory);

/***********************************************************************
 * |PothosDoc Conjugate
 *
 * The complex conjugate of every input element on the GPU:
 *
 * out[n] = conj(in[n])
 *
 * |category /Math
 * |keywords math conjugate complex conj gpu hip
 *
 * |param dtype[Data Type] Element type of the stream (complex types; vectors allowed).
 * |widget DTypeChooser(cfloat=1,cint=1,dim=1)
 * |default "complex_float32"
 * |preview disable
 *
 * |param device[Device] Ordinal of the GPU that carries the block.
 * |default 0
 * |widget SpinBox(minimum=0)
 * |preview disable
 * |tab Device
 *
 * |param portSlabBytes[Port Slab Bytes] Size of the page-locked port buffers the block asks the framework for.
 * Larger slabs carry more samples per call (throughput), smaller ones return sooner (latency).
 * |default 67108864
 * |units bytes
 * |preview disable
 * |tab Device
 *
 * |factory /comms/conjugate(dtype)
 * |initializer setPortSlabBytes(portSlabBytes)
 * |initializer setDevice(device)
 **********************************************************************/
class Conjugate : public DeviceBlock {
public:
    Conjugate(const DType &dtype, int scalar) : _scalar(scalar)
    {
        this->setupInput(0, dtype);
        this->setupOutput(0, dtype);
    }
    void work()
    {
        const auto elems = this->workInfo().minElements;
        if (elems == 0) return;
        auto inPort = this->input(0);
        auto outPort = this->output(0);
        const size_t N = elems * inPort->dtype().dimension();
        OnDevice on(_device);
        check(pcx_conj(_scalar, inPort->buffer().template as<const void *>(), outPort->buffer().template as<void *>(), N),
              "Conjugate::work()");
        inPort->consume(elems);
        outPort->produce(elems);
    }

private:
    int _scalar;
};
Block *conjugateFactory(const DType &dtype)
{
    int scalar;
    bool cplx;
    if (parseElemType(dtype, scalar, cplx) && cplx) return new Conjugate(dtype, scalar);
    throw InvalidArgumentException("conjugateFactory(" + dtype.toString() + ")", "unsupported type");
}
pcxfw::BlockRegistry registerConjugate("/comms/conjugate", &conjugateFactory);

/***********************************************************************
 * /comms/arithmetic (+ /blocks/arithmetic)   math/Arithmetic.cpp:150-305   
 **********************************************************************/
/***********************************************************************
 * |PothosDoc Arithmetic
 *
 * Element-wise arithmetic across the input ports on the GPU, folded from port 0 to the last:
 *
 * out[n] = in0[n] $op in1[n] $op ... $op in_last[n]
 *
 * |category /Math
 * |keywords math arithmetic add subtract multiply divide gpu hip
 * |alias /blocks/arithmetic
 *
 * |param dtype[Data Type] Element type of every port.
 * |widget DTypeChooser(float=1,cfloat=1,int=1,cint=1,uint=1,cuint=1,dim=1)
 * |default "complex_float32"
 * |preview disable
 *
 * |param operation The operation between the ports.
 * |default "ADD"
 * |option [Add] "ADD"
 * |option [Subtract] "SUB"
 * |option [Multiply] "MUL"
 * |option [Divide] "DIV"
 *
 * |param numInputs[Num Inputs] How many input ports the block has.
 * |default 2
 * |widget SpinBox(minimum=2)
 * |preview disable
 *
 * |param preload Elements of zeros queued on each input at activation (a list, one count per port):
 * what a feedback loop needs to start.
 * |default []
 * |widget ComboBox(editable=true)
 * |option [Ignored] \[\]
 * |preview disable
 *
 * |param device[Device] Ordinal of the GPU that carries the block.
 * |default 0
 * |widget SpinBox(minimum=0)
 * |preview disable
 * |tab Device
 *
 * |param portSlabBytes[Port Slab Bytes] Size of the page-locked port buffers the block asks the framework for.
 * Larger slabs carry more samples per call (throughput), smaller ones return sooner (latency).
 * |default 67108864
 * |units bytes
 * |preview disable
 * |tab Device
 *
 * |factory /comms/arithmetic(dtype, operation)
 * |initializer setNumInputs(numInputs)
 * |initializer setPreload(preload)
 * |initializer setPortSlabBytes(portSlabBytes)
 * |initializer setDevice(device)
 **********************************************************************/
class Arithmetic : public DeviceBlock {
public:
    Arithmetic(const DType &dtype, int scalar, bool cplx, int op) : _scalar(scalar), _cplx(cplx), _op(op), _numInlineBuffers(0)
    {
        this->registerCall(this, PCX_FCN_TUPLE(Arithmetic, setNumInputs));
        this->registerCall(this, PCX_FCN_TUPLE(Arithmetic, setPreload));
        this->registerCall(this, PCX_FCN_TUPLE(Arithmetic, preload));
        this->registerCall(this, PCX_FCN_TUPLE(Arithmetic, getNumInlineBuffers));
        this->setupInput(0, dtype);
        this->setupOutput(0, dtype, this->uid());   // unique domain because of inline buffer forwarding
        this->output(0)->setReadBeforeWrite(this->input(0));
    }
    void setNumInputs(const size_t numInputs)
    {
        if (numInputs < 2) throw pcxfw::RangeException("Arithmetic::setNumInputs(" + std::to_string(numInputs) + ")", "require inputs >= 2");
        for (size_t i = this->inputs().size(); i < numInputs; i++) this->setupInput(i, this->input(0)->dtype());
    }
    void setPreload(const std::vector<size_t> &preload)
    {
        this->setNumInputs(std::max<size_t>(2, preload.size()));
        _preload = preload;
    }
    std::vector<size_t> preload() const { return _preload; }
    void activate()
    {
        // feedback ports start with `preload` zero elements queued (Arithmetic.cpp:190-201)
        for (size_t i = 0; i < _preload.size(); i++) {
            const auto bytes = _preload[i] * this->input(i)->dtype().size();
            if (bytes == 0) continue;
            BufferChunk buffer(bytes);
            std::memset(buffer.as<void *>(), 0, buffer.length);
            this->input(i)->clear();
            this->input(i)->pushBuffer(buffer);
        }
    }
    void work()
    {
        const auto elems = this->workInfo().minElements;
        if (elems == 0) return;
        const std::vector<pcxfw::InputPort *> &inputs = this->inputs();
        auto output = this->output(0);
        void *out = output->buffer().template as<void *>();
        const void *in0 = inputs[0]->buffer().template as<const void *>();
        if (out == in0) _numInlineBuffers++;   // track buffer inlining
        const size_t N = elems * output->dtype().dimension();
        OnDevice on(_device);
        // left fold over the ports, the running result living in `out` (Arithmetic.cpp:217-224)
        for (size_t i = 1; i < inputs.size(); i++) {
            check(pcx_arith(_scalar, _cplx ? 1 : 0, _op, in0, inputs[i]->buffer().template as<const void *>(), out, N), "Arithmetic::work()");
            in0 = out;
            inputs[i]->consume(elems);
        }
        inputs[0]->consume(elems);
        output->produce(elems);
    }
    void propagateLabels(const pcxfw::InputPort *port)
    {
        // a feedback port: do not propagate labels from it
        if (_preload.size() > size_t(port->index()) && _preload[port->index()] > 0) return;
        Block::propagateLabels(port);
    }
    size_t getNumInlineBuffers() const { return _numInlineBuffers; }

private:
    int _scalar;
    bool _cplx;
    int _op;
    size_t _numInlineBuffers;
    std::vector<size_t> _preload;
};
Block *arithmeticFactory(const DType &dtype, const std::string &operation)
{
    int scalar;
    bool cplx;
    const int op = operation == "ADD" ? PCX_ARITH_ADD : operation == "SUB" ? PCX_ARITH_SUB : operation == "MUL" ? PCX_ARITH_MUL
                 : operation == "DIV" ? PCX_ARITH_DIV : -1;
    if (op >= 0 && parseArithType(dtype, scalar, cplx)) return new Arithmetic(dtype, scalar, cplx, op);
    throw InvalidArgumentException("arithmeticFactory(" + dtype.toString() + ", " + operation + ")", "unsupported args");
}
pcxfw::BlockRegistry registerArithmetic("/comms/arithmetic", &arithmeticFactory);
pcxfw::BlockRegistry registerArithmeticOldPath("/blocks/arithmetic", &arithmeticFactory);

/***********************************************************************
 * /comms/split_complex, /comms/combine_complex   utility/SplitComplex.cpp:39-77, utility/CombineComplex.cpp:38-76
 **********************************************************************/
/***********************************************************************
 * |PothosDoc Split Complex
 *
 * A complex stream taken apart on the GPU: real parts on output port "re", imaginary parts on "im".
 *
 * |category /Utility
 * |category /Convert
 *
 * |param dtype[Data Type] Element type of the two real output streams (the input is its complex type).
 * |widget DTypeChooser(float=1,int=1,dim=1)
 * |default "float32"
 * |preview disable
 *
 * |param device[Device] Ordinal of the GPU that carries the block.
 * |default 0
 * |widget SpinBox(minimum=0)
 * |preview disable
 * |tab Device
 *
 * |param portSlabBytes[Port Slab Bytes] Size of the page-locked port buffers the block asks the framework for.
 * Larger slabs carry more samples per call (throughput), smaller ones return sooner (latency).
 * |default 67108864
 * |units bytes
 * |preview disable
 * |tab Device
 *
 * |factory /comms/split_complex(dtype)
 * |initializer setPortSlabBytes(portSlabBytes)
 * |initializer setDevice(device)
 **********************************************************************/
class SplitComplex : public DeviceBlock {
public:
    SplitComplex(const DType &dtype, int scalar) : _scalar(scalar)
    {
        this->setupInput(0, complexOf(dtype));
        _rePort = this->setupOutput("re", dtype);
        _imPort = this->setupOutput("im", dtype);
    }
    void work()
    {
        const auto elems = this->workInfo().minAllElements;
        if (elems == 0) return;
        auto inPort = this->input(0);
        const size_t N = elems * inPort->dtype().dimension();
        OnDevice on(_device);
        check(pcx_split_complex(_scalar, inPort->buffer().template as<const void *>(), _rePort->buffer().template as<void *>(),
                                _imPort->buffer().template as<void *>(), N),
              "SplitComplex::work()");
        inPort->consume(elems);
        _rePort->produce(elems);
        _imPort->produce(elems);
    }

private:
    int _scalar;
    pcxfw::OutputPort *_rePort;
    pcxfw::OutputPort *_imPort;
};
/***********************************************************************
 * |PothosDoc Combine Complex
 *
 * A complex stream put together on the GPU from the real stream on input port "re" and the one on "im".
 *
 * |category /Utility
 * |category /Convert
 *
 * |param dtype[Data Type] Element type of the two real input streams (the output is its complex type).
 * |widget DTypeChooser(float=1,int=1,dim=1)
 * |default "float32"
 * |preview disable
 *
 * |param device[Device] Ordinal of the GPU that carries the block.
 * |default 0
 * |widget SpinBox(minimum=0)
 * |preview disable
 * |tab Device
 *
 * |param portSlabBytes[Port Slab Bytes] Size of the page-locked port buffers the block asks the framework for.
 * Larger slabs carry more samples per call (throughput), smaller ones return sooner (latency).
 * |default 67108864
 * |units bytes
 * |preview disable
 * |tab Device
 *
 * |factory /comms/combine_complex(dtype)
 * |initializer setPortSlabBytes(portSlabBytes)
 * |initializer setDevice(device)
 **********************************************************************/
class CombineComplex : public DeviceBlock {
public:
    CombineComplex(const DType &dtype, int scalar) : _scalar(scalar)
    {
        _rePort = this->setupInput("re", dtype);
        _imPort = this->setupInput("im", dtype);
        this->setupOutput(0, complexOf(dtype));
    }
    void work()
    {
        const auto elems = this->workInfo().minAllElements;
        if (elems == 0) return;
        auto outPort = this->output(0);
        const size_t N = elems * outPort->dtype().dimension();
        OnDevice on(_device);
        check(pcx_combine_complex(_scalar, _rePort->buffer().template as<const void *>(), _imPort->buffer().template as<const void *>(),
                                  outPort->buffer().template as<void *>(), N),
              "CombineComplex::work()");
        outPort->produce(elems);
        _rePort->consume(elems);
        _imPort->consume(elems);
    }

private:
    int _scalar;
    pcxfw::InputPort *_rePort;
    pcxfw::InputPort *_imPort;
};
// both factories take the REAL element type (splitComplexFactory / combineComplexFactory :60-70)
Block *splitComplexFactory(const DType &dtype)
{
    int scalar;
    bool cplx;
    if (parseElemType(dtype, scalar, cplx) && !cplx) return new SplitComplex(dtype, scalar);
    throw InvalidArgumentException("splitComplexFactory(" + dtype.toString() + ")", "unsupported type");
}
Block *combineComplexFactory(const DType &dtype)
{
    int scalar;
    bool cplx;
    if (parseElemType(dtype, scalar, cplx) && !cplx) return new CombineComplex(dtype, scalar);
    throw InvalidArgumentException("combineComplexFactory(" + dtype.toString() + ")", "unsupported type");
}
pcxfw::BlockRegistry registerSplitComplex("/comms/split_complex", &splitComplexFactory);
pcxfw::BlockRegistry registerCombineComplex("/comms/combine_complex", &combineComplexFactory);

}  // namespace
