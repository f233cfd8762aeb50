// pcx_framework.hpp -- the block-framework surface the MI355X comms blocks are written against.
//
// The blocks in comms_blocks.cpp use ONLY the part of the Pothos block API that the seven
// reference files use (Block::{setupInput,setupOutput,registerCall,input,output,
// workInfo}, InputPort::{elements,buffer,labels,setReserve,consume,dtype}, OutputPort::{elements,
// buffer,produce,postLabel}, BufferChunk, Label, Object, DType, BufferManager, BlockRegistry,
// InvalidArgumentException).
//
//   -DPCX_WITH_POTHOS : `namespace pcxfw` IS Pothos (<Pothos/Framework.hpp>): the same block
//                       sources build into a Pothos plugin module.
//   default           : a small self-contained runtime of that surface, so the blocks build
//                       and are exercised (tests/, the runner ABI in include/pcx_blocks.h) on
//                       machines without PothosCore -- this container and the GPU box.  It
//                       is our own host runtime, not a stand-in used to compile reference
//                       code: nothing under /root/reference is built against it.
#pragma once

#ifdef PCX_WITH_POTHOS
#include <Pothos/Framework.hpp>
namespace pcxfw = Pothos;
#define PCX_FCN_TUPLE(c, m) POTHOS_FCN_TUPLE(c, m)
#else

#include <algorithm>
#include <complex>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <typeindex>
#include <typeinfo>
#include <vector>

#include "pcx.h"   // pcx_host_alloc: page-locked slabs of the BufferManager stand-in

namespace pcxfw {

class Exception : public std::runtime_error {
public:
    Exception(const std::string &where, const std::string &what) : std::runtime_error(where + ": " + what) {}
    explicit Exception(const std::string &what) : std::runtime_error(what) {}
};
class InvalidArgumentException : public Exception {
public:
    using Exception::Exception;
};
class RangeException : public Exception {
public:
    RangeException(const std::string &where, const std::string &what) : Exception(where, what) {}
};
class BlockCallNotFound : public Exception {
public:
    using Exception::Exception;
};

// ---- DType: element type name, element size, dimension ----
class DType {
public:
    DType() : _name("unspecified"), _elemSize(1), _dim(1) {}
    DType(const std::string &name, size_t dimension = 1) : _dim(dimension) { init(name); }
    DType(const char *name, size_t dimension = 1) : _dim(dimension) { init(name); }
    DType(const std::type_info &t, size_t dimension = 1) : _dim(dimension) { init(nameOf(t)); }
    static DType fromDType(const DType &dt, size_t dimension) { DType d = dt; d._dim = dimension; return d; }
    const std::string &name() const { return _name; }
    std::string toString() const { return _dim == 1 ? _name : _name + "[" + std::to_string(_dim) + "]"; }
    size_t elemSize() const { return _elemSize; }
    size_t size() const { return _elemSize * _dim; }
    size_t dimension() const { return _dim; }
    bool isComplex() const { return _name.compare(0, 8, "complex_") == 0; }
    bool isFloat() const { return _name.find("float") != std::string::npos; }
    bool isInteger() const { return _name.find("int") != std::string::npos; }
    bool operator==(const DType &o) const { return _name == o._name && _dim == o._dim; }
    bool operator!=(const DType &o) const { return !(*this == o); }

private:
    static std::string nameOf(const std::type_info &t)
    {
#define PCXFW_T(T, n) if (t == typeid(T)) return n; if (t == typeid(std::complex<T>)) return std::string("complex_") + n;
        PCXFW_T(double, "float64") PCXFW_T(float, "float32") PCXFW_T(int64_t, "int64") PCXFW_T(int32_t, "int32")
        PCXFW_T(int16_t, "int16") PCXFW_T(int8_t, "int8")
#undef PCXFW_T
        throw InvalidArgumentException("DType(typeid)", "unknown type");
    }
    void init(std::string n)
    {
        // numpy-style aliases Pothos also accepts
        if (n == "complex64") n = "complex_float32";
        if (n == "complex128") n = "complex_float64";
        if (n == "float") n = "float32";
        if (n == "double") n = "float64";
        std::string base = n;
        size_t mult = 1;
        if (base.compare(0, 8, "complex_") == 0) { base = base.substr(8); mult = 2; }
        static const std::map<std::string, size_t> sz = {{"float64", 8}, {"float32", 4}, {"int64", 8}, {"int32", 4},
                                                         {"int16", 2}, {"int8", 1}, {"uint64", 8}, {"uint32", 4},
                                                         {"uint16", 2}, {"uint8", 1}};
        auto it = sz.find(base);
        if (it == sz.end()) throw InvalidArgumentException("DType(" + n + ")", "unknown name");
        _name = n;
        _elemSize = it->second * mult;
    }
    std::string _name;
    size_t _elemSize, _dim;
};

// ---- Object: a small type-erased value (numbers, strings, tap vectors, DType) ----
class Object {
public:
    Object() : _t(&typeid(void)) {}
    Object(bool v) : _t(&typeid(bool)), _d(v), _i(v) {}
    Object(int v) : _t(&typeid(int)), _d(v), _i(v) {}
    Object(long v) : _t(&typeid(long)), _d((double)v), _i(v) {}
    Object(long long v) : _t(&typeid(long long)), _d((double)v), _i(v) {}
    Object(unsigned v) : _t(&typeid(unsigned)), _d(v), _i(v) {}
    Object(unsigned long v) : _t(&typeid(unsigned long)), _d((double)v), _i((long long)v) {}
    Object(unsigned long long v) : _t(&typeid(unsigned long long)), _d((double)v), _i((long long)v) {}
    Object(float v) : _t(&typeid(float)), _d(v), _i((long long)v) {}
    Object(double v) : _t(&typeid(double)), _d(v), _i((long long)v) {}
    Object(const char *s) : _t(&typeid(std::string)), _s(s) {}
    Object(const std::string &s) : _t(&typeid(std::string)), _s(s) {}
    Object(const DType &d) : _t(&typeid(DType)), _i((long long)d.dimension()), _s(d.name()) {}
    Object(const std::vector<double> &v) : _t(&typeid(std::vector<double>)), _vd(v) {}
    Object(const std::vector<std::complex<double>> &v) : _t(&typeid(std::vector<std::complex<double>>)), _vc(v) {}
    Object(const std::vector<size_t> &v) : _t(&typeid(std::vector<size_t>)), _vs(v) {}

    const std::type_info &type() const { return *_t; }
    bool isNumber() const
    {
        return *_t == typeid(bool) || *_t == typeid(int) || *_t == typeid(long) || *_t == typeid(long long) ||
               *_t == typeid(unsigned) || *_t == typeid(unsigned long) || *_t == typeid(unsigned long long) ||
               *_t == typeid(float) || *_t == typeid(double);
    }
    bool canConvert(const std::type_info &to) const
    {
        if (to == *_t) return true;
        const Object probe(0.0);
        const bool toNumber = to == typeid(bool) || to == typeid(int) || to == typeid(long) || to == typeid(long long) ||
                              to == typeid(unsigned) || to == typeid(unsigned long) || to == typeid(unsigned long long) ||
                              to == typeid(float) || to == typeid(double);
        if (toNumber) return isNumber();
        if (to == typeid(DType)) return *_t == typeid(std::string);
        if (to == typeid(std::vector<std::complex<double>>)) return *_t == typeid(std::vector<double>);
        if (to == typeid(std::vector<size_t>)) return *_t == typeid(std::vector<double>);
        return false;
    }
    template <typename T>
    T convert() const
    {
        if (!canConvert(typeid(T))) throw Exception("Object::convert()", std::string("cannot convert ") + _t->name());
        return get(static_cast<T *>(nullptr));
    }
    template <typename T>
    explicit operator T() const { return convert<T>(); }

private:
    template <typename T>
    typename std::enable_if<std::is_arithmetic<T>::value, T>::type get(T *) const
    {
        if (*_t == typeid(float) || *_t == typeid(double)) return (T)_d;
        return (T)_i;
    }
    std::string get(std::string *) const { return _s; }
    DType get(DType *) const { return *_t == typeid(DType) ? DType(_s, (size_t)_i) : DType(_s); }
    std::vector<double> get(std::vector<double> *) const { return _vd; }
    std::vector<std::complex<double>> get(std::vector<std::complex<double>> *) const
    {
        if (*_t == typeid(std::vector<double>)) return std::vector<std::complex<double>>(_vd.begin(), _vd.end());
        return _vc;
    }
    std::vector<size_t> get(std::vector<size_t> *) const
    {
        if (*_t == typeid(std::vector<double>)) return std::vector<size_t>(_vd.begin(), _vd.end());
        return _vs;
    }
    const std::type_info *_t;
    double _d = 0;
    long long _i = 0;
    std::string _s;
    std::vector<double> _vd;
    std::vector<std::complex<double>> _vc;
    std::vector<size_t> _vs;
};

// ---- BufferChunk: a view (optionally owning) of typed memory ----
class BufferChunk {
public:
    size_t address = 0;
    size_t length = 0;
    DType dtype;
    BufferChunk() {}
    BufferChunk(const DType &dt, size_t numElems) : length(numElems * dt.size()), dtype(dt)
    {
        _own.reset(new char[length ? length : 1], std::default_delete<char[]>());
        address = reinterpret_cast<size_t>(_own.get());
    }
    BufferChunk(const std::type_info &t, size_t numElems) : BufferChunk(DType(t), numElems) {}
    explicit BufferChunk(size_t numBytes) : BufferChunk(DType("uint8"), numBytes) {}
    static BufferChunk view(void *p, size_t bytes, const DType &dt)
    {
        BufferChunk b;
        b.address = reinterpret_cast<size_t>(p);
        b.length = bytes;
        b.dtype = dt;
        return b;
    }
    size_t elements() const { return length / dtype.size(); }
    template <typename T>
    T as() const { return reinterpret_cast<T>(address); }
    template <typename T>
    operator T *() const { return reinterpret_cast<T *>(address); }

private:
    std::shared_ptr<char> _own;
};

// ---- Label ----
struct Label {
    std::string id;
    Object data;
    size_t index = 0;
    size_t width = 1;
    Label() {}
    Label(const std::string &id_, const Object &data_, size_t index_, size_t width_ = 1) : id(id_), data(data_), index(index_), width(width_) {}
    // Pothos::Label::toAdjusted [ext]: index and width scaled by mult/div (integer arithmetic)
    Label toAdjusted(size_t mult, size_t div) const
    {
        Label l = *this;
        l.index = l.index * mult / div;
        l.width = l.width * mult / div;
        return l;
    }
};

struct WorkInfo {
    // minElements: over the indexed ports; minAllElements: indexed and named ports alike
    size_t minElements = 0, minInElements = 0, minOutElements = 0, minAllElements = 0;
};

struct BufferManagerArgs {
    size_t numBuffers = 4;
    size_t bufferSize = 8 * 1024;
    long nodeAffinity = -1;
    // extension: slabs of page-locked host memory (pcx_host_alloc).  The device path runs its kernels directly on such
    // buffers (include/pcx.h, host-pointer entry points); pageable slabs are staged through a device workspace.
    bool pinned = false;
    // extension: slabs in DEVICE memory (pcx_dev_alloc) for an edge between two blocks of this module: the downstream block's
    // kernels read what the upstream block's kernels wrote, and the samples never cross PCIe.  Host code must not touch them.
    bool device = false;
};
// What a block hands the scheduler from getInputBufferManager / getOutputBufferManager.  The stand-in keeps the two
// manager names the reference uses ("generic", "circular": FIRFilter.cpp:196-199, FFT.cpp:54-59) and really owns its
// slabs: the runner (pcxb_acquire_buffer) draws port buffers from it round-robin, as the Pothos scheduler draws them
// from the manager a block returns.  In a real Pothos build the same role is played by a Pothos::BufferManager
// subclass whose SharedBuffers wrap pcx_host_alloc slabs.
class BufferManager {
public:
    typedef std::shared_ptr<BufferManager> Sptr;
    static Sptr make(const std::string &name, const BufferManagerArgs &args = BufferManagerArgs())
    {
        Sptr p(new BufferManager());
        p->name = name;
        p->args = args;
        return p;
    }
    ~BufferManager()
    {
        for (void *s : _slabs) release(s);
    }
    // next slab, at least `minBytes` long (allocated on first use; slabs grow to the largest request seen)
    void *acquire(size_t minBytes, size_t *bytes)
    {
        const size_t want = std::max(minBytes, args.bufferSize);
        if (_slabs.empty()) {
            _slabs.assign(std::max<size_t>(args.numBuffers, 1), nullptr);
            _sizes.assign(_slabs.size(), 0);
            _kind = args.device ? 2 : args.pinned ? 1 : 0;
        }
        const size_t i = _next++ % _slabs.size();
        if (_sizes[i] < want) {
            release(_slabs[i]);
            _slabs[i] = nullptr; _sizes[i] = 0;
            void *p = nullptr;
            if (_kind == 2) { if (pcx_dev_alloc(&p, want) != PCX_OK) throw std::runtime_error(std::string("BufferManager: pcx_dev_alloc: ") + pcx_last_error()); }
            else if (_kind == 1) { if (pcx_host_alloc(&p, want) != PCX_OK) throw std::runtime_error(std::string("BufferManager: pcx_host_alloc: ") + pcx_last_error()); }
            else if (!(p = std::malloc(want))) throw std::bad_alloc();
            _slabs[i] = p; _sizes[i] = want;
        }
        if (bytes) *bytes = _sizes[i];
        return _slabs[i];
    }
    std::string name;
    BufferManagerArgs args;

private:
    BufferManager() = default;
    void release(void *s)
    {
        if (!s) return;
        if (_kind == 2) (void)pcx_dev_free(s);
        else if (_kind == 1) (void)pcx_host_free(s);
        else std::free(s);
    }
    std::vector<void *> _slabs;
    std::vector<size_t> _sizes;
    size_t _next = 0;
    int _kind = 0;     // 0 pageable, 1 page-locked, 2 device
};

class Block;
class InputPort {
public:
    const BufferChunk &buffer() const { return _buffer; }
    size_t elements() const { return _buffer.elements(); }
    const std::vector<Label> &labels() const { return _labels; }
    void consume(size_t n) { _consumed += n; }
    void setReserve(size_t n) { _reserve = n; _reserveSet = true; }
    const DType &dtype() const { return _dtype; }
    const std::string &domain() const { return _domain; }   // Pothos port domain [ext]: which memory the port's block can address
    std::string _domain;
    int index() const { return _index; }            // -1 for a named (non-indexed) port
    const std::string &name() const { return _name; }
    // feedback preload (Arithmetic::activate): drop what is queued, queue a caller-made buffer
    void clear() { _pushed.clear(); }
    void pushBuffer(const BufferChunk &b) { _pushed.push_back(b); }
    // runtime side
    BufferChunk _buffer;
    std::vector<Label> _labels;
    size_t _consumed = 0, _reserve = 0;
    bool _reserveSet = false;
    DType _dtype;
    int _index = -1;
    std::string _name;
    std::vector<BufferChunk> _pushed;
};
class OutputPort {
public:
    const BufferChunk &buffer() const { return _buffer; }
    size_t elements() const { return _buffer.elements(); }
    void produce(size_t n) { _produced += n; }
    void postLabel(const Label &l) { _posted.push_back(l); }
    const DType &dtype() const { return _dtype; }
    const std::string &domain() const { return _domain; }
    std::string _domain;
    int index() const { return _index; }
    const std::string &name() const { return _name; }
    // buffer-inlining hint (Arithmetic ctor): the runner decides whether out aliases that input
    void setReadBeforeWrite(InputPort *p) { _readBeforeWrite = p; }
    BufferChunk _buffer;
    size_t _produced = 0;
    std::vector<Label> _posted;
    DType _dtype;
    int _index = -1;
    std::string _name;
    InputPort *_readBeforeWrite = nullptr;
};

// ---- registered calls: name -> type-erased member function ----
namespace detail {
template <typename T>
struct decay_arg { typedef typename std::remove_cv<typename std::remove_reference<T>::type>::type type; };
template <typename C, typename R, typename... A, size_t... I>
Object invoke(C *self, R (C::*f)(A...), const std::vector<Object> &args, std::index_sequence<I...>, std::false_type)
{
    return Object((self->*f)(args[I].template convert<typename decay_arg<A>::type>()...));
}
template <typename C, typename R, typename... A, size_t... I>
Object invoke(C *self, R (C::*f)(A...), const std::vector<Object> &args, std::index_sequence<I...>, std::true_type)
{
    (self->*f)(args[I].template convert<typename decay_arg<A>::type>()...);
    return Object();
}
}  // namespace detail

class Block {
public:
    virtual ~Block() {}
    virtual void work() {}
    virtual void activate() {}
    virtual void deactivate() {}
    virtual void propagateLabels(const InputPort *port)
    {
        for (const auto &l : port->labels()) {
            for (auto &o : _outputs) o->postLabel(l);
            for (auto &o : _namedOutputs) o->postLabel(l);
        }
    }
    virtual BufferManager::Sptr getInputBufferManager(const std::string &, const std::string &) { return BufferManager::Sptr(); }
    virtual BufferManager::Sptr getOutputBufferManager(const std::string &, const std::string &) { return BufferManager::Sptr(); }

    InputPort *setupInput(size_t i, const DType &dt = DType(), const std::string &domain = "")
    {
        if (_inputs.size() <= i) _inputs.resize(i + 1);
        _inputs[i].reset(new InputPort());
        _inputs[i]->_domain = domain;
        _inputs[i]->_dtype = dt;
        _inputs[i]->_index = (int)i;
        _inputs[i]->_name = std::to_string(i);
        return _inputs[i].get();
    }
    OutputPort *setupOutput(size_t i, const DType &dt = DType(), const std::string &domain = "")
    {
        if (_outputs.size() <= i) _outputs.resize(i + 1);
        _outputs[i].reset(new OutputPort());
        _outputs[i]->_domain = domain;
        _outputs[i]->_dtype = dt;
        _outputs[i]->_index = (int)i;
        _outputs[i]->_name = std::to_string(i);
        return _outputs[i].get();
    }
    // named ports (SplitComplex "re"/"im", CombineComplex): kept behind the indexed ones
    InputPort *setupInput(const std::string &name, const DType &dt = DType(), const std::string &domain = "")
    {
        _namedInputs.emplace_back(new InputPort());
        _namedInputs.back()->_domain = domain;
        _namedInputs.back()->_dtype = dt;
        _namedInputs.back()->_name = name;
        return _namedInputs.back().get();
    }
    OutputPort *setupOutput(const std::string &name, const DType &dt = DType(), const std::string &domain = "")
    {
        _namedOutputs.emplace_back(new OutputPort());
        _namedOutputs.back()->_domain = domain;
        _namedOutputs.back()->_dtype = dt;
        _namedOutputs.back()->_name = name;
        return _namedOutputs.back().get();
    }
    InputPort *input(size_t i) const { return _inputs.at(i).get(); }
    OutputPort *output(size_t i) const { return _outputs.at(i).get(); }
    InputPort *input(const std::string &name) const
    {
        for (const auto &p : _namedInputs) if (p->name() == name) return p.get();
        for (const auto &p : _inputs) if (p && p->name() == name) return p.get();
        throw Exception("Block::input(" + name + ")", "no such port");
    }
    OutputPort *output(const std::string &name) const
    {
        for (const auto &p : _namedOutputs) if (p->name() == name) return p.get();
        for (const auto &p : _outputs) if (p && p->name() == name) return p.get();
        throw Exception("Block::output(" + name + ")", "no such port");
    }
    // the indexed ports, in index order (Pothos::Block::inputs())
    const std::vector<InputPort *> &inputs()
    {
        _inputPtrs.clear();
        for (const auto &p : _inputs) _inputPtrs.push_back(p.get());
        return _inputPtrs;
    }
    const std::vector<OutputPort *> &outputs()
    {
        _outputPtrs.clear();
        for (const auto &p : _outputs) _outputPtrs.push_back(p.get());
        return _outputPtrs;
    }
    // every port, indexed first (runtime side)
    std::vector<InputPort *> allInputs() const
    {
        std::vector<InputPort *> v;
        for (const auto &p : _inputs) v.push_back(p.get());
        for (const auto &p : _namedInputs) v.push_back(p.get());
        return v;
    }
    std::vector<OutputPort *> allOutputs() const
    {
        std::vector<OutputPort *> v;
        for (const auto &p : _outputs) v.push_back(p.get());
        for (const auto &p : _namedOutputs) v.push_back(p.get());
        return v;
    }
    std::string uid() const { return "pcx-" + std::to_string(reinterpret_cast<size_t>(this)); }
    const WorkInfo &workInfo() const { return _workInfo; }
    WorkInfo &workInfoMutable() { return _workInfo; }

    template <typename C, typename R, typename... A>
    void registerCall(C *self, const std::string &name, R (C::*f)(A...))
    {
        _calls[name] = [self, f, name](const std::vector<Object> &args) {
            if (args.size() != sizeof...(A)) throw Exception("Block::call(" + name + ")", "wrong number of arguments");
            return detail::invoke(self, f, args, std::index_sequence_for<A...>(), std::is_void<R>());
        };
        _callArity[name] = sizeof...(A);
    }
    template <typename C, typename R, typename... A>
    void registerCall(C *self, const std::string &name, R (C::*f)(A...) const)
    {
        registerCall(self, name, reinterpret_cast<R (C::*)(A...)>(f));
    }
    Object call(const std::string &name, const std::vector<Object> &args = std::vector<Object>())
    {
        auto it = _calls.find(name);
        if (it == _calls.end()) throw BlockCallNotFound("Block::call(" + name + ")", "method does not exist in registry");
        return it->second(args);
    }
    bool hasCall(const std::string &name) const { return _calls.count(name) != 0; }
    // the registered calls and how many arguments each takes (what a block description's |setter / |initializer lines are checked against)
    const std::map<std::string, size_t> &callArities() const { return _callArity; }

    // ---- signals (Pothos::Block::registerSignal / emitSignal; Topology::connect(src, "sig", dst, "slot")) ----
    void registerSignal(const std::string &name) { _signals[name]; }
    bool hasSignal(const std::string &name) const { return _signals.count(name) != 0; }
    void connectSignal(const std::string &name, Block *dst, const std::string &slot)
    {
        auto it = _signals.find(name);
        if (it == _signals.end()) throw Exception("Block::connectSignal(" + name + ")", "no such signal");
        if (!dst->hasCall(slot)) throw BlockCallNotFound("Block::connectSignal(" + slot + ")", "method does not exist in registry");
        it->second.push_back(std::make_pair(dst, slot));
    }
    template <typename... A>
    void emitSignal(const std::string &name, const A &... a)
    {
        auto it = _signals.find(name);
        if (it == _signals.end()) throw Exception("Block::emitSignal(" + name + ")", "no such signal");
        const std::vector<Object> args{Object(a)...};
        for (const auto &d : it->second) d.first->call(d.second, args);
    }
    // true from just before activate() until deactivate() returns, as the Pothos actor keeps it
    bool isActive() const { return _active; }
    void setActiveState(bool on) { _active = on; }

private:
    std::vector<std::unique_ptr<InputPort>> _inputs, _namedInputs;
    std::vector<std::unique_ptr<OutputPort>> _outputs, _namedOutputs;
    std::vector<InputPort *> _inputPtrs;
    std::vector<OutputPort *> _outputPtrs;
    WorkInfo _workInfo;
    std::map<std::string, std::function<Object(const std::vector<Object> &)>> _calls;
    std::map<std::string, size_t> _callArity;
    std::map<std::string, std::vector<std::pair<Block *, std::string>>> _signals;
    bool _active = false;
};

#define PCX_FCN_TUPLE(c, m) #m, &c::m

// ---- BlockRegistry: path -> factory ----
class BlockRegistry {
public:
    typedef std::function<Block *(const std::vector<Object> &)> Factory;
    template <typename... A>
    BlockRegistry(const std::string &path, Block *(*f)(A...))
    {
        table()[path] = [f, path](const std::vector<Object> &args) -> Block * {
            if (args.size() != sizeof...(A)) throw InvalidArgumentException("BlockRegistry::make(" + path + ")", "wrong number of arguments");
            return call(f, args, std::index_sequence_for<A...>());
        };
        arities()[path] = sizeof...(A);
    }
    // how many arguments the factory behind `path` takes (-1: no such path)
    static long arity(const std::string &path)
    {
        auto it = arities().find(path);
        return it == arities().end() ? -1L : (long)it->second;
    }
    static Block *make(const std::string &path, const std::vector<Object> &args)
    {
        auto it = table().find(path);
        if (it == table().end()) throw InvalidArgumentException("BlockRegistry::make(" + path + ")", "no such registry path");
        return it->second(args);
    }
    static bool doesBlockExist(const std::string &path) { return table().count(path) != 0; }
    static std::vector<std::string> paths()
    {
        std::vector<std::string> p;
        for (const auto &kv : table()) p.push_back(kv.first);
        return p;
    }

private:
    template <typename... A, size_t... I>
    static Block *call(Block *(*f)(A...), const std::vector<Object> &args, std::index_sequence<I...>)
    {
        return f(args[I].template convert<typename detail::decay_arg<A>::type>()...);
    }
    static std::map<std::string, Factory> &table()
    {
        static std::map<std::string, Factory> t;
        return t;
    }
    static std::map<std::string, size_t> &arities()
    {
        static std::map<std::string, size_t> t;
        return t;
    }
};

}  // namespace pcxfw
#endif  // PCX_WITH_POTHOS
