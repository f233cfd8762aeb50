// DECLARATION-ONLY stand-in for <Pothos/Framework.hpp> -- test infrastructure, NOT PothosCore and not used by any build.
//
// Purpose: tests/test_pothos_syntax_cpu.py runs `g++ -fsyntax-only -DPCX_WITH_POTHOS -Itests/pothos_decl` over the block
// sources (csrc/blocks/comms_blocks.cpp, fir_designer.cpp), so that the branch which builds the blocks into a real Pothos
// plugin module is at least PARSED and TYPE-CHECKED on machines without PothosCore -- this container and the
// GPU box.  It proves nothing about PothosCore's behaviour: nothing here has a body worth the name, nothing links, and the
// signatures are the PothosCore 0.7 public API as the seven reference files use it (Block::{setupInput,
// setupOutput, registerCall, input, output, workInfo}, InputPort::{elements, buffer, labels, setReserve, consume, dtype},
// OutputPort::{elements, buffer, produce, postLabel}, BufferChunk, Label, Object, DType, BufferManager, BlockRegistry,
// InvalidArgumentException; e.g. /root/reference/filter/FIRFilter.cpp:113-124,196-199,385-389, fft/FFT.cpp:54-59) plus what the
// page-locked buffer manager needs (BufferManager's virtual interface, SharedBuffer, ManagedBuffer), written down from the
// public headers' documentation.  A real build uses the installed <Pothos/Framework.hpp>; if a signature here is wrong the
// real build says so and this file is what gets corrected.
//
// No reference source is compiled against this file (the reference's own tree is never built with stand-in headers).
#pragma once
#include <complex>
#include <cstddef>
#include <exception>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <typeinfo>
#include <vector>

namespace Pothos {

class Exception : public std::exception {
public:
    Exception(const std::string &msg, int code = 0);
    Exception(const std::string &msg, const std::string &arg, int code = 0);
    const char *what() const noexcept override;
    std::string message() const;
    std::string displayText() const;
};
#define PCX_DECL_EXCEPTION(CLS, BASE)                                  \
    class CLS : public BASE {                                          \
    public:                                                            \
        CLS(const std::string &msg, int code = 0);                     \
        CLS(const std::string &msg, const std::string &arg, int code = 0); \
    };
PCX_DECL_EXCEPTION(LogicException, Exception)
PCX_DECL_EXCEPTION(InvalidArgumentException, LogicException)
PCX_DECL_EXCEPTION(RangeException, LogicException)
PCX_DECL_EXCEPTION(RuntimeException, Exception)
PCX_DECL_EXCEPTION(NotFoundException, RuntimeException)
PCX_DECL_EXCEPTION(BlockCallNotFound, Exception)
#undef PCX_DECL_EXCEPTION

class DType {
public:
    DType();
    DType(const char *markup);
    DType(const std::string &markup);
    DType(const std::string &alias, const size_t dimension);
    DType(const std::type_info &type, const size_t dimension = 1);
    static DType fromDType(const DType &dtype, const size_t dimension);
    const std::string &name() const;
    size_t elemSize() const;
    size_t dimension() const;
    size_t size() const;
    std::string toString() const;
    explicit operator bool() const;
    bool isCustom() const;
    bool isFloat() const;
    bool isInteger() const;
    bool isSigned() const;
    bool isComplex() const;
};
bool operator==(const DType &lhs, const DType &rhs);
inline bool operator!=(const DType &lhs, const DType &rhs) { return !(lhs == rhs); }

class Object {
public:
    Object();
    template <typename ValueType> explicit Object(ValueType &&value);
    template <typename ValueType> static Object make(ValueType &&value);
    const std::type_info &type() const;
    bool canConvert(const std::type_info &type) const;
    template <typename ValueType> ValueType convert() const;
    template <typename ValueType> const ValueType &extract() const;
    template <typename ValueType> operator ValueType() const;
    explicit operator bool() const;
    std::string toString() const;
};

class Label {
public:
    Label();
    template <typename ValueType> Label(const std::string &id, ValueType &&data, const unsigned long long index, const size_t width = 1);
    Label toAdjusted(const size_t mult, const size_t div) const;
    Label &adjust(const size_t mult, const size_t div);
    std::string id;
    Object data;
    unsigned long long index;
    size_t width;
};
class LabelIteratorRange {
public:
    const Label *begin() const;
    const Label *end() const;
};

class BufferManager;
class SharedBuffer {
public:
    SharedBuffer();
    static SharedBuffer make(const size_t numBytes, const long nodeAffinity = -1);
    static SharedBuffer makeCirc(const size_t numBytes, const long nodeAffinity = -1);
    SharedBuffer(const size_t address, const size_t length, std::shared_ptr<void> container);
    SharedBuffer(const size_t address, const size_t length, const SharedBuffer &buffer);
    size_t getAddress() const;
    size_t getLength() const;
    explicit operator bool() const;
};
class ManagedBuffer {
public:
    ManagedBuffer();
    void reset(std::shared_ptr<BufferManager> manager, const SharedBuffer &buff, const size_t slabIndex = 0);
    const SharedBuffer &getBuffer() const;
    size_t getSlabIndex() const;
    explicit operator bool() const;
};
class BufferChunk {
public:
    static const BufferChunk &null();
    BufferChunk();
    BufferChunk(const size_t numBytes);
    BufferChunk(const DType &dtype, const size_t numElems);
    BufferChunk(const SharedBuffer &buffer);
    BufferChunk(const ManagedBuffer &buffer);
    size_t address;
    size_t length;
    DType dtype;
    size_t elements() const;
    void setElements(const size_t numElements);
    const SharedBuffer &getBuffer() const;
    const ManagedBuffer &getManagedBuffer() const;
    size_t getEnd() const;
    template <typename ElementType> ElementType as() const;
    template <typename ElementType> operator ElementType() const;
    explicit operator bool() const;
};

struct BufferManagerArgs {
    BufferManagerArgs();
    size_t numBuffers;
    size_t bufferSize;
    long nodeAffinity;
};
class BufferManager {
public:
    typedef std::shared_ptr<BufferManager> Sptr;
    virtual ~BufferManager();
    static Sptr make(const std::string &name);
    static Sptr make(const std::string &name, const BufferManagerArgs &args);
    virtual void init(const BufferManagerArgs &args);
    virtual bool empty() const = 0;
    const BufferChunk &front() const;
    virtual void pop(const size_t numBytes) = 0;
    virtual void push(const ManagedBuffer &buff) = 0;
    void setCallback(const std::function<void(const ManagedBuffer &)> &callback);
    bool isInitialized() const;

protected:
    BufferManager();
    void pushExternal(const ManagedBuffer &buff);
    void setFrontBuffer(const BufferChunk &buff);
};

class WorkInfo {
public:
    WorkInfo();
    std::vector<const void *> inputPointers;
    std::vector<void *> outputPointers;
    size_t minElements, minInElements, minOutElements, minAllElements, minAllInElements, minAllOutElements;
    long long maxTimeoutNs;
};

class InputPort {
public:
    int index() const;
    const std::string &name() const;
    const DType &dtype() const;
    const std::string &domain() const;
    const BufferChunk &buffer() const;
    size_t elements() const;
    unsigned long long totalElements() const;
    bool hasMessage();
    LabelIteratorRange labels() const;
    void removeLabel(const Label &label);
    void consume(const size_t numElements);
    Object popMessage();
    void setReserve(const size_t numElements);
    void pushBuffer(const BufferChunk &buffer);       // (math/Arithmetic.cpp:198-199 uses both)
    void clear();
};
class OutputPort {
public:
    int index() const;
    const std::string &name() const;
    const DType &dtype() const;
    const std::string &domain() const;
    const BufferChunk &buffer() const;
    size_t elements() const;
    unsigned long long totalElements() const;
    void produce(const size_t numElements);
    BufferChunk getBuffer(const size_t numElements);
    void popElements(const size_t numElements);
    template <typename... ArgsType> void postLabel(ArgsType &&... args);
    template <typename ValueType> void postMessage(ValueType &&message);
    void postBuffer(const BufferChunk &buffer);
    void setReadBeforeWrite(InputPort *port);
};

class Callable {
public:
    Callable();
    template <typename ReturnType, typename... ArgsType> Callable(ReturnType (*fcn)(ArgsType...));
    template <typename ReturnType, typename ClassType, typename... ArgsType> Callable(ReturnType (ClassType::*fcn)(ArgsType...));
};

class Block {
public:
    explicit Block();
    virtual ~Block();

protected:
    virtual void work();
    virtual void activate();
    virtual void deactivate();
    virtual void propagateLabels(const InputPort *input);
    virtual Object opaqueCallHandler(const std::string &name, const Object *inputArgs, const size_t numArgs);
    virtual std::shared_ptr<BufferManager> getInputBufferManager(const std::string &name, const std::string &domain);
    virtual std::shared_ptr<BufferManager> getOutputBufferManager(const std::string &name, const std::string &domain);

public:
    void setName(const std::string &name);
    const std::string &getName() const;
    std::string uid() const;
    const WorkInfo &workInfo() const;
    InputPort *input(const std::string &name) const;
    InputPort *input(const size_t index) const;
    OutputPort *output(const std::string &name) const;
    OutputPort *output(const size_t index) const;
    const std::vector<InputPort *> &inputs() const;
    const std::vector<OutputPort *> &outputs() const;
    const std::map<std::string, InputPort *> &allInputs() const;
    const std::map<std::string, OutputPort *> &allOutputs() const;
    InputPort *setupInput(const std::string &name, const DType &dtype = DType(), const std::string &domain = "");
    InputPort *setupInput(const size_t index, const DType &dtype = DType(), const std::string &domain = "");
    OutputPort *setupOutput(const std::string &name, const DType &dtype = DType(), const std::string &domain = "");
    OutputPort *setupOutput(const size_t index, const DType &dtype = DType(), const std::string &domain = "");
    template <typename ClassType, typename ReturnType, typename... ArgsType>
    void registerCall(ClassType *obj, const std::string &name, ReturnType (ClassType::*method)(ArgsType...));
    template <typename ClassType, typename ReturnType, typename... ArgsType>
    void registerCall(ClassType *obj, const std::string &name, ReturnType (ClassType::*method)(ArgsType...) const);
    void registerCallable(const std::string &name, const Callable &call);
    void registerSignal(const std::string &name);
    void registerSlot(const std::string &name);
    void registerProbe(const std::string &name, const std::string &signalName = "", const std::string &slotName = "");
    template <typename... ArgsType> void emitSignal(const std::string &name, ArgsType &&... args);
    template <typename ReturnType, typename... ArgsType> ReturnType call(const std::string &name, ArgsType &&... args) const;
    template <typename... ArgsType> Object call(const std::string &name, ArgsType &&... args) const;
    bool isActive() const;
    void yield();
};

class BlockRegistry {
public:
    BlockRegistry(const std::string &path, const Callable &factory);
    static bool doesBlockExist(const std::string &path);
};

}  // namespace Pothos

#define POTHOS_FCN_TUPLE(classPath, functionName) #functionName, &classPath::functionName
