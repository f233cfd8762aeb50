"""Thin Python handles over the C ABI (include/pcx.h) -- plumbing for tests, bench.py
and the multi-GPU stream driver.  Host arrays are numpy, device arrays are torch CUDA
(ROCm) tensors used purely as device memory; every call goes through libpcx_hip.so.

Stream layout: a complex stream of scalar type T is an array of shape (n, 2) of T
(interleaved re, im = std::complex<T>); numpy complex64/complex128 are accepted and
viewed that way.  Real streams are 1-D arrays of T.
"""
import ctypes as C
import math

import numpy as np

from . import _lib
from ._lib import F64, F32, I64, I32, I16, I8, U64, U32, U16, U8  # noqa: F401  (re-exported)

NP_SCALAR = {F64: np.float64, F32: np.float32, I64: np.int64, I32: np.int32, I16: np.int16, I8: np.int8,
             U64: np.uint64, U32: np.uint32, U16: np.uint16, U8: np.uint8}   # unsigned: arithmetic only
ARITH_OPS = {"ADD": _lib.ARITH_ADD, "SUB": _lib.ARITH_SUB, "MUL": _lib.ARITH_MUL, "DIV": _lib.ARITH_DIV}
SCALAR_OF_NP = {np.dtype(v): k for k, v in NP_SCALAR.items()}

# Pothos DType names (DType::toString) -> (scalar code, is_complex)
DTYPE_NAMES = {}
for _code, _nm in ((F64, "float64"), (F32, "float32"), (I64, "int64"), (I32, "int32"), (I16, "int16"), (I8, "int8"),
                   (U64, "uint64"), (U32, "uint32"), (U16, "uint16"), (U8, "uint8")):
    DTYPE_NAMES[_nm] = (_code, False)
    DTYPE_NAMES["complex_" + _nm] = (_code, True)
DTYPE_NAMES["complex64"] = (F32, True)    # Pothos accepts numpy-style aliases
DTYPE_NAMES["complex128"] = (F64, True)
DTYPE_NAMES["float"] = (F32, False)
DTYPE_NAMES["double"] = (F64, False)


def parse_dtype(dtype):
    """'complex_float32' / np.dtype / (scalar, is_complex) -> (scalar code, is_complex)."""
    if isinstance(dtype, tuple):
        return dtype
    if isinstance(dtype, str):
        if dtype not in DTYPE_NAMES:
            raise _lib.InvalidArgument(_lib.ERR_ARG, "unknown dtype %r" % dtype)
        return DTYPE_NAMES[dtype]
    dt = np.dtype(dtype)
    if dt == np.complex64:
        return (F32, True)
    if dt == np.complex128:
        return (F64, True)
    return (SCALAR_OF_NP[dt], False)


def as_pairs(a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.complex64:
        return a.view(np.float32).reshape(-1, 2)
    if a.dtype == np.complex128:
        return a.view(np.float64).reshape(-1, 2)
    return a


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _dev_ptr(t):
    if not t.is_cuda:
        raise ValueError("device variant needs a CUDA/ROCm tensor")
    if not t.is_contiguous():
        raise ValueError("device buffers must be contiguous")
    return C.c_void_p(t.data_ptr())


def _gate_ptr(t):
    """a gate word: device memory, or page-locked host memory (the device reads it in place; the host opens it with a plain store)"""
    if not t.is_cuda and t.is_pinned():
        return C.c_void_p(t.data_ptr())
    return _dev_ptr(t)


def _stream_ptr(stream=None):
    import torch
    s = torch.cuda.current_stream() if stream is None else stream
    return C.c_void_p(s if isinstance(s, int) else s.cuda_stream)      # a torch stream, or a raw hipStream_t


class _Handle:
    _destroy = None

    def __init__(self):
        self._h = C.c_void_p()

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            getattr(_lib.load(), self._destroy)(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FirFilter(_Handle):
    """pcx_fir_*: the loop of filter/FIRFilter.cpp:278-308 behind FIRFilterFactory's type matrix."""
    _destroy = "pcx_fir_destroy"

    def __init__(self, dtype="complex_float32", taps_type="COMPLEX"):
        super().__init__()
        self.scalar, self.is_complex = parse_dtype(dtype)
        if taps_type not in ("REAL", "COMPLEX"):
            raise _lib.InvalidArgument(_lib.ERR_ARG, "FIRFilterFactory: unsupported types")
        self.complex_taps = taps_type == "COMPLEX"
        _lib.check(_lib.load().pcx_fir_create(self.scalar, int(self.is_complex), int(self.complex_taps), C.byref(self._h)))

    def set_taps(self, taps):
        t = np.asarray(taps)
        if self.complex_taps:
            t = np.ascontiguousarray(t.astype(np.complex128)).view(np.float64)
            n = t.size // 2
        else:
            t = np.ascontiguousarray(np.real(t).astype(np.float64))
            n = t.size
        _lib.check(_lib.load().pcx_fir_set_taps(self._h, _np_ptr(t), n))

    def set_decimation(self, m):
        _lib.check(_lib.load().pcx_fir_set_decimation(self._h, m))

    def set_interpolation(self, l):
        _lib.check(_lib.load().pcx_fir_set_interpolation(self._h, l))

    def set_algo(self, algo):
        _lib.check(_lib.load().pcx_fir_set_algo(self._h, algo))

    def set_qformat(self, q):
        """pcx_fir_set_qformat: the floatToQ / fromQ reading of an integer filter -- a (frac, float_to_q, from_q) triple of the
        _lib.Q_* constants, or None for the process-wide reading"""
        _lib.check(_lib.load().pcx_fir_set_qformat(self._h, _lib.qformat_ptr(q)))

    @property
    def last_algo(self):
        return _lib.load().pcx_fir_last_algo(self._h)

    def set_slots(self, slots):
        """pcx_fir_set_slots: the share of the device's 1024 resident workgroup slots this handle's launches take"""
        _lib.check(_lib.load().pcx_fir_set_slots(self._h, int(slots)))

    def geometry(self):
        k, r = C.c_size_t(), C.c_size_t()
        _lib.check(_lib.load().pcx_fir_get_geometry(self._h, C.byref(k), C.byref(r)))
        return k.value, r.value

    @property
    def K(self):
        return self.geometry()[0]

    def process(self, inbuf, out_cap):
        """Host buffers.  Returns (out[:produced], consumed, produced)."""
        x = as_pairs(inbuf)
        n_in = x.shape[0]
        shape = (out_cap, 2) if self.is_complex else (out_cap,)
        y = np.zeros(shape, dtype=NP_SCALAR[self.scalar])
        c, p = C.c_size_t(), C.c_size_t()
        _lib.check(_lib.load().pcx_fir_process(self._h, _np_ptr(x), n_in, _np_ptr(y), out_cap, C.byref(c), C.byref(p)))
        return y[:p.value], c.value, p.value

    def process_dev(self, x, y, in_elems=None, out_cap=None, stream=None):
        """Device tensors (x: in_elems elements incl. K-1 history).  Returns (consumed, produced)."""
        w = 2 if self.is_complex else 1
        if in_elems is None:
            in_elems = x.numel() // w
        if out_cap is None:
            out_cap = y.numel() // w
        c, p = C.c_size_t(), C.c_size_t()
        _lib.check(_lib.load().pcx_fir_process_dev(self._h, _dev_ptr(x), in_elems, _dev_ptr(y), out_cap,
                                                   C.byref(c), C.byref(p), _stream_ptr(stream)))
        return c.value, p.value

    def process_dev_gated(self, x, y, gate, value, in_elems=None, out_cap=None, stream=None):
        """pcx_fir_process_dev_gated: one launch over a shard whose first K-1 samples (the halo) are still on their way; the blocks
        that read them wait until the 32-bit word `gate` (a device tensor) has reached `value`.  Returns (consumed, produced, gated);
        gated == False: nothing was queued (no gated kernel for this configuration)."""
        w = 2 if self.is_complex else 1
        if in_elems is None:
            in_elems = x.numel() // w
        if out_cap is None:
            out_cap = y.numel() // w
        c, p, g = C.c_size_t(), C.c_size_t(), C.c_int()
        _lib.check(_lib.load().pcx_fir_process_dev_gated(self._h, _dev_ptr(x), in_elems, _dev_ptr(y), out_cap, C.byref(c), C.byref(p),
                                                         _gate_ptr(gate), value & 0xFFFFFFFF, _stream_ptr(stream), C.byref(g)))
        return c.value, p.value, bool(g.value)


def gate_signal(gate, value, stream=None):
    """pcx_gate_signal_dev: a one-thread kernel on `stream` (default: torch's current one) that sets the gate word to `value`."""
    _lib.check(_lib.load().pcx_gate_signal_dev(_dev_ptr(gate), value & 0xFFFFFFFF, _stream_ptr(stream)))


def gate_signal_host(gate, value):
    """Open a gate that lives in page-locked HOST memory: a plain 32-bit store, no stream and no hardware queue behind it -- the way
    to signal LATE (behind the gated launch), where a stream's signal kernel could sit in the launch's own queue (include/pcx.h)."""
    if gate.is_cuda or not gate.is_pinned():
        raise ValueError("gate_signal_host needs a page-locked host tensor")
    C.c_uint32.from_address(gate.data_ptr()).value = value & 0xFFFFFFFF


class Fft(_Handle):
    """pcx_fft_*: FFTAux::transform (fft/FFTAux.h) over whole frames."""
    _destroy = "pcx_fft_destroy"

    def __init__(self, dtype, num_bins, inverse=False):
        super().__init__()
        self.scalar, cplx = parse_dtype(dtype)
        if not cplx:
            raise _lib.InvalidArgument(_lib.ERR_ARG, "FFTFactory: unsupported type")
        self.num_bins = int(num_bins)
        _lib.check(_lib.load().pcx_fft_create(self.scalar, self.num_bins, int(bool(inverse)), C.byref(self._h)))

    def transform(self, x, nframes=None):
        xp = as_pairs(x)
        nf = xp.shape[0] // self.num_bins if nframes is None else nframes
        y = np.zeros_like(xp[:nf * self.num_bins])
        _lib.check(_lib.load().pcx_fft_transform(self._h, _np_ptr(xp), _np_ptr(y), nf))
        return y

    def transform_dev(self, x, y, nframes, stream=None):
        _lib.check(_lib.load().pcx_fft_transform_dev(self._h, _dev_ptr(x), _dev_ptr(y), nframes, _stream_ptr(stream)))


class FreqDemod(_Handle):
    """pcx_freqdemod_*: demod/FreqDemod.cpp:44-71 with _prev carried on the device."""
    _destroy = "pcx_freqdemod_destroy"

    def __init__(self, dtype="complex_float32"):
        super().__init__()
        self.scalar, cplx = parse_dtype(dtype)
        if not cplx:
            raise _lib.InvalidArgument(_lib.ERR_ARG, "FreqDemodFactory: unsupported types")
        _lib.check(_lib.load().pcx_freqdemod_create(self.scalar, C.byref(self._h)))

    def reset(self):
        _lib.check(_lib.load().pcx_freqdemod_reset(self._h))

    def process(self, x, out=None):
        xp = as_pairs(x)
        y = np.zeros(xp.shape[0], dtype=NP_SCALAR[self.scalar]) if out is None else out
        _lib.check(_lib.load().pcx_freqdemod_process(self._h, _np_ptr(xp), _np_ptr(y), xp.shape[0]))
        return y

    def process_dev(self, x, y, n, stream=None):
        _lib.check(_lib.load().pcx_freqdemod_process_dev(self._h, _dev_ptr(x), _dev_ptr(y), n, _stream_ptr(stream)))


class FmChain(_Handle):
    """pcx_fmchain_*: Rotate -> FIR -> FreqDemod in one kernel (complex_float32 -> float32)."""
    _destroy = "pcx_fmchain_destroy"

    def __init__(self):
        super().__init__()
        _lib.check(_lib.load().pcx_fmchain_create(C.byref(self._h)))

    def set_phase(self, phase):
        _lib.check(_lib.load().pcx_fmchain_set_phase(self._h, float(phase)))

    def set_taps(self, taps, complex_taps=None):
        t = np.asarray(taps)
        if complex_taps is None:
            complex_taps = np.iscomplexobj(t)
        if complex_taps:
            t = np.ascontiguousarray(t.astype(np.complex128)).view(np.float64)
            n = t.size // 2
        else:
            t = np.ascontiguousarray(np.real(t).astype(np.float64))
            n = t.size
        _lib.check(_lib.load().pcx_fmchain_set_taps(self._h, _np_ptr(t), n, int(bool(complex_taps))))

    def reset(self):
        _lib.check(_lib.load().pcx_fmchain_reset(self._h))

    def set_algo(self, algo):
        _lib.check(_lib.load().pcx_fmchain_set_algo(self._h, algo))

    @property
    def last_algo(self):
        return _lib.load().pcx_fmchain_last_algo(self._h)

    def set_slots(self, slots):
        _lib.check(_lib.load().pcx_fmchain_set_slots(self._h, int(slots)))

    def process(self, x, out_cap):
        xp = as_pairs(x)
        y = np.zeros(out_cap, dtype=np.float32)
        c, p = C.c_size_t(), C.c_size_t()
        _lib.check(_lib.load().pcx_fmchain_process(self._h, _np_ptr(xp), xp.shape[0], _np_ptr(y), out_cap, C.byref(c), C.byref(p)))
        return y[:p.value], c.value, p.value

    def process_dev(self, x, y, in_elems, out_cap, stream=None):
        c, p = C.c_size_t(), C.c_size_t()
        _lib.check(_lib.load().pcx_fmchain_process_dev(self._h, _dev_ptr(x), in_elems, _dev_ptr(y), out_cap,
                                                       C.byref(c), C.byref(p), _stream_ptr(stream)))
        return c.value, p.value

    def process_dev_gated(self, x, y, gate, value, in_elems, out_cap, stream=None):
        """pcx_fmchain_process_dev_gated (see FirFilter.process_dev_gated; the halo is K samples).  Returns (consumed, produced, gated)."""
        c, p, g = C.c_size_t(), C.c_size_t(), C.c_int()
        _lib.check(_lib.load().pcx_fmchain_process_dev_gated(self._h, _dev_ptr(x), in_elems, _dev_ptr(y), out_cap, C.byref(c), C.byref(p),
                                                             _gate_ptr(gate), value & 0xFFFFFFFF, _stream_ptr(stream), C.byref(g)))
        return c.value, p.value, bool(g.value)


# ---- stateless maps ------------------------------------------------------------------
def _map(host_fn, dev_fn, scalar_args, x, out_shape_fn, n, out=None, stream=None):
    L = _lib.load()
    if _is_torch(x):
        _lib.check(getattr(L, dev_fn)(*scalar_args, _dev_ptr(x), _dev_ptr(out), n, _stream_ptr(stream)))
        return out
    y = np.zeros(out_shape_fn(x), dtype=x.dtype) if out is None else out
    _lib.check(getattr(L, host_fn)(*scalar_args, _np_ptr(x), _np_ptr(y), n))
    return y


_libm = None


def _polar(phase):
    """std::polar(1.0, phase) as an optimised C++ build evaluates it (Rotate.cpp:74): GCC merges the cos and
    sin of one argument into a single glibc sincos() call, whose sine differs from sin() in the last bit for
    some arguments (e.g. 2.747554270528532) -- and so do numpy's own vector routines.  The block layer
    (comms_blocks.cpp) gets this for free; the Python wrapper asks libm for the same call."""
    global _libm
    if _libm is None:
        try:
            _libm = C.CDLL("libm.so.6")
            _libm.sincos.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
            _libm.sincos.restype = None
        except (OSError, AttributeError):
            _libm = False
    if not _libm:
        return math.cos(phase), math.sin(phase)
    s, c = C.c_double(), C.c_double()
    _libm.sincos(phase, C.byref(s), C.byref(c))
    return c.value, s.value


def set_qformat(q=None):
    """pcx_set_qformat: the process-wide floatToQ / fromQ reading (a (frac, float_to_q, from_q) triple; None: the built-in default)"""
    _lib.check(_lib.load().pcx_set_qformat(_lib.qformat_ptr(q)))


def get_qformat():
    q = _lib.QFormat()
    _lib.check(_lib.load().pcx_get_qformat(C.byref(q)))
    return (q.frac, q.float_to_q, q.from_q)


def rotate(x, phase, scalar=None, out=None, n=None, stream=None, qformat=None):
    """arrayRotate (math/Rotate.cpp:15-23).  phase=None: block whose setPhase was never called.  qformat: pcx_rotate_q's reading."""
    pr, pi = (0.0, 0.0) if phase is None else _polar(float(phase))
    if not _is_torch(x):
        x = as_pairs(x)
        scalar, n = SCALAR_OF_NP[x.dtype], x.shape[0]
    if qformat is not None:
        return _map("pcx_rotate_q", "pcx_rotate_q_dev", (scalar, pr, pi, _lib.qformat_ptr(qformat)), x, lambda a: a.shape, n, out, stream)
    return _map("pcx_rotate", "pcx_rotate_dev", (scalar, pr, pi), x, lambda a: a.shape, n, out, stream)


def scale(x, factor, is_complex, scalar=None, out=None, n=None, stream=None, qformat=None):
    if not _is_torch(x):
        x = as_pairs(x)
        scalar, n = SCALAR_OF_NP[x.dtype], x.shape[0]
    if qformat is not None:
        return _map("pcx_scale_q", "pcx_scale_q_dev", (scalar, int(is_complex), float(factor), _lib.qformat_ptr(qformat)), x, lambda a: a.shape, n, out, stream)
    return _map("pcx_scale", "pcx_scale_dev", (scalar, int(is_complex), float(factor)), x, lambda a: a.shape, n, out, stream)


def abs_(x, is_complex, scalar=None, out=None, n=None, stream=None):
    if not _is_torch(x):
        x = as_pairs(x)
        scalar, n = SCALAR_OF_NP[x.dtype], x.shape[0]
    return _map("pcx_abs", "pcx_abs_dev", (scalar, int(is_complex)), x, lambda a: (a.shape[0],), n, out, stream)


def conj(x, scalar=None, out=None, n=None, stream=None):
    if not _is_torch(x):
        x = as_pairs(x)
        scalar, n = SCALAR_OF_NP[x.dtype], x.shape[0]
    return _map("pcx_conj", "pcx_conj_dev", (scalar,), x, lambda a: a.shape, n, out, stream)


def angle(x, scalar=None, out=None, n=None, stream=None):
    """getAngle per element (math/Angle.cpp): complex in, real out."""
    if not _is_torch(x):
        x = as_pairs(x)
        scalar, n = SCALAR_OF_NP[x.dtype], x.shape[0]
    return _map("pcx_angle", "pcx_angle_dev", (scalar,), x, lambda a: (a.shape[0],), n, out, stream)


def arith(op, a, b, is_complex, scalar=None, out=None, n=None, stream=None):
    """out[i] = a[i] OP b[i], OP in "ADD"/"SUB"/"MUL"/"DIV" (math/Arithmetic.cpp:70-110, factory :279-297).
    torch operands: device buffers (out may be a or b); numpy: host path."""
    L = _lib.load()
    if op not in ARITH_OPS:
        raise _lib.InvalidArgument(_lib.ERR_ARG, "arithmeticFactory: unsupported args (operation %r)" % (op,))
    if _is_torch(a):
        _lib.check(L.pcx_arith_dev(scalar, int(is_complex), ARITH_OPS[op], _dev_ptr(a), _dev_ptr(b), _dev_ptr(out), n, _stream_ptr(stream)))
        return out
    a, b = as_pairs(a), as_pairs(b)
    if a.shape != b.shape or a.dtype != b.dtype:
        raise ValueError("operands must match")
    y = np.zeros_like(a) if out is None else out
    _lib.check(L.pcx_arith(SCALAR_OF_NP[a.dtype], int(is_complex), ARITH_OPS[op], _np_ptr(a), _np_ptr(b), _np_ptr(y), a.shape[0]))
    return y


def split_complex(x, scalar=None, re=None, im=None, n=None, stream=None):
    """arraySplitComplex (utility/SplitComplex.cpp:10-18): (n, 2) interleaved -> two planes."""
    L = _lib.load()
    if _is_torch(x):
        _lib.check(L.pcx_split_complex_dev(scalar, _dev_ptr(x), _dev_ptr(re), _dev_ptr(im), n, _stream_ptr(stream)))
        return re, im
    x = as_pairs(x)
    re = np.zeros(x.shape[0], dtype=x.dtype) if re is None else re
    im = np.zeros(x.shape[0], dtype=x.dtype) if im is None else im
    _lib.check(L.pcx_split_complex(SCALAR_OF_NP[x.dtype], _np_ptr(x), _np_ptr(re), _np_ptr(im), x.shape[0]))
    return re, im


def combine_complex(re, im, scalar=None, out=None, n=None, stream=None):
    """arrayCombineComplex (utility/CombineComplex.cpp:10-17)."""
    L = _lib.load()
    if _is_torch(re):
        _lib.check(L.pcx_combine_complex_dev(scalar, _dev_ptr(re), _dev_ptr(im), _dev_ptr(out), n, _stream_ptr(stream)))
        return out
    re, im = np.ascontiguousarray(re), np.ascontiguousarray(im)
    y = np.zeros((re.shape[0], 2), dtype=re.dtype) if out is None else out
    _lib.check(L.pcx_combine_complex(SCALAR_OF_NP[re.dtype], _np_ptr(re), _np_ptr(im), _np_ptr(y), re.shape[0]))
    return y


def fill_uniform_f32_dev(t, seed, offset=0, stream=None):
    """Fill a float32 CUDA tensor with the deterministic synthetic stream (uniform [-1,1))."""
    _lib.check(_lib.load().pcx_fill_uniform_f32_dev(_dev_ptr(t), t.numel(), seed, offset, _stream_ptr(stream)))
    return t


def clock_probe(out, spin_us, stream):
    """pcx_clock_probe_dev: one wave on `stream` (a raw stream pointer or torch stream) spins for spin_us and writes the shader clock in
    MHz to the float32 CUDA tensor `out`."""
    _lib.check(_lib.load().pcx_clock_probe_dev(_dev_ptr(out), int(spin_us), _stream_ptr(stream)))


class NodeStream(_Handle):
    """pcx_shard_*: ONE complex_float32 stream over several devices from ONE process -- per-device FIR handles and
    streams, the tap-length halo exchanged natively by RCCL send/recv (or peer copies), include/pcx.h."""
    _destroy = "pcx_shard_destroy"
    RCCL, PEER_COPY = 0, 1

    def __init__(self, devices, transport=0):
        super().__init__()
        devs = (C.c_int * len(devices))(*devices)
        _lib.check(_lib.load().pcx_shard_create(len(devices), devs, transport, C.byref(self._h)))
        self.nshards = len(devices)

    def set_taps(self, taps, complex_taps=True):
        t = np.asarray(taps)
        if complex_taps:
            t = np.ascontiguousarray(t.astype(np.complex128)).view(np.float64)
            n = t.size // 2
        else:
            t = np.ascontiguousarray(np.real(t).astype(np.float64))
            n = t.size
        _lib.check(_lib.load().pcx_shard_set_taps(self._h, t.ctypes.data_as(C.POINTER(C.c_double)), n, int(complex_taps)))
        self.K = n

    def set_algo(self, algo):
        _lib.check(_lib.load().pcx_shard_set_algo(self._h, algo))

    def set_gated(self, enable):
        """pcx_shard_set_gated: False = two launches per shard and pass (body, halo event, head) instead of one gated launch"""
        _lib.check(_lib.load().pcx_shard_set_gated(self._h, int(bool(enable))))

    def set_submit_threads(self, enable):
        """pcx_shard_set_submit_threads: one thread per device queues that device's share of a pass"""
        _lib.check(_lib.load().pcx_shard_set_submit_threads(self._h, int(bool(enable))))

    def set_chain(self, enable, phase=0.0):
        """pcx_shard_set_chain: the shards run Rotate(phase) -> FIR -> FreqDemod (float32 outputs, halo of K samples)."""
        _lib.check(_lib.load().pcx_shard_set_chain(self._h, int(bool(enable)), float(phase)))
        self.chain = bool(enable)

    def configure(self, shard_elems):
        _lib.check(_lib.load().pcx_shard_configure(self._h, shard_elems))
        self.C = shard_elems

    def info(self):
        """pcx_shard_info -> (nshards, K, shard_elems, transport) as the handle holds them"""
        n, K, Ce, t = C.c_int(), C.c_size_t(), C.c_size_t(), C.c_int()
        _lib.check(_lib.load().pcx_shard_info(self._h, C.byref(n), C.byref(K), C.byref(Ce), C.byref(t)))
        return n.value, K.value, Ce.value, t.value

    def buffers(self, g):
        """(in_dev, out_dev, stream, device) of shard g as integers."""
        i, o, s, d = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int()
        _lib.check(_lib.load().pcx_shard_buffers(self._h, g, C.byref(i), C.byref(o), C.byref(s), C.byref(d)))
        return i.value, o.value, s.value, d.value

    def scatter(self, x):
        xp = as_pairs(x)
        _lib.check(_lib.load().pcx_shard_scatter(self._h, _np_ptr(xp), xp.shape[0]))

    def step(self):
        _lib.check(_lib.load().pcx_shard_step(self._h))

    def post_exchange(self):
        """pcx_shard_post_exchange: the halos of what the shard buffers hold now start travelling (double-buffered streaming over two NodeStreams)"""
        _lib.check(_lib.load().pcx_shard_post_exchange(self._h))

    def compute(self):
        """pcx_shard_compute: the pass whose exchange post_exchange() queued"""
        _lib.check(_lib.load().pcx_shard_compute(self._h))

    def sync(self):
        _lib.check(_lib.load().pcx_shard_sync(self._h))

    def gather(self, out=None):
        shape = (self.nshards * self.C,) if getattr(self, "chain", False) else (self.nshards * self.C, 2)
        y = np.empty(shape, np.float32) if out is None else out
        _lib.check(_lib.load().pcx_shard_gather(self._h, _np_ptr(y), y.shape[0]))
        return y
