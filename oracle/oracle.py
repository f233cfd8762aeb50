"""ctypes front-end of the CPU oracle.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  It wraps oracle/libpcx_oracle.so (the plain-C restatement,
oracle/pcx_oracle.c) and, when present, oracle/_ref/libpcx_ref.so (the
reference's Pothos-free sources compiled where they lie, oracle/ref_driver.cpp).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

# scalar type codes (same numbering as include/pcx.h)
F64, F32, I64, I32, I16, I8, U64, U32, U16, U8 = range(10)   # unsigned: /comms/arithmetic only
NP_SCALAR = {F64: np.float64, F32: np.float32, I64: np.int64, I32: np.int32, I16: np.int16, I8: np.int8,
             U64: np.uint64, U32: np.uint32, U16: np.uint16, U8: np.uint8}
ADD, SUB, MUL, DIV = range(4)
SCALAR_OF_NP = {np.dtype(v): k for k, v in NP_SCALAR.items()}


def build(ref=True):
    """(Re)build the oracle; `_ref` only when /root/reference is present."""
    targets = ["all"] + (["_ref"] if ref else [])
    subprocess.check_call(["make", "-s", "-C", _HERE] + targets)


def _load(path):
    if not os.path.exists(path):
        return None
    return C.CDLL(path)


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        p = os.path.join(_HERE, "libpcx_oracle.so")
        if not os.path.exists(p):
            build(ref=False)
        _lib = C.CDLL(p)
        _declare(_lib)
    return _lib


def ref():
    """The compiled reference (None when oracle/_ref was not built)."""
    global _ref
    if _ref is None:
        _ref = _load(os.path.join(_HERE, "_ref", "libpcx_ref.so"))
        if _ref is not None:
            _ref.ref_fxpt_atan2.restype = C.c_uint16
            _ref.ref_fxpt_atan2.argtypes = [C.c_int16, C.c_int16]
    return _ref


class Label(C.Structure):
    _fields_ = [("id", C.c_char), ("index", C.c_uint64), ("width", C.c_uint64),
                ("length", C.c_uint64), ("has_length", C.c_int)]


def _declare(L):
    vp, sz = C.c_void_p, C.c_size_t
    L.orc_fir_create.restype = vp
    L.orc_fir_create.argtypes = [C.c_int, C.c_int, C.c_int]
    L.orc_fir_destroy.argtypes = [vp]
    L.orc_fir_set_taps.argtypes = [vp, vp, sz]
    L.orc_fir_set_decimation.argtypes = [vp, sz]
    L.orc_fir_set_interpolation.argtypes = [vp, sz]
    L.orc_fir_set_wait_taps.argtypes = [vp, C.c_int]
    L.orc_fir_set_frame_ids.argtypes = [vp, C.c_int, C.c_int]
    L.orc_fir_activate.argtypes = [vp]
    L.orc_fir_K.restype = sz
    L.orc_fir_K.argtypes = [vp]
    L.orc_fir_input_require.restype = sz
    L.orc_fir_input_require.argtypes = [vp]
    L.orc_fir_work.argtypes = [vp, vp, sz, C.POINTER(Label), sz, vp, sz,
                               C.POINTER(sz), C.POINTER(sz), C.POINTER(sz)]
    L.orc_fir_cf32_chunk.argtypes = [vp, vp, vp, sz]
    L.orc_fft_create.restype = vp
    L.orc_fft_create.argtypes = [C.c_int, sz, C.c_int]
    L.orc_fft_destroy.argtypes = [vp]
    L.orc_fft_transform.argtypes = [vp, vp, vp, sz]
    L.orc_fft_work.argtypes = [vp, vp, vp, C.POINTER(sz), C.POINTER(sz)]
    L.orc_fxpt_atan2.restype = C.c_uint16
    L.orc_fxpt_atan2.argtypes = [C.c_int16, C.c_int16]
    L.orc_freqdemod_create.restype = vp
    L.orc_freqdemod_create.argtypes = [C.c_int]
    L.orc_freqdemod_destroy.argtypes = [vp]
    L.orc_freqdemod_activate.argtypes = [vp]
    L.orc_freqdemod_work.argtypes = [vp, vp, vp, sz]
    L.orc_rotate.argtypes = [C.c_int, C.c_double, vp, vp, sz]
    L.orc_rotate_unset.argtypes = [C.c_int, vp, vp, sz]
    L.orc_scale.argtypes = [C.c_int, C.c_int, C.c_double, vp, vp, sz]
    L.orc_coeff_label_scan.restype = sz
    L.orc_coeff_label_scan.argtypes = [sz, vp, vp, sz, C.c_int, C.POINTER(C.c_long)]
    L.orc_abs.argtypes = [C.c_int, C.c_int, vp, vp, sz]
    L.orc_conj.argtypes = [C.c_int, vp, vp, sz]
    L.orc_angle.argtypes = [C.c_int, vp, vp, sz]
    L.orc_fill_uniform_f32.argtypes = [vp, sz, C.c_uint64, C.c_uint64]
    L.orc_arith.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, sz]
    L.orc_split_complex.argtypes = [C.c_int, vp, vp, vp, sz]
    L.orc_combine_complex.argtypes = [C.c_int, vp, vp, vp, sz]


# --------------------------------------------------------------------------- #
# numpy conventions: a complex stream of scalar type T is an array of shape
# (n, 2) of T (interleaved re, im) -- works for integer complex types too.
# complex64 / complex128 arrays are accepted and viewed that way.
# --------------------------------------------------------------------------- #
def as_pairs(a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.complex64:
        return a.view(np.float32).reshape(-1, 2)
    if a.dtype == np.complex128:
        return a.view(np.float64).reshape(-1, 2)
    return a


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def scalar_code(a):
    return SCALAR_OF_NP[np.dtype(a.dtype)]


class Fir:
    """Restatement of the /comms/fir_filter block (filter/FIRFilter.cpp)."""

    def __init__(self, scalar, is_complex, complex_taps):
        self.h = lib().orc_fir_create(scalar, int(is_complex), int(complex_taps))
        if not self.h:
            raise ValueError("unsupported types")  # FIRFilter.cpp:383
        self.scalar, self.is_complex, self.complex_taps = scalar, is_complex, complex_taps

    def __del__(self):
        if getattr(self, "h", None):
            try:
                lib().orc_fir_destroy(self.h)
            except TypeError:     # interpreter shutdown: the module globals are already gone
                pass
            self.h = None

    def set_taps(self, taps):
        t = np.asarray(taps)
        if self.complex_taps:
            t = np.ascontiguousarray(t.astype(np.complex128)).view(np.float64)
            n = t.size // 2
        else:
            t = np.ascontiguousarray(t.astype(np.float64))
            n = t.size
        if lib().orc_fir_set_taps(self.h, _ptr(t), n) != 0:
            raise ValueError("taps cannot be empty")

    def set_decimation(self, m):
        if lib().orc_fir_set_decimation(self.h, m) != 0:
            raise ValueError("decimation cannot be 0")

    def set_interpolation(self, l):
        if lib().orc_fir_set_interpolation(self.h, l) != 0:
            raise ValueError("interpolation cannot be 0")

    def set_wait_taps(self, w):
        lib().orc_fir_set_wait_taps(self.h, int(w))

    def set_frame_ids(self, have_start, have_end):
        lib().orc_fir_set_frame_ids(self.h, int(have_start), int(have_end))

    def activate(self):
        lib().orc_fir_activate(self.h)

    @property
    def K(self):
        return lib().orc_fir_K(self.h)

    @property
    def input_require(self):
        return lib().orc_fir_input_require(self.h)

    def work(self, inbuf, out_elems, labels=()):
        """One work() call.  Returns (out[:produced], consumed, produced, reserve)."""
        x = as_pairs(inbuf)
        n_in = x.shape[0]
        shape = (out_elems, 2) if self.is_complex else (out_elems,)
        y = np.zeros(shape, dtype=NP_SCALAR[self.scalar])
        labs = (Label * max(1, len(labels)))()
        for i, (lid, index, width, length) in enumerate(labels):
            labs[i].id = lid.encode() if isinstance(lid, str) else lid
            labs[i].index, labs[i].width = index, width
            labs[i].has_length = int(length is not None)
            labs[i].length = 0 if length is None else length
        c, p, r = C.c_size_t(), C.c_size_t(), C.c_size_t()
        lib().orc_fir_work(self.h, _ptr(x), n_in, labs, len(labels), _ptr(y), out_elems,
                           C.byref(c), C.byref(p), C.byref(r))
        reserve = None if r.value == C.c_size_t(-1).value else r.value
        return y[:p.value], c.value, p.value, reserve


def fft(x, nbins, inverse=False, nframes=None):
    """orc_fft_transform over whole frames of x ((n,2) pairs or complex array)."""
    xp = as_pairs(x)
    sc = scalar_code(xp)
    h = lib().orc_fft_create(sc, nbins, int(inverse))
    if not h:
        raise ValueError("unsupported type")  # FFT.cpp:92
    try:
        nf = xp.shape[0] // nbins if nframes is None else nframes
        y = np.zeros_like(xp[:nf * nbins])
        lib().orc_fft_transform(h, _ptr(xp), _ptr(y), nf)
    finally:
        lib().orc_fft_destroy(h)
    return y


def ref_fft(x, nbins, inverse=False):
    """The compiled reference kissfft / kiss_fft on the same frames."""
    r = ref()
    xp = as_pairs(x)
    nf = xp.shape[0] // nbins
    y = np.zeros_like(xp[:nf * nbins])
    fn = {F32: r.ref_kissfft_f32, F64: r.ref_kissfft_f64, I16: r.ref_kiss_fft_i16}[scalar_code(xp)]
    fn.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
    fn(nbins, int(inverse), _ptr(xp), _ptr(y), nf)
    return y


class FreqDemod:
    def __init__(self, scalar):
        self.h = lib().orc_freqdemod_create(scalar)
        self.scalar = scalar
        lib().orc_freqdemod_activate(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_freqdemod_destroy(self.h)
            self.h = None

    def activate(self):
        lib().orc_freqdemod_activate(self.h)

    def work(self, x):
        xp = as_pairs(x)
        y = np.zeros(xp.shape[0], dtype=NP_SCALAR[self.scalar])
        lib().orc_freqdemod_work(self.h, _ptr(xp), _ptr(y), xp.shape[0])
        return y


def rotate(x, phase):
    xp = as_pairs(x)
    y = np.zeros_like(xp)
    if phase is None:
        lib().orc_rotate_unset(scalar_code(xp), _ptr(xp), _ptr(y), xp.shape[0])
    else:
        lib().orc_rotate(scalar_code(xp), float(phase), _ptr(xp), _ptr(y), xp.shape[0])
    return y


def scale(x, factor, is_complex):
    xp = as_pairs(x)
    y = np.zeros_like(xp)
    n = xp.shape[0]
    lib().orc_scale(scalar_code(xp), int(is_complex), float(factor), _ptr(xp), _ptr(y), n)
    return y


def abs_(x, is_complex):
    xp = as_pairs(x)
    n = xp.shape[0]
    y = np.zeros(n, dtype=xp.dtype)
    lib().orc_abs(scalar_code(xp), int(is_complex), _ptr(xp), _ptr(y), n)
    return y


def conj(x):
    xp = as_pairs(x)
    y = np.zeros_like(xp)
    lib().orc_conj(scalar_code(xp), _ptr(xp), _ptr(y), xp.shape[0])
    return y


def angle(x):
    xp = as_pairs(x)
    y = np.zeros(xp.shape[0], dtype=xp.dtype)
    lib().orc_angle(scalar_code(xp), _ptr(xp), _ptr(y), xp.shape[0])
    return y


def arith(op, a, b, is_complex):
    """out[i] = a[i] OP b[i] (math/Arithmetic.cpp:70-110); complex operands as (n, 2) arrays."""
    ap, bp = as_pairs(a), as_pairs(b)
    y = np.zeros_like(ap)
    n = ap.shape[0]
    if lib().orc_arith(scalar_code(ap), int(is_complex), op, _ptr(ap), _ptr(bp), _ptr(y), n) != 0:
        raise ValueError("unsupported args")   # Arithmetic.cpp:297
    return y


def ref_arith(op, a, b, is_complex):
    """The same expression evaluated by the C++ operators themselves (oracle/ref_driver.cpp)."""
    ap, bp = as_pairs(a), as_pairs(b)
    y = np.zeros_like(ap)
    fn = ref().ref_std_arith
    fn.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    if fn(scalar_code(ap), int(is_complex), op, _ptr(ap), _ptr(bp), _ptr(y), ap.shape[0]) != 0:
        raise ValueError("unsupported args")
    return y


def split_complex(x):
    xp = as_pairs(x)
    n = xp.shape[0]
    re, im = np.zeros(n, dtype=xp.dtype), np.zeros(n, dtype=xp.dtype)
    lib().orc_split_complex(scalar_code(xp), _ptr(xp), _ptr(re), _ptr(im), n)
    return re, im


def combine_complex(re, im):
    re, im = np.ascontiguousarray(re), np.ascontiguousarray(im)
    y = np.zeros((re.shape[0], 2), dtype=re.dtype)
    lib().orc_combine_complex(scalar_code(re), _ptr(re), _ptr(im), _ptr(y), re.shape[0])
    return y


def coeff_label_scan(elems, label_index, label_match, have_label_id=True):
    li = np.ascontiguousarray(label_index, dtype=np.uint64)
    lm = np.ascontiguousarray(label_match, dtype=np.int32)
    idx = C.c_long()
    n = lib().orc_coeff_label_scan(elems, _ptr(li), _ptr(lm), li.size, int(have_label_id), C.byref(idx))
    return n, idx.value


Q_HALF_Q, Q_HALF_ELEM = 0, 1            # pcx_q_frac
Q_TRUNCATE, Q_NEAREST = 0, 1            # pcx_q_to
Q_FLOOR, Q_TOWARD_ZERO, Q_ROUND = 0, 1, 2   # pcx_q_from
QFORMATS = [(f, t, r) for f in (0, 1) for t in (0, 1) for r in (0, 1, 2)]   # the twelve readings (profiles/r02/qformat_enumeration.txt)


def set_qformat(frac=0, float_to_q=0, from_q=0):
    """The Q-format reading of the integer Fir / rotate / scale restatements (orc_set_qformat; process-wide).  Set it BEFORE creating
    a Fir or setting its taps: taps are quantised when they are set.  set_qformat() restores the default."""
    lib().orc_set_qformat(int(frac), int(float_to_q), int(from_q))


def fill_uniform_f32(n_scalars, seed, offset=0):
    a = np.empty(n_scalars, dtype=np.float32)
    lib().orc_fill_uniform_f32(_ptr(a), n_scalars, seed, offset)
    return a


def ref_angle(x):
    xp = as_pairs(x)
    y = np.zeros(xp.shape[0], dtype=xp.dtype)
    name = {F64: "f64", F32: "f32", I64: "i64", I32: "i32", I16: "i16", I8: "i8"}[scalar_code(xp)]
    fn = getattr(ref(), "ref_angle_" + name)
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    fn(_ptr(xp), _ptr(y), xp.shape[0])
    return y


def ref_abs(x, is_complex):
    xp = as_pairs(x)
    n = xp.shape[0]
    y = np.zeros(n, dtype=xp.dtype)
    name = {F64: "f64", F32: "f32", I64: "i64", I32: "i32", I16: "i16", I8: "i8"}[scalar_code(xp)]
    fn = getattr(ref(), "ref_abs_" + name + ("_cplx" if is_complex else "_real"))
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    fn(_ptr(xp), _ptr(y), n)
    return y
