"""Python face of the host-side block layer (libpcx_blocks.so, include/pcx_blocks.h).

`make(path, dtype, ...)` is BlockRegistry::make for the MI355X-backed /comms blocks; the
returned Block exposes the registered calls by name (`call("setTaps", taps)`), `activate()`
and `work(inbuf, out_elems, labels)` -- one scheduler work() on host buffers, returning what
the block consumed / produced / reserved and the labels it posted.  Used by the tests to show
the blocks behave like the reference blocks; inside Pothos the same C++ blocks load as a
plugin module and this wrapper is not involved.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from .device import NP_SCALAR, as_pairs, parse_dtype

_HERE = os.path.dirname(os.path.abspath(__file__))
BLOCKS_LIB_PATH = os.path.join(_HERE, "libpcx_blocks.so")
NONE, SIZE, DOUBLE, STRING = 0, 1, 2, 3
_SIZE_MAX = C.c_size_t(-1).value


class PcxbLabel(C.Structure):
    _fields_ = [("id", C.c_char * 32), ("index", C.c_uint64), ("width", C.c_uint64), ("kind", C.c_int),
                ("uval", C.c_uint64), ("dval", C.c_double), ("sval", C.c_char * 32)]


_blib = None


def load():
    global _blib
    if _blib is not None:
        return _blib
    _lib.load()    # one HIP runtime per process, libpcx_hip.so first
    path = os.environ.get("PCX_BLOCKS_LIBRARY") or BLOCKS_LIB_PATH     # the override selects the sanitizer build (make asan)
    if not os.path.exists(path):
        raise ImportError("%s is missing: build with make -C pothoscomms_amd/csrc" % path)
    L = C.CDLL(path)
    vp, sz, i, cp = C.c_void_p, C.c_size_t, C.c_int, C.c_char_p
    L.pcxb_last_error.restype = cp
    L.pcxb_registry_has.argtypes = [cp]
    L.pcxb_registry_count.restype = sz
    L.pcxb_registry_path.restype = cp
    L.pcxb_registry_path.argtypes = [sz]
    L.pcxb_registry_arity.restype = C.c_long
    L.pcxb_registry_arity.argtypes = [cp]
    L.pcxb_call_count.argtypes = [vp, C.POINTER(sz)]
    L.pcxb_call_name.argtypes = [vp, sz, cp, sz]
    L.pcxb_call_arity.restype = C.c_long
    L.pcxb_call_arity.argtypes = [vp, cp]
    L.pcxb_make.argtypes = [cp, cp, sz, cp, sz, i, C.POINTER(vp)]
    L.pcxb_destroy.argtypes = [vp]
    L.pcxb_call_double.argtypes = [vp, cp, C.c_double]
    L.pcxb_call_size.argtypes = [vp, cp, sz]
    L.pcxb_call_bool.argtypes = [vp, cp, i]
    L.pcxb_call_string.argtypes = [vp, cp, cp]
    L.pcxb_call_taps.argtypes = [vp, cp, vp, sz, i]
    L.pcxb_get_double.argtypes = [vp, cp, C.POINTER(C.c_double)]
    L.pcxb_get_size.argtypes = [vp, cp, C.POINTER(sz)]
    L.pcxb_get_bool.argtypes = [vp, cp, C.POINTER(i)]
    L.pcxb_get_string.argtypes = [vp, cp, cp, sz]
    L.pcxb_get_taps.argtypes = [vp, cp, vp, sz, C.POINTER(sz), i]
    L.pcxb_activate.argtypes = [vp]
    L.pcxb_deactivate.argtypes = [vp]
    L.pcxb_connect_signal.argtypes = [vp, cp, vp, cp]
    L.pcxb_port_dtype.argtypes = [vp, i, cp, sz, C.POINTER(sz), C.POINTER(sz)]
    L.pcxb_buffer_manager.argtypes = [vp, i, cp, sz, C.POINTER(sz)]
    L.pcxb_initial_reserve.argtypes = [vp, C.POINTER(sz)]
    L.pcxb_acquire_buffer.argtypes = [vp, i, sz, C.POINTER(vp), C.POINTER(sz), C.POINTER(i)]
    L.pcxb_link_buffer.argtypes = [vp, vp, sz, C.POINTER(vp), C.POINTER(sz), C.POINTER(i)]
    L.pcxb_work_loop.argtypes = [vp, vp, sz, vp, sz, sz, C.POINTER(C.c_double), C.POINTER(sz), C.POINTER(sz)]
    L.pcxb_circular_create.argtypes = [sz, C.POINTER(vp), C.POINTER(sz)]
    L.pcxb_circular_destroy.argtypes = [vp]
    L.pcxb_call_sizes.argtypes = [vp, cp, C.POINTER(sz), sz]
    L.pcxb_get_sizes.argtypes = [vp, cp, C.POINTER(sz), sz, C.POINTER(sz)]
    L.pcxb_num_ports.argtypes = [vp, i, C.POINTER(sz)]
    L.pcxb_port_info.argtypes = [vp, i, sz, cp, sz, cp, sz, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz)]
    L.pcxb_work_ports.argtypes = [vp, sz, C.POINTER(vp), C.POINTER(sz), sz, C.POINTER(vp), C.POINTER(sz), C.POINTER(sz), C.POINTER(sz)]
    L.pcxb_work.argtypes = [vp, vp, sz, C.POINTER(PcxbLabel), sz, vp, sz, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz),
                            C.POINTER(PcxbLabel), sz, C.POINTER(sz)]
    _blib = L
    return L


def _check(rc):
    if rc == 0:
        return
    msg = load().pcxb_last_error().decode("utf-8", "replace")
    if rc == _lib.ERR_ARG:
        raise _lib.InvalidArgument(rc, msg)
    if rc == _lib.ERR_UNSUPPORTED:
        raise _lib.Unsupported(rc, msg)
    raise _lib.PcxError(rc, msg)


def registry_paths():
    L = load()
    return sorted(L.pcxb_registry_path(i).decode() for i in range(L.pcxb_registry_count()))


def registry_arity(path):
    """how many arguments the factory registered at `path` takes (-1: no such path)"""
    return int(load().pcxb_registry_arity(path.encode()))


class Label:
    """(id, index, width, data) -- data is None, an int (size_t), a float or a str."""

    def __init__(self, id, index, data=None, width=1):
        self.id, self.index, self.data, self.width = id, int(index), data, int(width)

    def __repr__(self):
        return "Label(%r, index=%d, data=%r, width=%d)" % (self.id, self.index, self.data, self.width)

    def __eq__(self, o):
        return (self.id, self.index, self.data, self.width) == (o.id, o.index, o.data, o.width)

    def _to_c(self, c):
        c.id = self.id.encode()
        c.index, c.width = self.index, self.width
        if self.data is None:
            c.kind = NONE
        elif isinstance(self.data, bool) or isinstance(self.data, (int, np.integer)):
            c.kind, c.uval = SIZE, int(self.data)
        elif isinstance(self.data, float):
            c.kind, c.dval = DOUBLE, self.data
        else:
            c.kind, c.sval = STRING, str(self.data).encode()

    @staticmethod
    def _from_c(c):
        data = None
        if c.kind == SIZE:
            data = int(c.uval)
        elif c.kind == DOUBLE:
            data = float(c.dval)
        elif c.kind == STRING:
            data = c.sval.decode()
        return Label(c.id.decode(), c.index, data, c.width)


class Block:
    def __init__(self, path, dtype, *args, dimension=1):
        L = load()
        self.path = path
        self.dtype = dtype
        sarg, nbins, inverse = None, 0, 0
        if path.endswith("fir_filter") or path.endswith("/arithmetic") or path.endswith("fm_demod_chain"):
            sarg = (args[0] if args else "").encode()     # tapsType resp. operation
        elif path == "/comms/fft":
            nbins, inverse = int(args[0]), int(bool(args[1])) if len(args) > 1 else 0
        self._h = C.c_void_p()
        _check(L.pcxb_make(path.encode(), (dtype or "").encode(), dimension, sarg, nbins, inverse, C.byref(self._h)))
        self._slots = []          # blocks wired to this one's signals: kept alive as long as it can emit
        if path.endswith("fir_designer"):     # no stream ports: a signal source only
            self.in_dtype = self.out_dtype = None
            self.in_dim = self.out_dim = 0
            return
        self.in_dtype, self.in_dim, _ = self._port(0)
        self.out_dtype, self.out_dim, _ = self._port(1)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            load().pcxb_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _port(self, is_output):
        name = C.create_string_buffer(64)
        dim, nbytes = C.c_size_t(), C.c_size_t()
        _check(load().pcxb_port_dtype(self._h, is_output, name, 64, C.byref(dim), C.byref(nbytes)))
        return name.value.decode(), dim.value, nbytes.value

    # ---- registered calls ----
    def call(self, name, *args):
        L, n = load(), name.encode()
        if name == "setTaps":
            t = np.asarray(args[0])
            cplx = np.iscomplexobj(t)
            flat = np.ascontiguousarray(t.astype(np.complex128)).view(np.float64) if cplx else np.ascontiguousarray(t.astype(np.float64))
            return _check(L.pcxb_call_taps(self._h, n, flat.ctypes.data_as(C.c_void_p), t.size, int(cplx)))
        if name == "getTaps":
            cplx = bool(args[0]) if args else False
            buf = np.zeros(1 << 16, np.float64)
            cnt = C.c_size_t()
            _check(L.pcxb_get_taps(self._h, n, buf.ctypes.data_as(C.c_void_p), buf.size, C.byref(cnt), int(cplx)))
            return buf[:2 * cnt.value].view(np.complex128).copy() if cplx else buf[:cnt.value].copy()
        if name in ("setWindowArgs", "setFrequencies"):      # std::vector<double>
            flat = np.ascontiguousarray(np.asarray(args[0], dtype=np.float64))
            return _check(L.pcxb_call_taps(self._h, n, flat.ctypes.data_as(C.c_void_p), flat.size, 0))
        if name == "windowArgs":
            buf, cnt = np.zeros(64, np.float64), C.c_size_t()
            _check(L.pcxb_get_taps(self._h, n, buf.ctypes.data_as(C.c_void_p), buf.size, C.byref(cnt), 0))
            return buf[:cnt.value].copy()
        if name in ("setPreload", "setDevices"):
            v = (C.c_size_t * max(1, len(args[0])))(*[int(a) for a in args[0]])
            return _check(L.pcxb_call_sizes(self._h, n, v, len(args[0])))
        if name in ("preload", "getDevices"):
            v, cnt = (C.c_size_t * 64)(), C.c_size_t()
            _check(L.pcxb_get_sizes(self._h, n, v, 64, C.byref(cnt)))
            return [int(v[k]) for k in range(cnt.value)]
        if not args:   # getter
            if name in ("getDecimation", "getInterpolation", "getNumInlineBuffers", "numTaps", "getShardPasses", "getDevice", "getPortSlabBytes"):
                v = C.c_size_t()
                _check(L.pcxb_get_size(self._h, n, C.byref(v)))
                return v.value
            if name in ("getWaitTaps",):
                v = C.c_int()
                _check(L.pcxb_get_bool(self._h, n, C.byref(v)))
                return bool(v.value)
            if name in ("getPhase", "getFactor", "sampleRate", "frequencyLower", "frequencyUpper", "bandwidthTrans", "alpha",
                        "stopDB", "passDB", "gain"):
                v = C.c_double()
                _check(L.pcxb_get_double(self._h, n, C.byref(v)))
                return v.value
            s = C.create_string_buffer(128)
            _check(L.pcxb_get_string(self._h, n, s, 128))
            return s.value.decode()
        a = args[0]
        if isinstance(a, bool):
            return _check(L.pcxb_call_bool(self._h, n, int(a)))
        if isinstance(a, (int, np.integer)):
            return _check(L.pcxb_call_size(self._h, n, int(a)))
        if isinstance(a, float):
            return _check(L.pcxb_call_double(self._h, n, a))
        return _check(L.pcxb_call_string(self._h, n, str(a).encode()))

    def calls(self):
        """{name: number of arguments} of the calls the block registered"""
        L = load()
        n = C.c_size_t()
        _check(L.pcxb_call_count(self._h, C.byref(n)))
        out = {}
        buf = C.create_string_buffer(128)
        for i in range(n.value):
            _check(L.pcxb_call_name(self._h, i, buf, len(buf)))
            out[buf.value.decode()] = int(L.pcxb_call_arity(self._h, buf.value))
        return out

    def activate(self):
        _check(load().pcxb_activate(self._h))

    def deactivate(self):
        _check(load().pcxb_deactivate(self._h))

    def connect_signal(self, signal, dst, slot):
        """Topology::connect(self, signal, dst, slot): later emissions call dst's registered `slot` synchronously."""
        _check(load().pcxb_connect_signal(self._h, signal.encode(), dst._h, slot.encode()))
        self._slots.append(dst)

    def buffer_manager(self, is_output):
        name = C.create_string_buffer(64)
        sz = C.c_size_t()
        _check(load().pcxb_buffer_manager(self._h, int(is_output), name, 64, C.byref(sz)))
        return name.value.decode(), sz.value

    def port_buffer(self, is_output, shape, dtype):
        """The next port buffer from the manager the block handed the scheduler (pcxb_acquire_buffer), as a numpy
        array of `shape` / `dtype` over the slab -- page-locked for the device-backed blocks.  Returns (array, pinned)."""
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p, sz, pin = C.c_void_p(), C.c_size_t(), C.c_int()
        _check(load().pcxb_acquire_buffer(self._h, int(is_output), nbytes, C.byref(p), C.byref(sz), C.byref(pin)))
        arr = np.ctypeslib.as_array((C.c_char * nbytes).from_address(p.value)).view(dtype).reshape(shape)
        return arr, bool(pin.value)

    def link_buffer(self, downstream, nbytes):
        """The buffer a scheduler would plant on the edge self.output(0) -> downstream.input(0) (pcxb_link_buffer).
        Returns (address, kind): kind 2 = device memory (two blocks of this module), 1 = page-locked host, 0 = pageable."""
        p, sz, kind = C.c_void_p(), C.c_size_t(), C.c_int()
        _check(load().pcxb_link_buffer(self._h, downstream._h, nbytes, C.byref(p), C.byref(sz), C.byref(kind)))
        return p.value, kind.value

    def work_raw(self, in_ptr, in_elems, out_ptr, out_elems, labels=()):
        """One work() call on raw buffer addresses (host or device memory).  Returns (consumed, produced, reserve)."""
        labs, posted = (PcxbLabel * max(1, len(labels)))(), (PcxbLabel * 64)()
        for i, l in enumerate(labels):
            l._to_c(labs[i])
        c, p, r, npost = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t()
        _check(load().pcxb_work(self._h, C.c_void_p(in_ptr), in_elems, labs, len(labels), C.c_void_p(out_ptr), out_elems, C.byref(c), C.byref(p),
                                C.byref(r), posted, 64, C.byref(npost)))
        return c.value, p.value, (None if r.value == _SIZE_MAX else r.value)

    def work_loop(self, in_ptr, in_elems, out_ptr, out_elems, reps):
        """pcxb_work_loop: `reps` work() calls on the same buffers from native code.  Returns (seconds, consumed, produced) of the loop /
        its last call."""
        t, c, p = C.c_double(), C.c_size_t(), C.c_size_t()
        _check(load().pcxb_work_loop(self._h, C.c_void_p(in_ptr), in_elems, C.c_void_p(out_ptr), out_elems, reps, C.byref(t), C.byref(c), C.byref(p)))
        return t.value, c.value, p.value

    def initial_reserve(self):
        r = C.c_size_t()
        load().pcxb_initial_reserve(self._h, C.byref(r))
        return None if r.value == _SIZE_MAX else r.value

    def ports(self, is_output):
        """[(name, dtype name, dimension, bytes per element, preloaded elements)] -- indexed ports first."""
        L = load()
        cnt = C.c_size_t()
        _check(L.pcxb_num_ports(self._h, int(is_output), C.byref(cnt)))
        out = []
        for k in range(cnt.value):
            nm, dt = C.create_string_buffer(64), C.create_string_buffer(64)
            dim, nb, pre = C.c_size_t(), C.c_size_t(), C.c_size_t()
            _check(L.pcxb_port_info(self._h, int(is_output), k, nm, 64, dt, 64, C.byref(dim), C.byref(nb), C.byref(pre)))
            out.append((nm.value.decode(), dt.value.decode(), dim.value, nb.value, pre.value))
        return out

    def work_ports(self, ins, out_elems, inline=False):
        """One work() call with a buffer planted on every port (ins: one array per input port, out_elems:
        room per output port).  inline=True plants input 0's buffer as output 0 (the buffer forwarding the
        reference's setReadBeforeWrite enables).  Returns (outs trimmed to produced, consumed, produced)."""
        L = load()
        ip, op = self.ports(0), self.ports(1)
        xs = [np.ascontiguousarray(as_pairs(x)) for x in ins]
        if isinstance(out_elems, int):
            out_elems = [out_elems] * len(op)
        ys = []
        for (nm, dt, dim, nb, _), ne in zip(op, out_elems):
            scalar, cplx = parse_dtype(dt)
            ys.append(np.zeros([ne * dim] + ([2] if cplx else []), dtype=NP_SCALAR[scalar]))
        if inline:
            ys[0] = xs[0]
        in_ptrs = (C.c_void_p * len(xs))(*[x.ctypes.data for x in xs])
        in_n = (C.c_size_t * len(xs))(*[x.shape[0] // p[2] for x, p in zip(xs, ip)])
        out_ptrs = (C.c_void_p * len(ys))(*[y.ctypes.data for y in ys])
        out_n = (C.c_size_t * len(ys))(*[int(n) for n in out_elems])
        cons, prod = (C.c_size_t * len(xs))(), (C.c_size_t * len(ys))()
        _check(L.pcxb_work_ports(self._h, len(xs), in_ptrs, in_n, len(ys), out_ptrs, out_n, cons, prod))
        produced = [int(v) for v in prod]
        return [y[:p * port[2]] for y, p, port in zip(ys, produced, op)], [int(v) for v in cons], produced

    def work(self, inbuf, out_elems, labels=(), outbuf=None):
        """One work() call.  Returns (out[:produced], consumed, produced, reserve, posted_labels).
        outbuf: write into this array (e.g. a slab from port_buffer) instead of a fresh one."""
        x = as_pairs(inbuf)
        scalar, cplx = parse_dtype(self.out_dtype)
        shape = [out_elems * self.out_dim] + ([2] if cplx else [])
        if outbuf is None:
            y = np.zeros(shape, dtype=NP_SCALAR[scalar])
        else:
            y = outbuf
            assert y.flags.c_contiguous and y.dtype == NP_SCALAR[scalar] and y.size >= int(np.prod(shape))
        in_elems = x.shape[0] // self.in_dim
        labs = (PcxbLabel * max(1, len(labels)))()
        for i, l in enumerate(labels):
            l._to_c(labs[i])
        posted = (PcxbLabel * 64)()
        c, p, r, npost = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t()
        _check(load().pcxb_work(self._h, x.ctypes.data_as(C.c_void_p), in_elems, labs, len(labels),
                                y.ctypes.data_as(C.c_void_p), out_elems, C.byref(c), C.byref(p), C.byref(r),
                                posted, 64, C.byref(npost)))
        reserve = None if r.value == _SIZE_MAX else r.value
        return (y[:p.value * self.out_dim], c.value, p.value, reserve,
                [Label._from_c(posted[i]) for i in range(min(npost.value, 64))])


class CircularBuffer:
    """The framework's "circular" buffer (pcxb_circular_create): `size` bytes of PAGEABLE shared memory mapped twice back to back --
    view(off, n) for any off < size and n <= size is contiguous, running across the wrap into the second mapping.  What Pothos hands a
    block that asks for BufferManager::make("circular") (filter/FIRFilter.cpp:196-199).  close() it after the blocks that saw it."""

    def __init__(self, nbytes):
        p, n = C.c_void_p(), C.c_size_t()
        _check(load().pcxb_circular_create(nbytes, C.byref(p), C.byref(n)))
        self.base, self.size = p.value, n.value

    def view(self, offset, nbytes, dtype=np.uint8):
        assert 0 <= offset < self.size and 0 <= nbytes <= self.size
        return np.ctypeslib.as_array((C.c_char * nbytes).from_address(self.base + offset)).view(dtype)

    def close(self):
        if getattr(self, "base", None):
            _check(load().pcxb_circular_destroy(C.c_void_p(self.base)))
            self.base = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make(path, dtype=None, *args, dimension=1):
    """BlockRegistry::make(path, dtype, *args)."""
    return Block(path, dtype, *args, dimension=dimension)
