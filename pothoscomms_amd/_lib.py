"""ctypes binding of libpcx_hip.so (include/pcx.h).

The library is the product's only compute path.  If it is missing or fails to
load, importing the binding raises -- there is no CPU fallback anywhere in
this package (the CPU oracle under oracle/ is test infrastructure only).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpcx_hip.so")

# pcx_scalar
F64, F32, I64, I32, I16, I8, U64, U32, U16, U8 = range(10)   # unsigned: pcx_arith* only
ARITH_ADD, ARITH_SUB, ARITH_MUL, ARITH_DIV = range(4)
# pcx_status
OK, ERR_ARG, ERR_UNSUPPORTED, ERR_HIP, ERR_STATE = 0, -1, -2, -3, -4
# pcx_fir_algo
FIR_AUTO, FIR_DIRECT, FIR_OLS_FFT, FIR_EXACT = 0, 1, 2, 3


class PcxError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("pcx status %d: %s" % (status, msg))
        self.status = status


class InvalidArgument(PcxError, ValueError):
    """Mirror of Pothos::InvalidArgumentException at the block boundary."""


class Unsupported(PcxError, NotImplementedError):
    """Valid in the reference, not implemented on the device path."""


# pcx_q_frac / pcx_q_to / pcx_q_from
Q_FRAC_HALF_Q, Q_FRAC_HALF_ELEM = 0, 1
Q_TRUNCATE, Q_NEAREST = 0, 1
Q_FLOOR, Q_TOWARD_ZERO, Q_ROUND = 0, 1, 2


class QFormat(C.Structure):
    """pcx_qformat: the floatToQ / fromQ reading of the integer element types (include/pcx.h)."""
    _fields_ = [("frac", C.c_int), ("float_to_q", C.c_int), ("from_q", C.c_int)]


def qformat_ptr(q):
    """None -> NULL (the process-wide reading); a QFormat or a (frac, float_to_q, from_q) triple -> pointer to a pcx_qformat"""
    if q is None:
        return None
    if not isinstance(q, QFormat):
        q = QFormat(*[int(v) for v in q])
    return C.byref(q)


_vp, _sz, _i, _d = C.c_void_p, C.c_size_t, C.c_int, C.c_double
_psz = C.POINTER(C.c_size_t)

# name -> (restype, argtypes): every symbol include/pcx.h declares
SIGNATURES = {
    "pcx_last_error": (C.c_char_p, []),
    "pcx_version": (C.c_char_p, []),
    "pcx_device_count": (_i, [C.POINTER(_i)]),
    "pcx_set_device": (_i, [_i]),
    "pcx_get_device": (_i, [C.POINTER(_i)]),
    "pcx_dev_alloc": (_i, [C.POINTER(_vp), _sz]),
    "pcx_dev_free": (_i, [_vp]),
    "pcx_memcpy_h2d": (_i, [_vp, _vp, _sz, _vp]),
    "pcx_memcpy_d2h": (_i, [_vp, _vp, _sz, _vp]),
    "pcx_memcpy_d2d": (_i, [_vp, _vp, _sz, _vp]),
    "pcx_stream_sync": (_i, [_vp]),
    "pcx_pointer_kind": (_i, [_vp, C.POINTER(_i)]),
    "pcx_memcpy_to_host": (_i, [_vp, _vp, _sz]),
    "pcx_trace": (_i, [_i]),
    "pcx_host_alloc": (_i, [C.POINTER(_vp), _sz]),
    "pcx_host_free": (_i, [_vp]),
    "pcx_pcie_probe": (_i, [_sz, _i, C.POINTER(_d), C.POINTER(_d), C.POINTER(_d)]),
    "pcx_host_register": (_i, [_vp, _sz]),
    "pcx_host_unregister": (_i, [_vp]),
    "pcx_host_register_mapping": (_i, [_vp, _sz, _sz, C.POINTER(_vp), _psz]),
    "pcx_host_mapping_alive": (_i, [_vp, C.POINTER(_i)]),
    "pcx_host_release_range": (_i, [_vp, _sz]),
    "pcx_fill_uniform_f32_dev": (_i, [_vp, _sz, C.c_uint64, C.c_uint64, _vp]),
    "pcx_clock_probe_dev": (_i, [_vp, C.c_uint, _vp]),
    "pcx_fir_create": (_i, [_i, _i, _i, C.POINTER(_vp)]),
    "pcx_fir_destroy": (_i, [_vp]),
    "pcx_fir_set_taps": (_i, [_vp, _vp, _sz]),
    "pcx_fir_set_decimation": (_i, [_vp, _sz]),
    "pcx_fir_set_interpolation": (_i, [_vp, _sz]),
    "pcx_fir_set_algo": (_i, [_vp, _i]),
    "pcx_fir_set_qformat": (_i, [_vp, _vp]),
    "pcx_set_qformat": (_i, [_vp]),
    "pcx_get_qformat": (_i, [_vp]),
    "pcx_fir_get_geometry": (_i, [_vp, _psz, _psz]),
    "pcx_fir_last_algo": (_i, [_vp]),
    "pcx_fir_set_slots": (_i, [_vp, C.c_uint]),
    "pcx_fir_process": (_i, [_vp, _vp, _sz, _vp, _sz, _psz, _psz]),
    "pcx_fir_process_dev": (_i, [_vp, _vp, _sz, _vp, _sz, _psz, _psz, _vp]),
    "pcx_fir_process_dev_gated": (_i, [_vp, _vp, _sz, _vp, _sz, _psz, _psz, _vp, C.c_uint, _vp, C.POINTER(C.c_int)]),
    "pcx_gate_signal_dev": (_i, [_vp, C.c_uint, _vp]),
    "pcx_fft_create": (_i, [_i, _sz, _i, C.POINTER(_vp)]),
    "pcx_fft_destroy": (_i, [_vp]),
    "pcx_fft_transform": (_i, [_vp, _vp, _vp, _sz]),
    "pcx_fft_transform_dev": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "pcx_freqdemod_create": (_i, [_i, C.POINTER(_vp)]),
    "pcx_freqdemod_destroy": (_i, [_vp]),
    "pcx_freqdemod_reset": (_i, [_vp]),
    "pcx_freqdemod_process": (_i, [_vp, _vp, _vp, _sz]),
    "pcx_freqdemod_process_dev": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "pcx_rotate": (_i, [_i, _d, _d, _vp, _vp, _sz]),
    "pcx_rotate_dev": (_i, [_i, _d, _d, _vp, _vp, _sz, _vp]),
    "pcx_scale": (_i, [_i, _i, _d, _vp, _vp, _sz]),
    "pcx_scale_dev": (_i, [_i, _i, _d, _vp, _vp, _sz, _vp]),
    "pcx_rotate_q": (_i, [_i, _d, _d, _vp, _vp, _vp, _sz]),
    "pcx_rotate_q_dev": (_i, [_i, _d, _d, _vp, _vp, _vp, _sz, _vp]),
    "pcx_scale_q": (_i, [_i, _i, _d, _vp, _vp, _vp, _sz]),
    "pcx_scale_q_dev": (_i, [_i, _i, _d, _vp, _vp, _vp, _sz, _vp]),
    "pcx_abs": (_i, [_i, _i, _vp, _vp, _sz]),
    "pcx_abs_dev": (_i, [_i, _i, _vp, _vp, _sz, _vp]),
    "pcx_conj": (_i, [_i, _vp, _vp, _sz]),
    "pcx_conj_dev": (_i, [_i, _vp, _vp, _sz, _vp]),
    "pcx_angle": (_i, [_i, _vp, _vp, _sz]),
    "pcx_angle_dev": (_i, [_i, _vp, _vp, _sz, _vp]),
    "pcx_arith": (_i, [_i, _i, _i, _vp, _vp, _vp, _sz]),
    "pcx_arith_dev": (_i, [_i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "pcx_split_complex": (_i, [_i, _vp, _vp, _vp, _sz]),
    "pcx_split_complex_dev": (_i, [_i, _vp, _vp, _vp, _sz, _vp]),
    "pcx_combine_complex": (_i, [_i, _vp, _vp, _vp, _sz]),
    "pcx_combine_complex_dev": (_i, [_i, _vp, _vp, _vp, _sz, _vp]),
    "pcx_fmchain_create": (_i, [C.POINTER(_vp)]),
    "pcx_fmchain_destroy": (_i, [_vp]),
    "pcx_fmchain_set_phase": (_i, [_vp, _d]),
    "pcx_fmchain_set_taps": (_i, [_vp, _vp, _sz, _i]),
    "pcx_fmchain_reset": (_i, [_vp]),
    "pcx_fmchain_set_algo": (_i, [_vp, _i]),
    "pcx_fmchain_last_algo": (_i, [_vp]),
    "pcx_fmchain_set_slots": (_i, [_vp, C.c_uint]),
    "pcx_fmchain_process": (_i, [_vp, _vp, _sz, _vp, _sz, _psz, _psz]),
    "pcx_fmchain_process_dev": (_i, [_vp, _vp, _sz, _vp, _sz, _psz, _psz, _vp]),
    "pcx_fmchain_process_dev_gated": (_i, [_vp, _vp, _sz, _vp, _sz, _psz, _psz, _vp, C.c_uint, _vp, C.POINTER(C.c_int)]),
    "pcx_shard_create": (_i, [_i, C.POINTER(C.c_int), _i, C.POINTER(_vp)]),
    "pcx_shard_destroy": (_i, [_vp]),
    "pcx_shard_set_taps": (_i, [_vp, C.POINTER(C.c_double), _sz, _i]),
    "pcx_shard_set_gated": (_i, [_vp, _i]),
    "pcx_shard_set_submit_threads": (_i, [_vp, _i]),
    "pcx_shard_set_algo": (_i, [_vp, _i]),
    "pcx_shard_set_chain": (_i, [_vp, _i, _d]),
    "pcx_shard_configure": (_i, [_vp, _sz]),
    "pcx_shard_info": (_i, [_vp, C.POINTER(C.c_int), _psz, _psz, C.POINTER(C.c_int)]),
    "pcx_shard_buffers": (_i, [_vp, _i, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(C.c_int)]),
    "pcx_shard_scatter": (_i, [_vp, _vp, _sz]),
    "pcx_shard_step": (_i, [_vp]),
    "pcx_shard_post_exchange": (_i, [_vp]),
    "pcx_shard_compute": (_i, [_vp]),
    "pcx_shard_gather": (_i, [_vp, _vp, _sz]),
    "pcx_shard_sync": (_i, [_vp]),
}

_lib = None
_hip_runtime = None


def _preload_hip_runtime():
    """One HIP runtime per process.

    libpcx_hip.so needs `libamdhip64.so.7`.  PyTorch-ROCm wheels bundle their own copy
    (same SONAME) and load it by path; a process that ends up with both the system copy and
    torch's copy has two runtimes fighting over the device (the second one reports "no
    ROCm-capable device").  Whoever loads first decides, so when torch is installed we map
    torch's copy first -- by SONAME match libpcx_hip.so then binds to it, and a later
    `import torch` reuses the same mapping.  Without torch the RUNPATH (/opt/rocm/lib) copy
    is used.  PCX_HIP_RUNTIME=<path> overrides.
    """
    global _hip_runtime
    if _hip_runtime is not None:
        return
    path = os.environ.get("PCX_HIP_RUNTIME")
    if not path:
        try:
            import importlib.util
            spec = importlib.util.find_spec("torch")
            if spec is not None and spec.submodule_search_locations:
                cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
                if os.path.exists(cand):
                    path = cand
        except (ImportError, ValueError):
            path = None
    if path:
        _hip_runtime = C.CDLL(path, mode=C.RTLD_GLOBAL)
    else:
        _hip_runtime = False


def load():
    """Load libpcx_hip.so (raises if it is not built -- no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    path = LIB_PATH
    # tools/ only: PCX_HIP_LIBRARY points the loader at the diagnostic build (make -C pothoscomms_amd/csrc diag),
    # the one library that reads A/B switches from the environment and holds the timing-only kernel variants
    override = os.environ.get("PCX_HIP_LIBRARY")
    if override:
        if not os.path.exists(override):
            raise ImportError("PCX_HIP_LIBRARY=%s does not exist" % override)
        path = override
    if not os.path.exists(path):
        raise ImportError(
            "%s is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C pothoscomms_amd/csrc).  pothoscomms_amd has no CPU fallback." % LIB_PATH)
    _preload_hip_runtime()
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    return load().pcx_last_error().decode("utf-8", "replace")


def check(status):
    if status == OK:
        return
    msg = last_error()
    if status == ERR_ARG:
        raise InvalidArgument(status, msg)
    if status == ERR_UNSUPPORTED:
        raise Unsupported(status, msg)
    raise PcxError(status, msg)
