"""CPU suite: the host-side block layer (libpcx_blocks.so) -- registry paths, factory type
matrices, registered calls, buffer managers and every work() path that returns before any
device call.  Mirrors what the reference tests set up through BlockRegistry::make + call()."""
import numpy as np
import pytest

from pothoscomms_amd import blocks as B

ALL = ["float64", "float32", "int64", "int32", "int16", "int8"]


def test_registry_paths_match_the_reference():
    """FIRFilter.cpp:385-389, FFT.cpp:94-95, FreqDemod.cpp:94-95, Rotate.cpp:159-160, Scale.cpp:159-160,
    Abs.cpp:124-125, Conjugate.cpp:118-119"""
    assert B.registry_paths() == sorted(["/blocks/fir_filter", "/comms/abs", "/comms/angle", "/comms/conjugate", "/comms/fft",
                                         "/comms/fir_filter", "/comms/freq_demod", "/comms/rotate", "/comms/scale",
                                         # Arithmetic.cpp:300-304, SplitComplex.cpp:72-73, CombineComplex.cpp:71-72
                                         "/blocks/arithmetic", "/comms/arithmetic", "/comms/split_complex", "/comms/combine_complex",
                                         # FIRDesigner.cpp:479-483
                                         "/comms/fir_designer", "/blocks/fir_designer",
                                         # extension (no reference counterpart): rotate -> fir_filter -> freq_demod fused, configs[4]
                                         "/comms/fm_demod_chain"])
    with pytest.raises(ValueError):
        B.make("/comms/does_not_exist", "float32")


@pytest.mark.parametrize("t", ALL)
def test_fir_factory_matrix(t):
    """FIRFilterFactory: REAL on real and complex streams, COMPLEX on complex streams only"""
    for dtype, taps_type, ok in [(t, "REAL", True), ("complex_" + t, "REAL", True), ("complex_" + t, "COMPLEX", True),
                                 (t, "COMPLEX", False), ("complex_" + t, "BOTH", False)]:
        if ok:
            blk = B.make("/comms/fir_filter", dtype, taps_type)
            assert (blk.in_dtype, blk.out_dtype) == (dtype, dtype)
            assert blk.buffer_manager(False)[0] == "circular"          # FIRFilter.cpp:196-199
        else:
            with pytest.raises(ValueError, match="unsupported types"):
                B.make("/comms/fir_filter", dtype, taps_type)
    with pytest.raises(ValueError):
        B.make("/comms/fir_filter", "uint8", "REAL")
    assert B.make("/blocks/fir_filter", "complex_float32", "COMPLEX").in_dtype == "complex_float32"


def test_fir_registered_calls_round_trip():
    blk = B.make("/comms/fir_filter", "complex_float32", "COMPLEX")
    assert np.array_equal(blk.call("getTaps", True), [1.0 + 0j])      # ctor: setTaps({1})
    taps = np.array([1 + 2j, 3 - 4j, 0.5j])
    blk.call("setTaps", taps)
    assert np.array_equal(blk.call("getTaps", True), taps)
    blk.call("setDecimation", 3); blk.call("setInterpolation", 2)
    assert (blk.call("getDecimation"), blk.call("getInterpolation")) == (3, 2)
    blk.call("setWaitTaps", True)
    assert blk.call("getWaitTaps") is True
    blk.call("setFrameStartId", "START"); blk.call("setFrameEndId", "END")
    assert (blk.call("getFrameStartId"), blk.call("getFrameEndId")) == ("START", "END")
    with pytest.raises(ValueError, match="taps cannot be empty"):
        blk.call("setTaps", np.array([], dtype=np.complex128))
    with pytest.raises(ValueError, match="decimation cannot be 0"):
        blk.call("setDecimation", 0)
    with pytest.raises(ValueError, match="interpolation cannot be 0"):
        blk.call("setInterpolation", 0)
    with pytest.raises(B._lib.PcxError):
        blk.call("noSuchCall", 1)
    real = B.make("/comms/fir_filter", "float32", "REAL")
    real.call("setTaps", np.array([0.25, 0.5]))
    assert np.array_equal(real.call("getTaps"), [0.25, 0.5])


def test_fir_work_paths_without_device_work():
    """work() returns early: waiting for taps (:209), no input (:213), not enough input (:251-255),
    burst not complete (:243-247)."""
    blk = B.make("/comms/fir_filter", "complex_float32", "COMPLEX")
    blk.call("setTaps", np.ones(10, np.complex128)); blk.call("setDecimation", 4)
    blk.activate()
    x = np.zeros((12, 2), np.float32)
    _, c, p, r, _ = blk.work(x, 100)
    assert (c, p, r) == (0, 0, 13)                    # M + K - 1
    _, c, p, r, _ = blk.work(x[:0], 100)
    assert (c, p, r) == (0, 0, None)
    blk.call("setWaitTaps", True); blk.activate()
    _, c, p, r, _ = blk.work(np.zeros((100, 2), np.float32), 100)
    assert (c, p, r) == (0, 0, None)                  # armed until setTaps arrives
    blk = B.make("/comms/fir_filter", "float32", "REAL")
    blk.call("setTaps", np.ones(4)); blk.call("setFrameStartId", "S"); blk.activate()
    _, c, p, r, _ = blk.work(np.ones(20, np.float32), 100, [B.Label("S", 5, 100)])
    assert (c, p, r) == (0, 0, 105)                   # whole frame not here yet


@pytest.mark.parametrize("t,ok", [("float64", True), ("float32", True), ("int16", True), ("int32", False), ("int8", False), ("int64", False)])
def test_fft_factory_matrix(t, ok):
    """FFTFactory: complex<double>, complex<float>, complex<int16>"""
    if not ok:
        with pytest.raises(ValueError, match="unsupported type"):
            B.make("/comms/fft", "complex_" + t, 64, False)
        return
    with pytest.raises(ValueError):
        B.make("/comms/fft", t, 64, False)            # real stream


def test_fft_rejects_unimplemented_sizes_loudly():
    with pytest.raises((NotImplementedError, B._lib.PcxError)):
        B.make("/comms/fft", "complex_float32", 1 << 20, False)


@pytest.mark.parametrize("path,real_ok,out_real", [("/comms/freq_demod", False, True), ("/comms/rotate", False, False),
                                                   ("/comms/scale", True, False), ("/comms/abs", True, True),
                                                   ("/comms/conjugate", False, False), ("/comms/angle", False, True)])
@pytest.mark.parametrize("t", ALL)
def test_map_block_factories(path, real_ok, out_real, t):
    if path == "/comms/freq_demod":
        pytest.skip("construction allocates the carried-state slot on the device (GPU suite)")
    blk = B.make(path, "complex_" + t)
    assert blk.in_dtype == "complex_" + t
    assert blk.out_dtype == (t if out_real else "complex_" + t)
    if real_ok:
        assert B.make(path, t).out_dtype == t
    else:
        with pytest.raises(ValueError, match="unsupported type"):
            B.make(path, t)
    # vector dimension is carried on the ports (Rotate.cpp:126: N = elems * dimension)
    assert B.make(path, "complex_" + t, dimension=4).in_dim == 4


def test_rotate_scale_registered_calls():
    r = B.make("/comms/rotate", "complex_float32")
    assert r.call("getPhase") == 0.0
    r.call("setPhase", 1.25); r.call("setLabelId", "ph")
    assert (r.call("getPhase"), r.call("getLabelId")) == (1.25, "ph")
    s = B.make("/comms/scale", "float32")
    s.call("setFactor", -0.5); s.call("setLabelId", "gain")
    assert (s.call("getFactor"), s.call("getLabelId")) == (-0.5, "gain")
    # nothing to do: minElements == 0 returns before any device call
    _, c, p, _, _ = s.work(np.zeros(0, np.float32), 10)
    assert (c, p) == (0, 0)


# ---- /comms/arithmetic, /comms/split_complex, /comms/combine_complex (host logic) ------------
ARITH_TYPES = ALL + ["uint64", "uint32", "uint16", "uint8"]


@pytest.mark.parametrize("t", ARITH_TYPES)
def test_arithmetic_factory_matrix(t):
    """arithmeticFactory (Arithmetic.cpp:279-297): 10 scalar types x real/complex x ADD/SUB/MUL/DIV."""
    for dtype in (t, "complex_" + t):
        for op in ("ADD", "SUB", "MUL", "DIV"):
            blk = B.make("/comms/arithmetic", dtype, op)
            assert (blk.in_dtype, blk.out_dtype) == (dtype, dtype)
            assert [p[0] for p in blk.ports(0)] == ["0"] and [p[0] for p in blk.ports(1)] == ["0"]
        with pytest.raises(ValueError, match="unsupported args"):
            B.make("/comms/arithmetic", dtype, "POW")
    assert B.make("/blocks/arithmetic", "float32", "ADD").in_dtype == "float32"


def test_arithmetic_ports_and_preload():
    blk = B.make("/comms/arithmetic", "complex_float32", "ADD")
    with pytest.raises(B._lib.PcxError, match="require inputs >= 2"):
        blk.call("setNumInputs", 1)
    blk.call("setNumInputs", 4)
    assert [p[:2] for p in blk.ports(0)] == [(str(k), "complex_float32") for k in range(4)]
    blk.call("setNumInputs", 2)                      # never removes ports (Arithmetic.cpp:174-180)
    assert len(blk.ports(0)) == 4
    fb = B.make("/comms/arithmetic", "int16", "ADD")
    fb.call("setPreload", [0, 5])                    # feedback on port 1: five zero elements queued at activate()
    assert fb.call("preload") == [0, 5] and len(fb.ports(0)) == 2
    assert [p[4] for p in fb.ports(0)] == [0, 0]
    fb.activate()
    assert [p[4] for p in fb.ports(0)] == [0, 5]
    fb.activate()                                    # clear() + pushBuffer(): not cumulative
    assert [p[4] for p in fb.ports(0)] == [0, 5]
    fb3 = B.make("/comms/arithmetic", "int16", "ADD")
    fb3.call("setPreload", [1, 2, 3])                # setPreload grows the port list (:184-187)
    assert len(fb3.ports(0)) == 3
    # nothing to do: returns before touching the device, nothing consumed or produced
    outs, cons, prod = fb.work_ports([np.zeros(0, np.int16), np.zeros(8, np.int16)], 8)
    assert cons == [0, 0] and prod == [0]
    assert fb.call("getNumInlineBuffers") == 0


@pytest.mark.parametrize("t", ALL)
def test_split_combine_factories(t):
    """both factories take the REAL element type; ports are named (SplitComplex.cpp:42-46, CombineComplex.cpp:41-45)"""
    sp = B.make("/comms/split_complex", t)
    assert [p[:2] for p in sp.ports(0)] == [("0", "complex_" + t)]
    assert [p[:2] for p in sp.ports(1)] == [("re", t), ("im", t)]
    cb = B.make("/comms/combine_complex", t)
    assert [p[:2] for p in cb.ports(0)] == [("re", t), ("im", t)]
    assert [p[:2] for p in cb.ports(1)] == [("0", "complex_" + t)]
    for path in ("/comms/split_complex", "/comms/combine_complex"):
        with pytest.raises(ValueError, match="unsupported type"):
            B.make(path, "complex_" + t)
        with pytest.raises(ValueError):
            B.make(path, "uint8")
    # minAllElements = 0 -> no device call
    outs, cons, prod = sp.work_ports([np.zeros((4, 2), np.dtype(t))], [4, 0])
    assert cons == [0] and prod == [0, 0]



def test_the_runners_circular_buffer_is_one_object_mapped_twice():
    """pcxb_circular_create: the stand-in for the framework's "circular" manager (what FIRFilter.cpp:196-199 asks Pothos for) -- pageable
    shared memory mapped twice back to back, so that a window may start anywhere in the first mapping and run across the wrap; the two
    mappings are adjacent entries of /proc/self/maps on ONE object, which is what pcx_host_register_mapping looks for"""
    c = B.CircularBuffer(100000)
    assert c.size % 4096 == 0 and c.size >= 100000
    w = c.view(c.size - 16, 64)
    w[:] = np.arange(64, dtype=np.uint8)
    assert np.array_equal(c.view(0, 48), np.arange(16, 64, dtype=np.uint8))           # the part behind the wrap is the buffer's start
    assert np.array_equal(c.view(c.size - 16, 16), np.arange(16, dtype=np.uint8))
    rows = [ln.split() for ln in open("/proc/self/maps") if "pcx-circular" in ln]
    lo0, hi0 = (int(v, 16) for v in rows[0][0].split("-"))
    lo1, hi1 = (int(v, 16) for v in rows[1][0].split("-"))
    assert len(rows) == 2 and lo0 == c.base and hi0 == lo1 and hi1 - lo1 == hi0 - lo0 == c.size
    assert rows[0][1] == rows[1][1] == "rw-s" and rows[0][4] == rows[1][4]               # shared, the same inode
    c.close()
    assert not [ln for ln in open("/proc/self/maps") if "pcx-circular" in ln]
    with pytest.raises(Exception):
        B.CircularBuffer(0)


def test_host_register_mapping_looks_before_it_locks():
    """pcx_host_register_mapping without a GPU: the /proc/self/maps logic runs on any box -- a window in private memory is left alone
    (PCX_OK, nothing locked), a range beyond max_bytes likewise; only a shared double mapping reaches hipHostRegister, which on a box
    without a device fails with a status, not a crash (with a device it locks: tests/test_blocks_gpu.py)"""
    import ctypes as C

    import torch

    from pothoscomms_amd import _lib
    L = _lib.load()
    a = np.zeros(1 << 20, np.uint8)
    base, n = C.c_void_p(), C.c_size_t()
    assert L.pcx_host_register_mapping(C.c_void_p(a.ctypes.data), a.nbytes, 0, C.byref(base), C.byref(n)) == 0
    assert not base.value and n.value == 0
    circ = B.CircularBuffer(1 << 20)
    assert L.pcx_host_register_mapping(C.c_void_p(circ.base), 4096, 1 << 20, C.byref(base), C.byref(n)) == 0 and not base.value    # 2 MiB > max_bytes
    rc = L.pcx_host_register_mapping(C.c_void_p(circ.base + circ.size - 4096), 8192, 0, C.byref(base), C.byref(n))                  # a window across the wrap
    if torch.cuda.is_available():
        assert rc == 0 and base.value == circ.base and n.value == 2 * circ.size
        assert L.pcx_host_unregister(C.c_void_p(circ.base)) == 0
    else:
        assert rc != 0 and not base.value and _lib.last_error()
    assert L.pcx_host_unregister(C.c_void_p(circ.base + 4096)) != 0              # never the base of a range this library locked
    circ.close()
