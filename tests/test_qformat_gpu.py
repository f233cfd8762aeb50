"""GPU suite: the Q-format reading as a PARAMETER (include/pcx.h, pcx_qformat).

Pothos::Util::floatToQ / fromQ (call sites filter/FIRFilter.cpp:300,348, math/Rotate.cpp:21,74, math/Scale.cpp:21,73) live in a
PothosCore header that is not under /root/reference, and the reference's own tests leave twelve readings standing
(profiles/r02/qformat_enumeration.txt): fractional bits = half the Q word | half the element word; floatToQ truncating | to nearest;
fromQ floor | toward zero | to nearest.  Under EVERY one of them the device's integer Rotate, Scale and FIR -- time-domain window,
packed dot product, polyphase, and the double-precision overlap-save pipeline -- must be bit-identical to the oracle computing
under the same reading, and the reference tests' known answers (TestRotate.cpp:28-53, TestScale.cpp:28-52) must still come out.
"""
import numpy as np
import pytest

from tests.util import rand_stream

pytestmark = pytest.mark.gpu
NAMES = ["HALF_Q", "HALF_ELEM"], ["TRUNCATE", "NEAREST"], ["FLOOR", "TOWARD_ZERO", "ROUND"]


def _readings():
    from oracle import oracle as o
    return o.QFORMATS


def _id(q):
    return "%s-%s-%s" % (NAMES[0][q[0]], NAMES[1][q[1]], NAMES[2][q[2]])


@pytest.fixture
def reading(request, oracle):
    q = request.param
    oracle.set_qformat(*q)
    yield q
    oracle.set_qformat()


ALL = pytest.mark.parametrize("reading", _readings(), ids=_id, indirect=True)
INTS = ["int8", "int16", "int32", "int64"]


def _scalar(oracle, name):
    return {"int8": oracle.I8, "int16": oracle.I16, "int32": oracle.I32, "int64": oracle.I64}[name]


@ALL
@pytest.mark.parametrize("name", INTS)
def test_rotate_and_scale_under_every_reading(oracle, dev, reading, name):
    sc = _scalar(oracle, name)
    rng = np.random.default_rng(hash((name,) + tuple(reading)) % (1 << 31))
    x = rand_stream(rng, sc, 20000, True)
    x[:8] = np.iinfo(x.dtype).min
    x[8:16] = np.iinfo(x.dtype).max
    for phase in (0.0, 0.3, np.pi / 2, 2.747554270528532, -1.1):
        assert np.array_equal(dev.rotate(x, phase, qformat=reading), oracle.rotate(x, phase)), phase
    for factor in (-1.0, -0.5, 0.0, 0.5, 1.0, 0.3337, -1.77, 100.25):
        assert np.array_equal(dev.scale(x, factor, True, qformat=reading), oracle.scale(x, factor, True)), factor
        xr = np.ascontiguousarray(x[:, 0])
        assert np.array_equal(dev.scale(xr, factor, False, qformat=reading), oracle.scale(xr, factor, False)), factor


# hand-computed int16 cases that tell the readings apart (no oracle involved):
#   factor 0.5 on +-5:  floatToQ(0.5) = 2^(n-1) exactly, product / 2^n = +-2.5  ->  floor 2 / -3, toward zero 2 / -2, nearest (ties up) 3 / -2
#   factor 0.3 on 1000: n = 16: floatToQ = 19660 (19660.8 truncated) | 19661 (nearest) -> 19,660,000 >> 16 = 299 | 19,661,000 >> 16 = 300
#                       n =  8: floatToQ = 76 (76.8 truncated) | 77 -> 76,000 >> 8 = 296 | 77,000 >> 8 = 300
KAT = [((0, 0, 0), 0.5, [5, -5], [2, -3]), ((0, 0, 1), 0.5, [5, -5], [2, -2]), ((0, 0, 2), 0.5, [5, -5], [3, -2]),
       ((1, 0, 0), 0.5, [5, -5], [2, -3]), ((1, 0, 1), 0.5, [5, -5], [2, -2]), ((1, 0, 2), 0.5, [5, -5], [3, -2]),
       ((0, 0, 0), 0.3, [1000], [299]), ((0, 1, 0), 0.3, [1000], [300]), ((1, 0, 0), 0.3, [1000], [296]), ((1, 1, 0), 0.3, [1000], [300])]


@pytest.mark.parametrize("q,factor,xin,want", KAT, ids=lambda v: _id(v) if isinstance(v, tuple) else None)
def test_hand_computed_cases_tell_the_readings_apart(oracle, dev, q, factor, xin, want):
    x = np.array(xin, np.int16)
    assert dev.scale(x, factor, False, qformat=q).tolist() == want
    oracle.set_qformat(*q)
    try:
        assert oracle.scale(x, factor, False).tolist() == want
    finally:
        oracle.set_qformat()


@ALL
def test_process_wide_reading_is_what_the_plain_calls_use(oracle, dev, reading):
    rng = np.random.default_rng(5)
    x = rand_stream(rng, oracle.I16, 5000, True)
    dev.set_qformat(reading)
    try:
        assert dev.get_qformat() == tuple(reading)
        assert np.array_equal(dev.rotate(x, 0.7), oracle.rotate(x, 0.7))
        assert np.array_equal(dev.scale(x, -0.77, True), oracle.scale(x, -0.77, True))
        f = dev.FirFilter("complex_int16", "COMPLEX")        # a handle starts with the process-wide reading
        taps = (rng.normal(size=31) + 1j * rng.normal(size=31)) * 0.2
        f.set_taps(taps)
        ref = oracle.Fir(oracle.I16, True, True); ref.set_taps(taps); ref.activate()
        want, rc, rp, _ = ref.work(x, 5000)
        got, c, p = f.process(x, 5000)
        assert (c, p) == (rc, rp) and np.array_equal(got, want)
    finally:
        dev.set_qformat(None)
    assert dev.get_qformat() == (0, 0, 0)


@ALL
@pytest.mark.parametrize("name", INTS)
@pytest.mark.parametrize("kind", ["real-REAL", "complex-REAL", "complex-COMPLEX"])
def test_fir_time_domain_and_polyphase_under_every_reading(oracle, dev, reading, name, kind):
    """the sliding-window kernel (M = L = 1) and the polyphase kernel (L = 3, M = 2) of fir_generic.hip, every integer type"""
    sc = _scalar(oracle, name)
    is_complex, ctaps = kind != "real-REAL", kind == "complex-COMPLEX"
    rng = np.random.default_rng(hash((name, kind) + tuple(reading)) % (1 << 31))
    taps = (rng.normal(size=21) + (1j * rng.normal(size=21) if ctaps else 0)) / np.sqrt(21) * 0.9
    x = rand_stream(rng, sc, 6000, is_complex)
    for L, M in ((1, 1), (3, 2)):
        ref = oracle.Fir(sc, is_complex, ctaps)
        f = dev.FirFilter((sc, is_complex), "COMPLEX" if ctaps else "REAL")
        f.set_qformat(reading)
        for b in (ref, f):
            b.set_taps(taps); b.set_decimation(M); b.set_interpolation(L)
        ref.activate()
        want, rc, rp, _ = ref.work(x, 3 * 6000)
        f.set_algo(dev._lib.FIR_EXACT)
        got, c, p = f.process(x, 3 * 6000)
        assert (c, p) == (rc, rp)
        assert np.array_equal(got, want), (L, M)


@ALL
@pytest.mark.parametrize("name", ["int16", "int8"])
@pytest.mark.parametrize("ntaps", [37, 255])
def test_fir_complex_int_fast_paths_under_every_reading(oracle, dev, reading, name, ntaps):
    """complex_int16 / complex_int8 with complex taps: 37 taps -> the packed dot-product kernel (taps within 16 bits) or the 24-bit
    window; 255 taps -> the double-precision overlap-save pipeline, whose store applies the reading's shift and rounding"""
    sc = _scalar(oracle, name)
    rng = np.random.default_rng(hash((name, ntaps) + tuple(reading)) % (1 << 31))
    taps = (rng.normal(size=ntaps) + 1j * rng.normal(size=ntaps)) / np.sqrt(ntaps) * 0.6
    n = 3 * 4096 + 555
    x = rand_stream(rng, sc, n + ntaps - 1, True)
    ref = oracle.Fir(sc, True, True); ref.set_taps(taps); ref.activate()
    want, rc, rp, _ = ref.work(x, n)
    f = dev.FirFilter((sc, True), "COMPLEX")
    f.set_qformat(reading)
    f.set_taps(taps)
    got, c, p = f.process(x, n)
    assert (c, p) == (rc, rp)
    assert np.array_equal(got, want)
    if ntaps == 255:
        assert f.last_algo == dev._lib.FIR_OLS_FFT
    # changing the reading of a live handle re-quantises its taps
    other = (1 - reading[0], reading[1], reading[2])
    f.set_qformat(other)
    oracle.set_qformat(*other)
    ref2 = oracle.Fir(sc, True, True); ref2.set_taps(taps); ref2.activate()
    want2, _, _, _ = ref2.work(x, n)
    oracle.set_qformat(*reading)
    got2, _, _ = f.process(x, n)
    assert np.array_equal(got2, want2)


@ALL
@pytest.mark.parametrize("name", ["int16", "int8"])
def test_fir_real_and_resampling_int_on_the_double_pipeline_under_every_reading(oracle, dev, reading, name):
    """real int16 / int8 streams (two real blocks per transform), decimating (M = 4) and interpolating (L = 3) complex integer
    filters: the other stores of fir_ols_f64.hip"""
    sc = _scalar(oracle, name)
    rng = np.random.default_rng(hash((name,) + tuple(reading)) % (1 << 31))
    n = 2 * 4096 + 999
    # real stream, real taps
    taps = rng.normal(size=127) / np.sqrt(127) * 0.8
    x = rand_stream(rng, sc, n + 126, False)
    ref = oracle.Fir(sc, False, False); ref.set_taps(taps); ref.activate()
    want, rc, rp, _ = ref.work(x, n)
    f = dev.FirFilter((sc, False), "REAL"); f.set_qformat(reading); f.set_taps(taps)
    got, c, p = f.process(x, n)
    assert (c, p) == (rc, rp) and np.array_equal(got, want)
    assert f.last_algo == dev._lib.FIR_OLS_FFT
    # complex stream: decimation 4, interpolation 3
    ctaps = (rng.normal(size=200) + 1j * rng.normal(size=200)) / np.sqrt(200) * 0.7
    xc = rand_stream(rng, sc, n + 199, True)
    for L, M in ((1, 4), (3, 1)):
        ref = oracle.Fir(sc, True, True)
        f = dev.FirFilter((sc, True), "COMPLEX"); f.set_qformat(reading)
        for b in (ref, f):
            b.set_taps(ctaps * L); b.set_decimation(M); b.set_interpolation(L)
        ref.activate()
        want, rc, rp, _ = ref.work(xc, 3 * n)
        got, c, p = f.process(xc, 3 * n)
        assert (c, p) == (rc, rp)
        assert np.array_equal(got, want), (L, M)


@ALL
def test_reference_test_points_hold_under_every_reading(dev, reading):
    """math/TestRotate.cpp:28-32,50-53 and math/TestScale.cpp:28-31,49-52 (POTHOS_TEST_CLOSE(out, expected, 1)) on the device: what
    the enumeration says of the twelve -- the reference's tests cannot tell them apart"""
    import os
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden.npz"))

    def close(got, exp, name):
        got = np.asarray(got, np.float64)
        if name == "int8":
            exp = exp.astype(np.int64).astype(np.int8).astype(np.float64)
            d = np.abs(got - exp)
            return np.minimum(d, 256 - d)
        return np.abs(got - exp)
    for name in INTS:
        x = gold["rotate_in_" + name]
        for k, phase in enumerate([0.0, np.pi / 2, np.pi, 3 * np.pi / 2]):
            assert np.max(close(dev.rotate(x, phase, qformat=reading), gold["rotate_exp_%s_%d" % (name, k)], name)) <= 1.0, (name, phase)
        xs = gold["scale_in_" + name]
        for k, factor in enumerate([-1.0, -0.5, 0.0, 0.5, 1.0]):
            assert np.max(close(dev.scale(xs, factor, False, qformat=reading), gold["scale_exp_%s_%d" % (name, k)], name)) <= 1.0, (name, factor)


def test_blocks_take_the_reading_by_name(oracle):
    """/comms/fir_filter, /comms/rotate, /comms/scale: setQFormat("HALF_ELEM,NEAREST,ROUND") (an extension of the block interface)"""
    from pothoscomms_amd import blocks as B
    from pothoscomms_amd._lib import InvalidArgument
    rng = np.random.default_rng(9)
    x = rand_stream(rng, oracle.I16, 4000, True)
    q = (1, 1, 2)
    oracle.set_qformat(*q)
    try:
        rot = B.make("/comms/rotate", "complex_int16")
        rot.call("setPhase", 0.9)
        assert rot.call("getQFormat") == "DEFAULT"
        rot.call("setQFormat", "HALF_ELEM,NEAREST,ROUND")
        assert rot.call("getQFormat") == "HALF_ELEM,NEAREST,ROUND"
        y, c, p, _, _ = rot.work(x, 4000)
        assert np.array_equal(y[:p], oracle.rotate(x, 0.9)[:p])
        scl = B.make("/comms/scale", "complex_int16")
        scl.call("setFactor", -0.377)
        scl.call("setQFormat", "HALF_ELEM, NEAREST, ROUND")
        y, c, p, _, _ = scl.work(x, 4000)
        assert np.array_equal(y[:p], oracle.scale(x, -0.377, True)[:p])
        taps = (rng.normal(size=33) + 1j * rng.normal(size=33)) * 0.15
        fir = B.make("/comms/fir_filter", "complex_int16", "COMPLEX")
        fir.call("setTaps", taps)
        fir.call("setQFormat", "HALF_ELEM,NEAREST,ROUND")
        fir.activate()
        ref = oracle.Fir(oracle.I16, True, True); ref.set_taps(taps); ref.activate()
        want, rc, rp, _ = ref.work(x, 4000)
        y, c, p, _, _ = fir.work(x, 4000)
        assert (c, p) == (rc, rp) and np.array_equal(y[:p], want[:p])
        with pytest.raises((InvalidArgument, ValueError)):
            fir.call("setQFormat", "HALF_Q,SOMETIMES,FLOOR")
    finally:
        oracle.set_qformat()


@pytest.mark.parametrize("seed", range(24))
def test_integer_fir_randomised_geometry_and_reading(oracle, dev, seed):
    """random (integer type, taps kind, K, L, M, buffer sizes, output room) AND a random one of the twelve readings through AUTO --
    whatever kernel family the geometry lands on -- against the oracle's work() under the same reading, bit for bit"""
    rng = np.random.default_rng(7000 + seed)
    q = oracle.QFORMATS[int(rng.integers(0, 12))]
    scalar = [oracle.I16, oracle.I8, oracle.I32, oracle.I64][seed % 4]
    is_complex = bool(rng.integers(0, 3) > 0)
    ctaps = is_complex and bool(rng.integers(0, 2))
    L, M = int(rng.integers(1, 5)), int(rng.integers(1, 6))
    if seed % 3 == 0:
        L = M = 1
    ntaps = int(rng.integers(1, 400 if scalar in (oracle.I16, oracle.I8) else 80))
    K = -(-ntaps // L)
    n_in = int(rng.integers(K, K + 30000))
    out_cap = int(rng.integers(1, 2 * n_in * L // M + 10))
    taps = (rng.normal(size=ntaps) + (1j * rng.normal(size=ntaps) if ctaps else 0)) / np.sqrt(ntaps) * 0.9
    x = rand_stream(rng, scalar, n_in, is_complex, amp=1000 if scalar != oracle.I8 else 100)
    oracle.set_qformat(*q)
    try:
        ref = oracle.Fir(scalar, is_complex, ctaps)
        f = dev.FirFilter((scalar, is_complex), "COMPLEX" if ctaps else "REAL")
        f.set_qformat(q)
        for b in (ref, f):
            b.set_taps(taps); b.set_interpolation(L); b.set_decimation(M)
        ref.activate()
        want, rc, rp, _ = ref.work(x, out_cap)
    finally:
        oracle.set_qformat()
    got, gc, gp = f.process(x, out_cap)
    assert (gc, gp) == (rc, rp), (q, scalar, is_complex, ctaps, L, M, ntaps, n_in, out_cap)
    if rp:
        assert np.array_equal(got, want), (q, scalar, is_complex, ctaps, L, M, ntaps, f.last_algo)
