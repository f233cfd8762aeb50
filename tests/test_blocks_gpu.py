"""GPU suite: the /comms blocks driven work() by work() like a Pothos topology would, against the
oracle's restatement of the reference work() functions on the same buffers and labels."""
import numpy as np
import pytest

from pothoscomms_amd import blocks as B
from tests.util import TOL, ang_err, nerr, rand_stream

pytestmark = pytest.mark.gpu
NAME = {0: "float64", 1: "float32", 2: "int64", 3: "int32", 4: "int16", 5: "int8"}


def _stream_through(blk, ref, x, chunk_sizes, out_cap, labels_at=()):
    """Feed x in pieces the way a circular buffer would: un-consumed input stays at the front."""
    pos, avail_end = 0, 0
    outs, routs = [], []
    sizes = list(chunk_sizes)
    while pos < len(x) and sizes:
        avail_end = min(len(x), avail_end + sizes.pop(0))
        win = x[pos:avail_end]
        labs = [B.Label(i, idx - pos, d) for (i, idx, d) in labels_at if pos <= idx < avail_end]
        rlabs = [(("S" if i == "S" else "E"), idx - pos, 1, d) for (i, idx, d) in labels_at if pos <= idx < avail_end]
        y, c, p, r, _ = blk.work(win, out_cap, labs)
        ry, rc, rp, rr = ref.work(win, out_cap, rlabs)
        assert (c, p, r) == (rc, rp, rr), (pos, avail_end)
        outs.append(y); routs.append(ry)
        pos += c
    return np.concatenate(outs), np.concatenate(routs)


@pytest.mark.parametrize("scalar", [1, 0, 4])
@pytest.mark.parametrize("L,M", [(1, 1), (3, 2)])
def test_fir_block_streams_like_the_reference(oracle, scalar, L, M):
    rng = np.random.default_rng(scalar * 7 + L)
    taps = (rng.normal(size=63) + 1j * rng.normal(size=63)) / 8
    x = rand_stream(rng, scalar, 20000, True, amp=1000)
    blk = B.make("/comms/fir_filter", "complex_" + NAME[scalar], "COMPLEX")
    ref = oracle.Fir(scalar, True, True)
    blk.call("setTaps", taps); ref.set_taps(taps)
    blk.call("setInterpolation", L); ref.set_interpolation(L)
    blk.call("setDecimation", M); ref.set_decimation(M)
    blk.activate(); ref.activate()
    got, want = _stream_through(blk, ref, x, [10, 50, 3, 5000, 1, 7000, 100000], 6000)
    assert got.shape == want.shape and got.shape[0] > 0
    if scalar == 4:
        assert np.array_equal(got, want)
    else:
        assert nerr(got, want) <= (TOL if scalar == 1 else 1e-12)


def test_fir_block_burst_mode(oracle):
    """frame start label -> exactly B outputs per burst, tail flushed with zeros (FIRFilter.cpp:218-272)"""
    rng = np.random.default_rng(4)
    taps = rng.normal(size=31) / 4
    B_len = 1000
    x = rand_stream(rng, oracle.F32, 3000, True)
    blk = B.make("/comms/fir_filter", "complex_float32", "REAL")
    ref = oracle.Fir(oracle.F32, True, False)
    for b in (blk, ref):
        (b.call("setTaps", taps) if b is blk else b.set_taps(taps))
    blk.call("setFrameStartId", "S"); ref.set_frame_ids(True, False)
    blk.activate(); ref.activate()
    got, want = _stream_through(blk, ref, x, [400, 400, 400, 400, 400, 400, 400, 400], 100000,
                                labels_at=[("S", 100, B_len)])
    assert nerr(got, want) <= TOL
    # 100 samples before the burst stream normally (K-1 stay behind), then the burst gives B_len outputs
    assert got.shape[0] >= B_len


def test_fir_block_propagates_labels(oracle):
    """propagateLabels: index/width scaled by L/M, rxRate rescaled (FIRFilter.cpp:311-323)"""
    blk = B.make("/comms/fir_filter", "complex_float32", "COMPLEX")
    blk.call("setInterpolation", 3); blk.call("setDecimation", 2); blk.activate()
    x = np.zeros((4096, 2), np.float32)
    labs = [B.Label("rxRate", 100, 1e6), B.Label("other", 200, 7, width=4), B.Label("late", 4000, "x")]
    _, c, p, _, posted = blk.work(x, 1 << 20, labs)
    assert c == 4096 and p == 6144
    assert posted == [B.Label("rxRate", 150, 1.5e6), B.Label("other", 300, 7, width=6), B.Label("late", 6000, "x")]


def test_fft_block(oracle):
    rng = np.random.default_rng(5)
    blk = B.make("/comms/fft", "complex_float32", 256, False)
    assert blk.initial_reserve() == 256                      # FFT.cpp:50
    name, size = blk.buffer_manager(True)
    assert name == "generic" and size % (256 * 8) == 0       # FFT.cpp:54-59 (several frames per slab)
    x = rand_stream(rng, oracle.F32, 256 * 5 + 100, True)
    y, c, p, _, _ = blk.work(x, 256 * 3)                     # room for 3 frames only
    assert (c, p) == (768, 768)
    assert nerr(y, oracle.fft(x[:768], 256)) <= TOL
    y, c, p, _, _ = blk.work(x[:200], 256 * 3)               # less than a frame: nothing happens
    assert (c, p) == (0, 0)
    inv = B.make("/comms/fft", "complex_int16", 64, True)
    xi = rand_stream(rng, oracle.I16, 128, True)
    y, c, p, _, _ = inv.work(xi, 128)
    assert (c, p) == (128, 128) and np.array_equal(y, oracle.fft(xi, 64, True))


@pytest.mark.parametrize("scalar", [1, 4])
def test_rotate_block_label_driven_phase(oracle, scalar):
    """Rotate.cpp:105-123: a label at the front sets the phase; one further in ends this call"""
    rng = np.random.default_rng(6)
    x = rand_stream(rng, scalar, 1000, True, amp=1000)
    blk = B.make("/comms/rotate", "complex_" + NAME[scalar])
    y, c, p, _, _ = blk.work(x, 1000)
    assert (c, p) == (1000, 1000) and not np.any(y)          # phasor never set -> zeros (Rotate.cpp:60-62)
    blk.call("setPhase", 0.3); blk.call("setLabelId", "ph")
    labs = [B.Label("ph", 400, 1.1), B.Label("noise", 10, 5.0), B.Label("ph", 700, -2.0)]
    y, c, p, _, _ = blk.work(x, 1000, labs)
    assert (c, p) == (400, 400) and np.array_equal(y, oracle.rotate(x[:400], 0.3))
    labs = [B.Label("ph", 0, 1.1), B.Label("ph", 300, -2.0)]
    y, c, p, _, _ = blk.work(x[400:], 1000, labs)
    assert (c, p) == (300, 300) and np.array_equal(y, oracle.rotate(x[400:700], 1.1))
    assert blk.call("getPhase") == 1.1
    y, c, p, _, _ = blk.work(x[700:], 1000, [B.Label("ph", 0, -2.0)])
    assert (c, p) == (300, 300) and np.array_equal(y, oracle.rotate(x[700:], -2.0))


def test_scale_block_label_and_dimension(oracle):
    rng = np.random.default_rng(7)
    x = rand_stream(rng, oracle.F32, 1200, False)
    blk = B.make("/comms/scale", "float32", dimension=4)     # 300 elements of dimension 4
    blk.call("setFactor", 2.0); blk.call("setLabelId", "g")
    y, c, p, _, _ = blk.work(x, 300, [B.Label("g", 100, 0.5)])
    assert (c, p) == (100, 100) and np.array_equal(y, oracle.scale(x[:400], 2.0, False))
    y, c, p, _, _ = blk.work(x[400:], 300, [B.Label("g", 0, 0.5)])
    assert (c, p) == (200, 200) and np.array_equal(y, oracle.scale(x[400:], 0.5, False))


@pytest.mark.parametrize("scalar", [1, 0, 3, 5])
def test_abs_conjugate_freqdemod_blocks(oracle, scalar):
    rng = np.random.default_rng(8 + scalar)
    x = rand_stream(rng, scalar, 5000, True, amp=100)
    y, c, p, _, _ = B.make("/comms/abs", "complex_" + NAME[scalar]).work(x, 4000)
    assert (c, p) == (4000, 4000)
    if scalar == 0:
        assert nerr(y, oracle.abs_(x[:4000], True)) < 1e-15 * 4
    else:
        assert np.array_equal(y, oracle.abs_(x[:4000], True))
    y, c, p, _, _ = B.make("/comms/conjugate", "complex_" + NAME[scalar]).work(x, 6000)
    assert (c, p) == (5000, 5000) and np.array_equal(y, oracle.conj(x))
    fd = B.make("/comms/freq_demod", "complex_" + NAME[scalar]); fd.activate()
    assert fd.out_dtype == NAME[scalar]
    ref = oracle.FreqDemod(scalar)
    for a, b in ((0, 1), (1, 2500), (2500, 5000)):
        y, c, p, _, _ = fd.work(x[a:b], 10000)
        want = ref.work(x[a:b])
        assert (c, p) == (b - a, b - a)
        if scalar in (0, 1):
            assert ang_err(y, want) <= TOL
        else:
            assert np.array_equal(y, want)
    fd.activate(); ref.activate()
    assert ang_err(fd.work(x[:10], 10)[0].astype(np.float64), ref.work(x[:10]).astype(np.float64)) <= TOL or scalar > 1


# ---- /comms/arithmetic, split_complex, combine_complex blocks ---------------------------------
@pytest.mark.parametrize("dtype", ["complex_float32", "int16", "complex_int8", "uint32", "float64"])
@pytest.mark.parametrize("op", ["ADD", "SUB", "MUL", "DIV"])
def test_arithmetic_block_folds_ports(oracle, dtype, op):
    """out = in0 OP in1 OP in2 (Arithmetic.cpp:217-224); every port consumes minElements."""
    from pothoscomms_amd.device import parse_dtype, NP_SCALAR
    scalar, cplx = parse_dtype(dtype)
    dt = np.dtype(NP_SCALAR[scalar])
    rng = np.random.default_rng(3)
    shape = lambda n: (n, 2) if cplx else (n,)
    def operand(n, divisor):
        if dt.kind == "f":
            return (rng.standard_normal(shape(n)) * 10 + (20 if divisor else 0)).astype(dt)
        lo, hi = (1, 9) if dt.kind == "u" or not divisor else (-9, 9)
        if divisor:
            v = rng.integers(lo, hi, size=shape(n), endpoint=True)
            return np.where(v == 0, 3, v).astype(dt)
        info = np.iinfo(dt)
        return (rng.integers(info.min, info.max, size=shape(n), dtype=dt, endpoint=True) // 4).astype(dt)
    blk = B.make("/comms/arithmetic", dtype, op)
    blk.call("setNumInputs", 3)
    blk.activate()
    ins = [operand(1000, False), operand(700, op == "DIV"), operand(900, op == "DIV")]
    outs, cons, prod = blk.work_ports(ins, 800)
    assert cons == [700, 700, 700] and prod == [700]
    o = getattr(oracle, op)
    ref = oracle.arith(o, oracle.arith(o, ins[0][:700], ins[1][:700], cplx), ins[2][:700], cplx)
    assert np.array_equal(outs[0].view(np.uint8), ref.view(np.uint8))
    assert blk.call("getNumInlineBuffers") == 0
    # the framework may forward input 0's buffer as the output buffer (setReadBeforeWrite): same result in place
    a0 = ins[0][:700].copy()
    outs, cons, prod = blk.work_ports([a0, ins[1], ins[2]], 700, inline=True)
    assert np.array_equal(outs[0].view(np.uint8), ref.view(np.uint8)) and blk.call("getNumInlineBuffers") == 1


def test_arithmetic_block_vector_dimension(oracle):
    blk = B.make("/comms/arithmetic", "float32", "MUL", dimension=4)
    rng = np.random.default_rng(5)
    a, b = rng.standard_normal(4 * 50).astype(np.float32), rng.standard_normal(4 * 60).astype(np.float32)
    blk.call("setNumInputs", 2)
    outs, cons, prod = blk.work_ports([a, b], 64)
    assert cons == [50, 50] and prod == [50]
    assert np.array_equal(outs[0], oracle.arith(oracle.MUL, a, b[:200], False))


@pytest.mark.parametrize("t", ["float32", "int16", "float64", "int8"])
def test_split_combine_blocks(oracle, t):
    """utility/TestComplex.cpp:13-60: combine -> split returns both planes."""
    dt = np.dtype(t)
    rng = np.random.default_rng(6)
    re, im = (rng.standard_normal(300) * 50).astype(dt), (rng.standard_normal(260) * 50).astype(dt)
    cb = B.make("/comms/combine_complex", t)
    outs, cons, prod = cb.work_ports([re, im], 280)
    assert cons == [260, 260] and prod == [260]
    z = outs[0]
    assert np.array_equal(z, oracle.combine_complex(re[:260], im[:260]))
    sp = B.make("/comms/split_complex", t)
    outs, cons, prod = sp.work_ports([z], [300, 250])
    assert cons == [250] and prod == [250, 250]
    assert np.array_equal(outs[0], re[:250]) and np.array_equal(outs[1], im[:250])


# ---- filter/TestFIRFilter.cpp:10-80, restated on the runner ---------------------------------------
@pytest.mark.parametrize("dtype", ["complex_float64", "complex_int16", "complex_float32"])
def test_fir_filter_tone_rms_like_the_reference(oracle, dtype):
    """A 30 kHz tone of amplitude 1000 at 1 MS/s through a 101-tap complex band-pass around it, for every
    decimation/interpolation in 1..3: the output RMS must stay above 0.1 * amplitude (TestFIRFilter.cpp:62-80).
    The reference gets its taps from /comms/fir_designer (spuce, absent here); the band edges are the
    test's own (waveFreq -+ 0.1 * sampRate at the filter's rate), the prototype a Hann-windowed sinc."""
    from pothoscomms_amd import taps as tp
    from pothoscomms_amd.device import parse_dtype, NP_SCALAR
    scalar, _ = parse_dtype(dtype)
    amplitude, rate, freq, total = 1000.0, 1e6, 30e3, 4096
    n = np.arange(total)
    wave = amplitude * np.exp(2j * np.pi * freq / rate * n)
    x = np.stack([wave.real, wave.imag], 1)
    x = (np.trunc(x) if scalar >= 2 else x).astype(NP_SCALAR[scalar])
    for decim in (1, 2, 3):
        for interp in (1, 2, 3):
            frate = rate * interp / decim                       # designer.setSampleRate((sampRate*interp)/decim)
            h = tp.complex_bandpass(101, 0.1 * rate / frate, freq / frate)
            if not (0.1 * rate / frate < 0.5 and abs(freq / frate) + 0.1 * rate / frate < 0.5):
                h = tp.complex_bandpass(101, 0.2, freq / frate)  # band edge beyond Nyquist at this rate: clamp
            blk = B.make("/comms/fir_filter", dtype, "COMPLEX")
            blk.call("setDecimation", decim); blk.call("setInterpolation", interp)
            blk.call("setWaitTaps", True)
            blk.activate()
            ref = oracle.Fir(scalar, True, True)
            ref.set_decimation(decim); ref.set_interpolation(interp); ref.set_wait_taps(True); ref.activate()
            cap = total * interp // decim + 8
            # armed by activate(): nothing moves until the designer's tapsChanged reaches setTaps (FIRFilter.cpp:201-216)
            assert blk.work(x, cap)[1:3] == (0, 0) and ref.work(x, cap)[1:3] == (0, 0)
            blk.call("setTaps", h * interp)                      # unity pass-band gain after zero-stuffing
            ref.set_taps(h * interp)
            y, c, p, _, _ = blk.work(x, cap)
            ry, rc, rp, _ = ref.work(x, cap)
            assert (c, p) == (rc, rp) and p > 0
            if scalar >= 2:
                assert np.array_equal(y, ry)
            else:
                assert nerr(y, ry) <= TOL
            z = y.astype(np.float64)
            rms = np.sqrt(np.mean(z[:, 0] ** 2 + z[:, 1] ** 2))
            assert rms > 0.1 * amplitude, (decim, interp, rms)


def test_arithmetic_feedback_like_the_reference():
    """math/TestArithmeticBlocks.cpp:300-338: adder with its output fed back to input 1 (preload 1 zero)
    accumulates a running sum.  The scheduler's part (queue per port, one element becomes available
    on the feedback port per work()) is played here."""
    adder = B.make("/comms/arithmetic", "int32", "ADD")
    adder.call("setPreload", [0, 1])
    adder.activate()
    pre = [p[4] for p in adder.ports(0)]
    assert pre == [0, 1]
    q0 = list(range(10))                       # feeder: 0..9
    q1 = [0] * pre[1]                          # feedback port: the preloaded zero
    collected = []
    while q0 and q1:
        n = min(len(q0), len(q1))
        outs, cons, prod = adder.work_ports([np.array(q0[:n], np.int32), np.array(q1[:n], np.int32)], n)
        assert cons == [n, n] and prod == [n]
        del q0[:n], q1[:n]
        y = outs[0].tolist()
        collected += y
        q1 += y                                # adder:0 -> adder:1
    last, want = 0, []
    for i in range(10):
        last = i + last
        want.append(last)
    assert collected == want


def test_arithmetic_inline_buffer_like_the_reference():
    """math/TestArithmeticBlocks.cpp:341-391: 4000 elements i and i+4000 added; the framework forwards
    input 0's buffer as the output buffer and the block counts it."""
    n = 4000
    a = np.arange(n, dtype=np.int32)
    b = (np.arange(n) + n).astype(np.int32)
    adder = B.make("/comms/arithmetic", "int32", "ADD")
    adder.call("setNumInputs", 2)
    outs, cons, prod = adder.work_ports([a.copy(), b], n, inline=True)
    assert prod == [n] and np.array_equal(outs[0], a + a + n)
    assert adder.call("getNumInlineBuffers") > 0


def test_fir_block_kernel_selection_and_nonfinite_locality(oracle):
    """setKernel (an extension): EXACT keeps the reference's operation order -- bit-identical floats, and an
    Inf sample touches K outputs only -- while the default frequency-domain kernel poisons its block"""
    rng = np.random.default_rng(8)
    K, n = 31, 12000
    taps = (rng.normal(size=K) + 1j * rng.normal(size=K)) / 6
    x = rand_stream(rng, oracle.F32, n, True)
    x[5000, 0] = np.inf
    ref = oracle.Fir(oracle.F32, True, True); ref.set_taps(taps); ref.activate()
    want, rc, rp, _ = ref.work(x, n)
    blk = B.make("/comms/fir_filter", "complex_float32", "COMPLEX")
    blk.call("setTaps", taps)
    assert blk.call("getKernel") == "AUTO"
    blk.call("setKernel", "EXACT"); blk.activate()
    y, c, p, _, _ = blk.work(x, n)
    assert (c, p) == (rc, rp)
    bad_ref = ~np.isfinite(want).all(axis=1)
    assert bad_ref.sum() == K and np.array_equal(~np.isfinite(y).all(axis=1), bad_ref)
    assert np.array_equal(y[~bad_ref], want[~bad_ref])
    blk.call("setKernel", "AUTO")
    y2, _, _, _, _ = blk.work(x, n)
    bad2 = ~np.isfinite(y2).all(axis=1)
    assert bad2[bad_ref].all() and bad2.sum() > K            # the whole 4096-sample block, as DESIGN.md states
    assert nerr(y2[~bad2], want[~bad2]) <= TOL
    with pytest.raises(ValueError, match="unknown kernel"):
        blk.call("setKernel", "FASTEST")


# ---- filter/TestFIRDesigner.cpp:137-230, restated on the runner ------------------------------------
def _run_until_dry(blk, x, out_cap, labels_at=()):
    """what the scheduler does with a finite source: work() until nothing more is consumed or produced"""
    pos, outs = 0, []
    for _ in range(64):
        labs = [B.Label(i, idx - pos, d) for (i, idx, d) in labels_at if pos <= idx < len(x)]
        y, c, p, _, _ = blk.work(x[pos:], out_cap, labs)
        outs.append(y)
        pos += c
        if c == 0 and p == 0:
            break
    return np.concatenate(outs)


@pytest.mark.parametrize("band", ["LOW_PASS", "HIGH_PASS", "BAND_PASS", "BAND_STOP", "COMPLEX_BAND_PASS", "COMPLEX_BAND_STOP"])
def test_fir_designer_topology_like_the_reference(oracle, band):
    """vector_source(impulse, START label) -> fir_filter(complex_float64, COMPLEX, waitTaps, frameStartId) -> fft(1024)
    -> collector, with designer.tapsChanged -> filter.setTaps: the power spectrum must sit above -30 dB in the
    middle of every pass region and below -80 dB in the middle of every stop region (TestFIRDesigner.cpp:110-135,
    185-230; filter type SINC).  The filter runs the double-precision overlap-save kernel, the FFT the double
    radix-16 plan; the same taps through the reference's work() restatement give the comparison stream."""
    rate, lo, hi, nfft, ntaps = 1e6, 1.5e5, 3.0e5, 1024, 101
    dtype = "complex_float64"
    flt = B.make("/comms/fir_filter", dtype, "COMPLEX")
    flt.call("setDecimation", 1); flt.call("setInterpolation", 1)
    flt.call("setWaitTaps", True); flt.call("setFrameStartId", "START")
    des = B.make("/comms/fir_designer")
    des.connect_signal("tapsChanged", flt, "setTaps")
    des.call("setSampleRate", rate); des.call("setFilterType", "SINC"); des.call("setBandType", band)
    des.call("setFrequencyLower", lo); des.call("setFrequencyUpper", hi); des.call("setBandwidthTrans", rate / 20)
    des.call("setNumTaps", ntaps)
    fft = B.make("/comms/fft", dtype, nfft, False)
    impulse = np.zeros((nfft, 2)); impulse[nfft - 1, 0] = float(nfft)
    # commit(): the filter activates armed (no taps yet), then the designer's activation delivers them
    flt.activate(); fft.activate()
    assert flt.work(impulse, nfft, [B.Label("START", 0, nfft)])[1:3] == (0, 0)
    des.activate()
    taps = flt.call("getTaps", True)
    assert len(taps) == ntaps
    y = _run_until_dry(flt, impulse, 4 * nfft, labels_at=[("START", 0, nfft)])
    assert y.shape[0] == nfft                                   # the frame comes out whole: tail flushed with zeros
    ref = oracle.Fir(oracle.F64, True, True)
    ref.set_taps(taps); ref.set_frame_ids(True, False); ref.activate()
    pos, routs = 0, []
    for _ in range(8):
        ry, rc, rp, _ = ref.work(impulse[pos:], 4 * nfft, [("S", 0, 1, nfft)] if pos == 0 else [])
        routs.append(ry); pos += rc
        if rc == 0 and rp == 0:
            break
    assert nerr(y, np.concatenate(routs)) <= 1e-13
    bins, c, p, _, _ = fft.work(y, nfft)
    assert (c, p) == (nfft, nfft)
    z = bins[:, 0] + 1j * bins[:, 1]
    power = np.fft.fftshift(10 * np.log10(np.abs(z) ** 2 + 1e-300) - 20 * np.log10(nfft))

    def level(freq):
        return power[int(nfft * ((freq + rate / 2) / rate))]

    PASS, STOP = True, False
    points = {
        "LOW_PASS": [(STOP, -(lo + rate / 2) / 2), (PASS, 0.0), (STOP, (lo + rate / 2) / 2)],
        "HIGH_PASS": [(PASS, -(lo + rate / 2) / 2), (STOP, 0.0), (PASS, (lo + rate / 2) / 2)],
        "BAND_PASS": [(STOP, -(hi + rate / 2) / 2), (PASS, -(lo + hi) / 2), (STOP, 0.0), (PASS, (lo + hi) / 2), (STOP, (hi + rate / 2) / 2)],
        "BAND_STOP": [(PASS, -(hi + rate / 2) / 2), (STOP, -(lo + hi) / 2), (PASS, 0.0), (STOP, (lo + hi) / 2), (PASS, (hi + rate / 2) / 2)],
        "COMPLEX_BAND_PASS": [(STOP, (lo - rate / 2) / 2), (PASS, (lo + hi) / 2), (STOP, (hi + rate / 2) / 2)],
        "COMPLEX_BAND_STOP": [(PASS, (lo - rate / 2) / 2), (STOP, (lo + hi) / 2), (PASS, (hi + rate / 2) / 2)],
    }[band]
    for is_pass, f in points:
        assert (level(f) > -30.0) if is_pass else (level(f) < -80.0), (band, f, level(f))


@pytest.mark.parametrize("dtype", ["complex_float64", "complex_int16"])
def test_fir_filter_tone_rms_with_the_designer(dtype):
    """filter/TestFIRFilter.cpp:10-80 as written there: the taps come from /comms/fir_designer (SINC,
    COMPLEX_BAND_PASS around the tone, 101 taps, designed at the filter's rate) over tapsChanged -> setTaps."""
    from pothoscomms_amd.device import parse_dtype, NP_SCALAR
    scalar, _ = parse_dtype(dtype)
    amplitude, rate, freq, total = 1000.0, 1e6, 30e3, 4096
    n = np.arange(total)
    wave = amplitude * np.exp(2j * np.pi * freq / rate * n)
    x = np.stack([wave.real, wave.imag], 1)
    x = (np.trunc(x) if scalar >= 2 else x).astype(NP_SCALAR[scalar])
    for decim in (1, 2, 3):
        for interp in (1, 2, 3):
            flt = B.make("/comms/fir_filter", dtype, "COMPLEX")
            flt.call("setDecimation", decim); flt.call("setInterpolation", interp); flt.call("setWaitTaps", True)
            des = B.make("/comms/fir_designer")
            des.connect_signal("tapsChanged", flt, "setTaps")
            des.call("setSampleRate", (rate * interp) / decim); des.call("setFilterType", "SINC")
            des.call("setBandType", "COMPLEX_BAND_PASS")
            des.call("setFrequencyLower", freq - 0.1 * rate); des.call("setFrequencyUpper", freq + 0.1 * rate)
            des.call("setBandwidthTrans", freq + 0.1 * rate); des.call("setNumTaps", 101)
            flt.activate()
            cap = total * interp // decim + 8
            assert flt.work(x, cap)[1:3] == (0, 0)
            des.activate()
            y, c, p, _, _ = flt.work(x, cap)
            assert p > 0
            z = y.astype(np.float64)
            rms = np.sqrt(np.mean(z[:, 0] ** 2 + z[:, 1] ** 2))
            assert rms > 0.1 * amplitude, (decim, interp, rms)


def test_fm_demod_chain_block_equals_the_three_reference_blocks(oracle):
    """/comms/fm_demod_chain (extension) fed in work()-sized pieces == the oracle's Rotate -> FIRFilter -> FreqDemod on the whole
    stream: same consume/produce as a FIR with M = L = 1, K-1 samples of history left on the port, state carried across calls,
    all-zero output until setPhase (Rotate.cpp:60-62), reset on activate (FreqDemod.cpp:44-47)"""
    from pothoscomms_amd import blocks as B, taps as tp
    from tests.util import ang_err
    rng = np.random.default_rng(21)
    n = 200000
    ph = np.cumsum(2 * np.pi * (0.02 + 0.01 * np.sin(2 * np.pi * np.arange(n) / 1000)))
    x = (np.stack([np.cos(ph), np.sin(ph)], 1) + rng.uniform(-1e-3, 1e-3, (n, 2))).astype(np.float32)
    h = tp.c4_taps()
    K = len(h)
    blk = B.make("/comms/fm_demod_chain", "complex_float32", "REAL")
    blk.call("setTaps", h)
    blk.activate()
    out0, c0, p0, _, _ = blk.work(x[:5000], 5000)
    assert (c0, p0) == (5000 - (K - 1), 5000 - (K - 1)) and not out0.any()       # no phase yet: zero phasor, zero output
    blk.call("setPhase", tp.C4_PHASE)
    blk.activate()
    fir = oracle.Fir(oracle.F32, True, False); fir.set_taps(h); fir.activate()
    y, _, p, _ = fir.work(oracle.rotate(x, tp.C4_PHASE), n - (K - 1))
    ref = oracle.FreqDemod(oracle.F32).work(y)
    got, pos = [], 0
    while True:
        take = int(rng.integers(1, 40000))
        buf = x[pos:pos + take + K - 1]
        o, c, pr, reserve, _ = blk.work(buf, take)
        if buf.shape[0] < K:
            assert (c, pr) == (0, 0) and reserve == K
            break
        assert c == pr == min(take, buf.shape[0] - (K - 1))
        got.append(o)
        pos += c
    got = np.concatenate(got)
    assert got.shape[0] == n - (K - 1)
    assert ang_err(got, ref) <= TOL


def test_fm_demod_chain_block_contract():
    """the fused-chain extension block: factory matrix, registered calls, buffer managers (its handle allocates the carried
    state on the device at construction, like /comms/freq_demod's)"""
    b = B.make("/comms/fm_demod_chain", "complex_float32", "REAL")
    assert (b.in_dtype, b.out_dtype) == ("complex_float32", "float32")
    b.call("setPhase", 0.25)
    assert b.call("getPhase") == 0.25
    b.call("setTaps", np.array([0.5, 0.25, 0.125]))
    assert list(b.call("getTaps")) == [0.5, 0.25, 0.125]
    assert b.buffer_manager(0)[0] == "circular" and b.buffer_manager(1)[0] == "generic"
    c = B.make("/comms/fm_demod_chain", "complex_float32", "COMPLEX")
    c.call("setTaps", np.array([1 + 1j, 2 - 1j]))
    assert list(c.call("getTaps", True)) == [1 + 1j, 2 - 1j]
    for bad in (("complex_float64", "REAL"), ("float32", "REAL"), ("complex_float32", "BOTH"), ("complex_int16", "COMPLEX")):
        with pytest.raises(Exception):
            B.make("/comms/fm_demod_chain", *bad)
    with pytest.raises(Exception):
        b.call("setTaps", np.array([]))


def _memcpy(dst, src, nbytes, h2d):
    import ctypes as C
    from pothoscomms_amd import _lib
    L = _lib.load()
    fn = L.pcx_memcpy_h2d if h2d else L.pcx_memcpy_d2h
    _lib.check(fn(C.c_void_p(dst), C.c_void_p(src), nbytes, None))
    _lib.check(L.pcx_stream_sync(None))


def test_three_separate_blocks_with_device_resident_edges(oracle):
    """Rotate -> FIRFilter -> FreqDemod as THREE blocks, wired as a scheduler would wire them (pcxb_link_buffer): both inner
    edges carry this module's port domain on either side, so their buffers are DEVICE slabs -- the FIR's circular input buffer
    and the FreqDemod's input live in HBM, only the first input and the last output are (page-locked) host memory -- and the
    host-pointer entry points run in place on all of them.  Fed in work()-sized pieces; equals the oracle chain."""
    from pothoscomms_amd import taps as tp
    rng = np.random.default_rng(33)
    n = 300000
    ph = np.cumsum(2 * np.pi * (0.02 + 0.01 * np.sin(2 * np.pi * np.arange(n) / 1000)))
    x = (np.stack([np.cos(ph), np.sin(ph)], 1) + rng.uniform(-1e-3, 1e-3, (n, 2))).astype(np.float32)
    h, phase = tp.c4_taps(), tp.C4_PHASE
    K = len(h)
    rot = B.make("/comms/rotate", "complex_float32"); rot.call("setPhase", phase)
    fir = B.make("/comms/fir_filter", "complex_float32", "REAL"); fir.call("setTaps", h)
    dem = B.make("/comms/freq_demod", "complex_float32")
    for b in (rot, fir, dem):
        b.activate()
    CH = 65536
    xin, pin0 = rot.port_buffer(0, (CH, 2), np.float32)
    e1, kind1 = rot.link_buffer(fir, (CH + K - 1) * 8)
    e2, kind2 = fir.link_buffer(dem, CH * 8)
    yout, pin3 = dem.port_buffer(1, (CH,), np.float32)
    assert pin0 and pin3 and (kind1, kind2) == (2, 2)
    assert fir.buffer_manager(0)[0] == "circular"
    got = []
    hist = 0                                    # samples of FIR history sitting at the front of the device edge buffer
    pos = 0
    while pos < n:
        m = min(CH, n - pos)
        xin[:m] = x[pos:pos + m]
        c, p, _ = rot.work_raw(xin.ctypes.data, m, e1 + hist * 8, m)            # rotate writes behind the history
        assert (c, p) == (m, m)
        avail = hist + m
        c, p, r = fir.work_raw(e1, avail, e2, CH)
        if avail < K:
            assert (c, p) == (0, 0)
            hist = avail
        else:
            assert c == p == avail - (K - 1)
            # the circular buffer keeps the unconsumed K-1 samples in front of the next ones: move them (device to device)
            import ctypes as C
            from pothoscomms_amd import _lib
            L = _lib.load()
            tmp = C.c_void_p()
            _lib.check(L.pcx_dev_alloc(C.byref(tmp), (K - 1) * 8))
            _lib.check(L.pcx_memcpy_d2d(tmp, C.c_void_p(e1 + c * 8), (K - 1) * 8, None))
            _lib.check(L.pcx_memcpy_d2d(C.c_void_p(e1), tmp, (K - 1) * 8, None))
            _lib.check(L.pcx_stream_sync(None))
            _lib.check(L.pcx_dev_free(tmp))
            hist = K - 1
            c2, p2, _ = dem.work_raw(e2, p, yout.ctypes.data, CH)
            assert (c2, p2) == (p, p)
            got.append(yout[:p].copy())
        pos += m
    got = np.concatenate(got)
    ref_fir = oracle.Fir(oracle.F32, True, False); ref_fir.set_taps(h); ref_fir.activate()
    y, _, p, _ = ref_fir.work(oracle.rotate(x, phase), n - (K - 1))
    ref = oracle.FreqDemod(oracle.F32).work(y)
    assert got.shape[0] == n - (K - 1)
    assert ang_err(got, ref) <= TOL


def test_edge_to_a_host_block_gets_host_memory():
    """a device block feeding a block of another domain (the designer has no ports; use a fresh generic request): pinned host"""
    rot = B.make("/comms/rotate", "complex_float32")
    arr, pinned = rot.port_buffer(1, (1024, 2), np.float32)
    assert pinned and isinstance(arr, np.ndarray)
    arr[:] = 1.0          # host-addressable


def test_burst_flush_on_a_device_resident_edge(oracle):
    """Rotate -> FIRFilter with setFrameStartId on an edge whose buffer is a DEVICE slab.  A burst shorter than
    M+K-1 takes FIRFilter::work's flush path (FIRFilter.cpp:263-272), which builds its zero-padded tail on the host: it must fetch
    the samples with a device copy, not with the CPU.  Equals the oracle block fed the same rotated samples and label."""
    import ctypes as C

    from pothoscomms_amd import _lib
    from pothoscomms_amd.blocks import Label
    L = _lib.load()
    rng = np.random.default_rng(44)
    taps = rng.normal(size=31) / 4
    K, burst = 31, 20                      # burst < K: the whole frame is shorter than the window
    x = rand_stream(rng, oracle.F32, burst, True)
    rot = B.make("/comms/rotate", "complex_float32"); rot.call("setPhase", 0.3)
    fir = B.make("/comms/fir_filter", "complex_float32", "REAL"); fir.call("setTaps", taps); fir.call("setFrameStartId", "S")
    for b in (rot, fir):
        b.activate()
    xin, _ = rot.port_buffer(0, (burst, 2), np.float32)
    e1, kind = rot.link_buffer(fir, 4096 * 8)
    assert kind == 2                       # device memory
    k = C.c_int(-1)
    _lib.check(L.pcx_pointer_kind(C.c_void_p(e1), C.byref(k)))
    assert k.value == 2
    yout, _ = fir.port_buffer(1, (4096, 2), np.float32)
    xin[:] = x
    assert rot.work_raw(xin.ctypes.data, burst, e1, burst)[:2] == (burst, burst)
    c, p, r = fir.work_raw(e1, burst, yout.ctypes.data, 4096, labels=[Label("S", 0, burst)])
    ref = oracle.Fir(oracle.F32, True, False)
    ref.set_taps(taps); ref.set_frame_ids(True, False); ref.activate()
    want, rc, rp, _ = ref.work(oracle.rotate(x, 0.3), 4096, [("S", 0, 1, burst)])      # (id, index, width, length)
    assert (c, p) == (rc, rp) and p == burst
    assert nerr(yout[:p], want) <= TOL


def test_fir_block_that_owns_a_sharded_stream(oracle):
    """/comms/fir_filter.setDevices([0, 0]) (an extension): every work() call spreads what the port holds over the listed devices --
    here two shards on device 0 over peer copies, the rehearsal of two GPUs -- through pcx_shard_* and equals the reference block's
    stream; consume / produce totals are the reference's, tails shorter than a shard set run on the single-device handle."""
    from pothoscomms_amd import taps as tp
    h = tp.c1_taps()
    K = len(h)
    rng = np.random.default_rng(77)
    n = 700000
    x = rand_stream(rng, oracle.F32, n, True)
    blk = B.make("/comms/fir_filter", "complex_float32", "COMPLEX")
    blk.call("setTaps", h)
    blk.call("setDevices", [0, 0])
    assert blk.call("getDevices") == [0, 0]
    ref = oracle.Fir(oracle.F32, True, True); ref.set_taps(h)
    blk.activate(); ref.activate()
    CH = 200000                       # what a port slab holds per call; the last call brings less
    got, want, pos = [], [], 0
    while pos + K - 1 < n:
        avail = min(CH + K - 1, n - pos)
        win = x[pos:pos + avail]
        y, c, p, r, _ = blk.work(win, CH)
        ry, rc, rp, _ = ref.work(win, CH)
        assert c == p and c > 0
        got.append(y); want.append(ry[:p] if p <= rp else ry)
        # the sharded call may consume a few samples less than the reference (2 shards of floor(N/2)); what matters is the stream
        pos += c
        if rp != p:                   # re-align the reference on what the block consumed
            ref = oracle.Fir(oracle.F32, True, True); ref.set_taps(h); ref.activate()
    got = np.concatenate(got)
    ref2 = oracle.Fir(oracle.F32, True, True); ref2.set_taps(h); ref2.activate()
    full, _, pf, _ = ref2.work(x, n)
    assert got.shape[0] == pf == n - (K - 1)
    assert nerr(got, full) <= TOL
    assert blk.call("getShardPasses") >= 3
    blk.call("setDevices", [])        # back to one device
    y, c, p, r, _ = blk.work(x[:50000], 50000)
    assert (c, p) == (50000 - (K - 1),) * 2 and nerr(y, full[:p]) <= TOL


def test_fir_block_page_locks_the_frameworks_circular_buffer_where_it_lies(oracle):
    """filter/FIRFilter.cpp:196-199: the reference FIR asks the framework for its "circular" input manager -- pageable memory mapped
    twice back to back.  The block page-locks that mapping the first time it sees it (both halves: pcx_host_register_mapping), so the
    kernel reads the window in place over PCIe instead of a CPU copy through the bounce buffer; results are the reference's, window
    after window, also when a window runs ACROSS the wrap, and the destructor unlocks."""
    import ctypes as C

    from pothoscomms_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(11)
    K = 255
    taps = (rng.normal(size=K) + 1j * rng.normal(size=K)) / 16
    blk = B.make("/comms/fir_filter", "complex_float32", "COMPLEX")
    ref = oracle.Fir(1, True, True)
    blk.call("setTaps", taps); ref.set_taps(taps)
    blk.activate(); ref.activate()
    circ = B.CircularBuffer(1 << 20)                      # 128 Ki complex_float32 samples
    cap = circ.size // 8
    kind = C.c_int()
    _lib.check(L.pcx_pointer_kind(C.c_void_p(circ.base), C.byref(kind)))
    assert kind.value == 0                                 # pageable: the framework's memory, not the module's
    x = rand_stream(rng, 1, 600000, True)
    rd = wr = 0                                            # sample counters of the stream; positions in the ring are modulo cap
    outs, routs = [], []
    for step, n_new in enumerate([40000, 90000, 70000, 100000, 100000, 100000, 100000]):
        n_new = min(n_new, cap - (wr - rd))                # the producer fills what is free
        ring = circ.view((wr % cap) * 8, n_new * 8, np.float32).reshape(-1, 2)
        ring[:] = x[wr:wr + n_new]                         # (contiguous through the second mapping when it wraps)
        wr += n_new
        avail = wr - rd
        win = circ.view((rd % cap) * 8, avail * 8, np.float32).reshape(-1, 2)
        assert np.array_equal(win, x[rd:wr])
        crosses = (rd % cap) + avail > cap
        y, c, p, r, _ = blk.work(win, 200000)
        ry, rc, rp, rr = ref.work(x[rd:wr], 200000)
        assert (c, p, r) == (rc, rp, rr) and c == avail - (K - 1)
        assert nerr(y, ry) <= TOL, (step, crosses)
        outs.append(crosses)
        rd += c
        _lib.check(L.pcx_pointer_kind(C.c_void_p(circ.base + 4096), C.byref(kind)))
        assert kind.value == 1                             # page-locked now, first half ...
        _lib.check(L.pcx_pointer_kind(C.c_void_p(circ.base + circ.size + 4096), C.byref(kind)))
        assert kind.value == 1                             # ... and the alias
    assert any(outs) and not all(outs)                     # windows on both sides of the wrap and across it
    blk.close()
    _lib.check(L.pcx_pointer_kind(C.c_void_p(circ.base), C.byref(kind)))
    assert kind.value == 0                                 # unlocked by the block's destructor
    circ.close()


def test_host_register_mapping_leaves_private_memory_alone():
    """pcx_host_register_mapping page-locks shared file mappings only: a window inside the heap (or any private mapping) is left as it
    is -- locking the arena would pin whatever else lives there -- and reported as 'nothing locked'."""
    import ctypes as C

    from pothoscomms_amd import _lib
    L = _lib.load()
    a = np.zeros(1 << 20, np.uint8)
    base, n = C.c_void_p(), C.c_size_t()
    _lib.check(L.pcx_host_register_mapping(C.c_void_p(a.ctypes.data), a.nbytes, 0, C.byref(base), C.byref(n)))
    assert not base.value and n.value == 0
    kind = C.c_int()
    _lib.check(L.pcx_pointer_kind(C.c_void_p(a.ctypes.data), C.byref(kind)))
    assert kind.value == 0
    # a shared mapping beyond max_bytes is left alone as well
    circ = B.CircularBuffer(1 << 20)
    _lib.check(L.pcx_host_register_mapping(C.c_void_p(circ.base), 4096, 1 << 20, C.byref(base), C.byref(n)))
    assert not base.value                                  # the two halves together are 2 MiB
    _lib.check(L.pcx_host_register_mapping(C.c_void_p(circ.base + circ.size + 8192), 4096, 0, C.byref(base), C.byref(n)))
    assert base.value == circ.base and n.value == 2 * circ.size       # asked about the SECOND half: both come along
    _lib.check(L.pcx_host_register_mapping(C.c_void_p(circ.base), 4096, 0, C.byref(base), C.byref(n)))
    assert base.value == circ.base and n.value == 2 * circ.size       # held already: the same range, one more holder
    with pytest.raises(_lib.PcxError):
        _lib.check(L.pcx_host_unregister(C.c_void_p(circ.base + 4096)))      # not the base of a range this library locked
    kind = C.c_int()
    _lib.check(L.pcx_host_unregister(C.c_void_p(circ.base)))
    _lib.check(L.pcx_pointer_kind(C.c_void_p(circ.base), C.byref(kind)))
    assert kind.value == 1                                 # the first holder has let go, the second keeps it locked
    _lib.check(L.pcx_host_unregister(C.c_void_p(circ.base)))
    _lib.check(L.pcx_pointer_kind(C.c_void_p(circ.base), C.byref(kind)))
    assert kind.value == 0
    with pytest.raises(_lib.PcxError):
        _lib.check(L.pcx_host_unregister(C.c_void_p(circ.base)))             # nobody holds it any more
    circ.close()


def test_fir_block_follows_a_framework_that_reallocates_its_circular_buffer(oracle):
    """the port's address leaves what the block has page-locked (the framework re-allocated: a topology re-commit) -- the new mapping
    is locked on first sight, the block keeps at most four ranges (the oldest is unlocked), and results stay the reference's on
    every one of six buffers in turn, including one that is seen again after it has been evicted"""
    import ctypes as C

    from pothoscomms_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(12)
    K = 63
    taps = (rng.normal(size=K) + 1j * rng.normal(size=K)) / 8
    blk = B.make("/comms/fir_filter", "complex_float32", "COMPLEX")
    blk.call("setTaps", taps)
    blk.activate()
    n = 40000
    circs = [B.CircularBuffer(1 << 20) for _ in range(6)]
    kind = C.c_int()

    def locked(c):
        _lib.check(L.pcx_pointer_kind(C.c_void_p(c.base + 64), C.byref(kind)))
        return kind.value == 1

    for i in list(range(6)) + [0]:
        c = circs[i]
        x = rand_stream(rng, 1, n, True)
        off = (c.size // 8 - 1000) * 8 if i % 2 else 4096 * 8           # every other window across the wrap
        win = c.view(off, n * 8, np.float32).reshape(-1, 2)
        win[:] = x
        ref = oracle.Fir(1, True, True); ref.set_taps(taps); ref.activate()
        y, cc, p, r, _ = blk.work(win, n)
        ry, rc, rp, rr = ref.work(x, n)
        assert (cc, p, r) == (rc, rp, rr) and nerr(y, ry) <= TOL, i
        assert locked(c), i
    assert sum(locked(c) for c in circs) == 4                           # never more than four ranges held
    assert locked(circs[0]) and not locked(circs[1]) and not locked(circs[2])        # 0 came back (evicting 3's predecessor ...), the oldest went
    blk.close()
    assert not any(locked(c) for c in circs)
    for c in circs:
        c.close()
