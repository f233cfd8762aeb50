"""Pins the oracle -- on the build box (-m "not gpu") and again on the GPU box (-m gpu), see `_where`.

  * against the committed golden vectors (tests/golden/golden.npz: the reference tests' own
    known-answer data + outputs of the compiled reference, see make_golden.py),
  * against oracle/_ref itself when it is present (build container),
  * FIR / FreqDemod, for which the reference's tests hold no numeric vectors, against an
    independent float64 computation and the survey's observed anchors,
  * the work()-level host logic (reserve / consume / produce, burst flush, labels).
"""
import os

import numpy as np
import pytest

from tests.util import NAMES, SCALARS, nerr, rand_stream

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden.npz"))


# Every test of this module runs on BOTH boxes: as "[build-box]" under -m "not gpu" and as "[gpu-box]" under -m gpu.
# The oracle is the checker of the GPU parity tests, and on the GPU box libpcx_oracle.so binds THAT box's glibc and
# libgcc_s (sincos, hypotf, __divsc3: the last-bit effects ROUND_NOTES lists): it is re-pinned there, against the same
# golden vectors, before anything is compared with it.
@pytest.fixture(autouse=True, params=[pytest.param("build-box"), pytest.param("gpu-box", marks=pytest.mark.gpu)])
def _where(request):
    return request.param


TYPES = {"int8": np.int8, "int16": np.int16, "int32": np.int32, "int64": np.int64, "float32": np.float32, "float64": np.float64}


# ---- FFT ------------------------------------------------------------------------------
def test_fft_kat_float(oracle):
    """fft/TestFFT.cpp:14-80: |got - expected| < 0.01, inverse returns input * N."""
    x = GOLD["fft_kat_in"].astype(np.float32)
    want = GOLD["fft_kat_out"]
    got = oracle.fft(x, 4, False)
    assert np.max(np.abs(got - want)) < 0.01
    back = oracle.fft(want.astype(np.float32), 4, True)
    assert np.max(np.abs(back - 4 * GOLD["fft_kat_in"])) < 0.01


def test_fft_kat_int16(oracle):
    """fft/TestFFT.cpp:95-156: forward = result/N, inverse of that = input."""
    x = (GOLD["fft_kat_in"] * 1000).astype(np.int16)
    want = GOLD["fft_kat_out"] * 1000
    got = oracle.fft(x, 4, False)
    assert np.max(np.abs(got - want / 4)) < 0.01 + 0.5   # integers: the test's 0.01 applies to exact quarters
    assert np.array_equal(got, np.array([[100, 550], [250, 350], [0, 150], [50, -450]], np.int16))
    # (the test feeds the UNSCALED spectrum to the inverse; /N scaling returns the input)
    back = oracle.fft(want.astype(np.int16), 4, True)
    assert np.max(np.abs(back - x)) <= 1


@pytest.mark.parametrize("n", [2, 3, 4, 5, 8, 9, 15, 16, 20, 64, 100, 210, 256, 1024, 4096])
@pytest.mark.parametrize("inv", [0, 1])
def test_fft_matches_compiled_reference_vectors(oracle, n, inv):
    for kind in ("f32", "f64", "i16"):
        key = "fft_%s_in_%d_%d" % (kind, n, inv)
        if key not in GOLD:
            continue
        got = oracle.fft(GOLD[key], n, bool(inv))
        assert np.array_equal(got, GOLD["fft_%s_out_%d_%d" % (kind, n, inv)]), kind


# ---- fxpt_atan2 / getAngle / getAbs -----------------------------------------------------
def test_fxpt_atan2_vectors(oracle):
    L = oracle.lib()
    yx = GOLD["atan2_in"]
    got = np.array([L.orc_fxpt_atan2(int(y), int(x)) for y, x in yx], dtype=np.uint16)
    assert np.array_equal(got, GOLD["atan2_out"])


@pytest.mark.parametrize("name", list(TYPES))
def test_angle_abs_vectors(oracle, name):
    z = GOLD["rand_in_" + name]
    assert np.array_equal(oracle.angle(z), GOLD["rand_angle_" + name])
    assert np.array_equal(oracle.abs_(z, True), GOLD["rand_abs_cplx_" + name], equal_nan=True)
    assert np.array_equal(oracle.abs_(np.ascontiguousarray(z[:, 0]), False), GOLD["rand_abs_real_" + name])


@pytest.mark.parametrize("name", list(TYPES))
def test_angle_reference_test_points(oracle, name):
    """math/TestAngle.cpp:53-65 tolerances on its 13 points."""
    zin = GOLD["angle_in_" + name]
    got = oracle.angle(zin)
    assert np.array_equal(got, GOLD["angle_ref_" + name])
    expected = np.arctan2(zin[:, 1].astype(np.float64), zin[:, 0].astype(np.float64))
    if name.startswith("float"):
        assert np.max(np.abs(got - expected)) <= np.pi / 500
    elif name != "int8":   # the reference test skips int8
        err = (got.astype(np.int64) - np.round(expected * (1 << 15) / np.pi).astype(np.int64)).astype(np.int16)
        assert np.max(np.abs(err[1:])) <= (np.pi / 500) * (1 << 15) / np.pi + 1


@pytest.mark.parametrize("name", list(TYPES))
def test_abs_reference_test_points(oracle, name):
    """math/TestAbs.cpp:63-66,96-98: exact equality with getAbs."""
    av = GOLD["abs_in_" + name]
    assert np.array_equal(oracle.abs_(av, False), GOLD["abs_real_exp_" + name])
    assert np.array_equal(oracle.abs_(av.reshape(50, 2), True), GOLD["abs_cplx_exp_" + name])


# ---- Rotate / Scale / Conjugate ---------------------------------------------------------
@pytest.mark.parametrize("name", list(TYPES))
def test_rotate_reference_test_points(oracle, name):
    """math/TestRotate.cpp:50-53: POTHOS_TEST_CLOSE(out, expected, 1) for every type."""
    x = GOLD["rotate_in_" + name]
    for k, phase in enumerate([0.0, np.pi / 2, np.pi, 3 * np.pi / 2]):
        got = oracle.rotate(x, phase).astype(np.float64)
        exp = GOLD["rotate_exp_%s_%d" % (name, k)]
        if name == "int8":   # expected = std::complex<int8>(double): the cast wraps
            exp = exp.astype(np.int64).astype(np.int8).astype(np.float64)
            d = np.abs(got - exp)
            d = np.minimum(d, 256 - d)
        else:
            d = np.abs(got - exp)
        assert np.max(d) <= 1.0, (name, phase)
        if name.startswith("float"):
            assert np.max(d) <= 1e-4 * 240


@pytest.mark.parametrize("name", list(TYPES))
def test_scale_reference_test_points(oracle, name):
    """math/TestScale.cpp:49-52."""
    x = GOLD["scale_in_" + name]
    for k, factor in enumerate([-1.0, -0.5, 0.0, 0.5, 1.0]):
        got = oracle.scale(x, factor, False).astype(np.float64)
        exp = GOLD["scale_exp_%s_%d" % (name, k)]
        if name == "int8":
            exp = exp.astype(np.int64).astype(np.int8).astype(np.float64)
            d = np.abs(got - exp)
            d = np.minimum(d, 256 - d)
        else:
            d = np.abs(got - exp)
        assert np.max(d) <= 1.0, (name, factor)


def test_rotate_unset_phase_is_zero(oracle):
    """Rotate.cpp:60-62: _phasor is never set by the constructor -> zero output until setPhase."""
    x = GOLD["rotate_in_float32"]
    assert not np.any(oracle.rotate(x, None))


def test_conjugate_reference_test_points(oracle):
    """math/TestConjugate.cpp:64-66: exact equality with std::conj."""
    x = GOLD["conj_in"]
    got = oracle.conj(x)
    assert np.array_equal(got[:, 0], x[:, 0]) and np.array_equal(got[:, 1], -x[:, 1])
    xi = x.astype(np.int16)
    goti = oracle.conj(xi)
    assert np.array_equal(goti[:, 1], -xi[:, 1])


def test_coefficient_label_scan(oracle):
    """Rotate.cpp:105-123 / Scale.cpp:104-122."""
    # label at index 0 -> apply it, process everything
    assert oracle.coeff_label_scan(100, [0], [1]) == (100, 0)
    # first matching label at index 40 -> stop before it
    assert oracle.coeff_label_scan(100, [40, 60], [1, 1]) == (40, -1)
    # label at 0 then another at 30: apply the first, stop at the second
    assert oracle.coeff_label_scan(100, [0, 30], [1, 1]) == (30, 0)
    # labels past the buffer or with another id are ignored
    assert oracle.coeff_label_scan(100, [10, 150], [0, 1]) == (100, -1)
    # no label id configured
    assert oracle.coeff_label_scan(100, [0], [1], have_label_id=False) == (100, -1)


# ---- FIR ----------------------------------------------------------------------------------
def _f64_fir(x, taps, L=1, M=1):
    """Independent float64 polyphase FIR: zero-stuff by L, convolve, keep every M-th (valid part)."""
    xz = x[:, 0].astype(np.float64) + 1j * x[:, 1].astype(np.float64) if x.ndim == 2 else x.astype(np.float64)
    K = -(-len(taps) // L)
    up = np.zeros(len(xz) * L, dtype=xz.dtype)
    up[::L] = xz
    full = np.convolve(up, np.asarray(taps))
    # output i (flat index n*L + j) uses x[n-k], n counted from the sample after the K-1 history
    start = (K - 1) * L
    flat = full[start:start + (len(xz) - (K - 1)) * L]
    return flat[M - 1::M]


@pytest.mark.parametrize("L,M", [(1, 1), (1, 2), (3, 1), (3, 2), (2, 3)])
@pytest.mark.parametrize("kind", ["real-REAL", "complex-REAL", "complex-COMPLEX"])
def test_fir_float_against_float64_convolution(oracle, kind, L, M):
    is_complex, ctaps = kind != "real-REAL", kind == "complex-COMPLEX"
    rng = np.random.default_rng(L * 10 + M)
    ntaps = 101
    taps = rng.normal(size=ntaps) / 10 + (1j * rng.normal(size=ntaps) / 10 if ctaps else 0)
    x = rand_stream(rng, oracle.F32, 4096, is_complex)
    blk = oracle.Fir(oracle.F32, is_complex, ctaps)
    blk.set_taps(taps); blk.set_interpolation(L); blk.set_decimation(M); blk.activate()
    y, c, p, reserve = blk.work(x, 1 << 16)
    assert reserve == 0
    K = -(-ntaps // L)
    assert c == ((4096 - (K - 1)) // M) * M and p == (c // M) * L
    taps32 = taps.astype(np.complex64 if ctaps else np.float32)
    ref = _f64_fir(x, taps32.astype(np.complex128) if ctaps else taps32.astype(np.float64), L, M)[:p]
    got = y[:, 0] + 1j * y[:, 1] if is_complex else y
    assert np.max(np.abs(got - ref)) / np.max(np.abs(ref)) < 2e-6


def test_fir_anchors(oracle):
    """K=63 on 4096 -> 4034/4034; L=3, M=2, K=21, 4096-element output -> 2730/4095."""
    rng = np.random.default_rng(1)
    x = rand_stream(rng, oracle.F32, 4096, True)
    blk = oracle.Fir(oracle.F32, True, True); blk.set_taps(rng.normal(size=63)); blk.activate()
    _, c, p, r = blk.work(x, 1 << 16)
    assert (c, p, r) == (4034, 4034, 0) and blk.K == 63 and blk.input_require == 63
    blk = oracle.Fir(oracle.F32, True, True); blk.set_taps(rng.normal(size=61))
    blk.set_interpolation(3); blk.set_decimation(2); blk.activate()
    assert blk.K == 21
    _, c, p, _ = blk.work(x, 4096)
    assert (c, p) == (2730, 4095)


def test_fir_insufficient_input_sets_reserve(oracle):
    """FIRFilter.cpp:251-255."""
    blk = oracle.Fir(oracle.F32, True, True); blk.set_taps(np.ones(10)); blk.set_decimation(4); blk.activate()
    x = np.zeros((12, 2), np.float32)     # need M + K - 1 = 13
    _, c, p, r = blk.work(x, 100)
    assert (c, p, r) == (0, 0, 13)


def test_fir_wait_taps(oracle):
    """setWaitTaps/activate/setTaps, FIRFilter.cpp:128-144,201-209."""
    blk = oracle.Fir(oracle.F32, False, False)
    blk.set_wait_taps(True); blk.activate()
    x = np.ones(100, np.float32)
    _, c, p, r = blk.work(x, 100)
    assert (c, p, r) == (0, 0, None)          # armed: work() returns immediately
    blk.set_taps([0.5])
    y, c, p, _ = blk.work(x, 100)
    assert (c, p) == (100, 100) and np.allclose(y, 0.5)


def test_fir_burst_flush(oracle):
    """FIRFilter.cpp:218-272: a frame of B samples yields exactly B outputs, tail flushed with zeros."""
    rng = np.random.default_rng(2)
    taps = rng.normal(size=16)
    B = 50
    x = rand_stream(rng, oracle.F32, B + 30, False)
    blk = oracle.Fir(oracle.F32, False, False); blk.set_taps(taps); blk.set_frame_ids(True, False); blk.activate()
    # frame start label at index 0, length B (width 1)
    y1, c1, p1, _ = blk.work(x, 1000, labels=[("S", 0, 1, B)])
    assert c1 == p1 == B - 15                  # streaming part of the burst
    y2, c2, p2, _ = blk.work(x[c1:], 1000)     # tail shorter than M+K-1 -> zero padded flush
    assert c2 == p2 == 15
    burst = np.concatenate([x[:B], np.zeros(15, np.float32)])
    ref = np.convolve(burst.astype(np.float64), taps.astype(np.float32).astype(np.float64))[15:15 + B]
    assert np.max(np.abs(np.concatenate([y1, y2]) - ref)) < 1e-5
    # after the burst the block is back in streaming mode
    _, c3, p3, r3 = blk.work(x[c1 + c2:], 1000)
    assert c3 == p3 == 30 - 15


def test_fir_burst_waits_for_whole_frame(oracle):
    """FIRFilter.cpp:243-247: end of burst beyond the available input -> setReserve(eob), no work."""
    blk = oracle.Fir(oracle.F32, False, False); blk.set_taps(np.ones(4)); blk.set_frame_ids(True, True); blk.activate()
    x = np.ones(20, np.float32)
    _, c, p, r = blk.work(x, 100, labels=[("S", 5, 1, 100)])
    assert (c, p, r) == (0, 0, 105)
    # frame END label: eob = index + width
    blk.activate()
    _, c, p, r = blk.work(x, 100, labels=[("E", 9, 1, None)])
    assert c == p == 10 - 3


@pytest.mark.parametrize("scalar", [s for s in SCALARS if s >= 2], ids=lambda s: NAMES[s])
def test_fir_integer_matches_exact_integer_convolution(oracle, scalar):
    """Integer types: exact convolution modulo 2^bits(Q), >> bits(Q)/2, truncated (FIRFilter.cpp:295-300)."""
    rng = np.random.default_rng(scalar)
    x = rand_stream(rng, scalar, 300, True, amp=100)
    taps = rng.normal(size=9) * 0.3 + 1j * rng.normal(size=9) * 0.3
    blk = oracle.Fir(scalar, True, True); blk.set_taps(taps); blk.activate()
    y, c, p, _ = blk.work(x, 1000)
    qb = {oracle.I64: 64, oracle.I32: 64, oracle.I16: 32, oracle.I8: 16}[scalar]
    tq = [(int(np.ldexp(t.real, qb // 2)), int(np.ldexp(t.imag, qb // 2))) for t in taps]   # python ints: exact
    ebits = oracle.NP_SCALAR[scalar]().itemsize * 8
    def wrap(v, bits):
        v &= (1 << bits) - 1
        return v - (1 << bits) if v >> (bits - 1) else v
    for n in (0, 1, 100, p - 1):
        ar = ai = 0
        for k, (a, b) in enumerate(tq):
            cx, d = int(x[8 + n - k, 0]), int(x[8 + n - k, 1])
            ar += a * cx - b * d
            ai += a * d + b * cx
        assert int(y[n, 0]) == wrap(wrap(ar, qb) >> (qb // 2), ebits)
        assert int(y[n, 1]) == wrap(wrap(ai, qb) >> (qb // 2), ebits)


@pytest.mark.parametrize("name", ["float32", "float64"])
def test_rotate_scale_against_the_compiled_operators(oracle, name):
    """math/Rotate.cpp:15-23,71-75 and math/Scale.cpp:15-23 for float types = the compiled std::complex / scalar multiplies on a phasor
    that glibc's sincos gives (tests/golden/make_golden.py section 6): the oracle's loops AND its evaluation of std::polar, bit for bit"""
    x = GOLD["rotscale_in_" + name]
    for k, phase in enumerate(GOLD["rotscale_phases"]):
        assert np.array_equal(oracle.rotate(x, float(phase)), GOLD["rotate_out_%s_%d" % (name, k)]), phase
    for k, factor in enumerate(GOLD["rotscale_factors"]):
        assert np.array_equal(oracle.scale(x, float(factor), True), GOLD["scale_out_%s_%d" % (name, k)]), factor


FIR_FIXTURES = [("c0_63c_f32", True), ("c1_255c_f32", True), ("c4_127r_f32", False), ("2049c_f32", True), ("31c_f64", True)]


@pytest.mark.parametrize("key,ctaps", FIR_FIXTURES)
def test_fir_against_the_compiled_complex_multiply_accumulate(oracle, key, ctaps):
    """filter/FIRFilter.cpp:294-300 for float element types = std::complex operator* and operator+=, k ascending from y_n = 0, on taps
    narrowed once (:348): the fixture composes the COMPILED operators tap by tap (tests/golden/make_golden.py section 5, the BASELINE tap
    sets of configs[0] / [1] / [4]); the oracle's loop must give the same bits -- order, operators, narrowing -- in one call and in two"""
    x, taps, want = GOLD["fir_%s_in" % key], GOLD["fir_%s_taps" % key], GOLD["fir_%s_out" % key]
    sc = oracle.scalar_code(x)
    K, n = len(taps), want.shape[0]
    blk = oracle.Fir(sc, True, ctaps)
    blk.set_taps(taps)
    blk.activate()
    got, c, p, _ = blk.work(x, n)
    assert (c, p) == (n, n) and np.array_equal(got, want)
    blk2 = oracle.Fir(sc, True, ctaps)
    blk2.set_taps(taps)
    blk2.activate()
    n1 = n // 3
    g1, c1, p1, _ = blk2.work(x[:K - 1 + n1], n1)
    g2, c2, p2, _ = blk2.work(x[c1:], n - n1)           # the K-1 unconsumed samples stay in front as history (:305-307)
    assert np.array_equal(np.concatenate([g1[:p1], g2[:p2]]), want)


# ---- the Q-format reading as a parameter (include/pcx.h pcx_qformat, oracle.set_qformat) ------------------
def _py_scale(x, factor, q, ebits, qbits):
    """arrayScale under a reading, in Python integers: an independent model of orc_scale (no shifts of negative numbers, no C)"""
    import math
    from fractions import Fraction
    n = ebits // 2 if q[0] else qbits // 2
    v = math.ldexp(factor, n)
    fq = int(math.floor(abs(v) + 0.5)) * (1 if v >= 0 else -1) if q[1] else int(v)      # nearest, ties away | truncation

    def wrap(v, bits):
        v &= (1 << bits) - 1
        return v - (1 << bits) if v >> (bits - 1) else v
    out = []
    for xi in x.tolist():
        acc = wrap(wrap(fq, qbits) * xi, qbits)
        r = Fraction(acc, 1 << n)
        if q[2] == 0:
            y = math.floor(r)
        elif q[2] == 1:
            y = math.trunc(r)
        else:
            y = math.floor(r + Fraction(1, 2))
        out.append(wrap(y, ebits))
    return out


@pytest.mark.parametrize("q", [(f, t, r) for f in (0, 1) for t in (0, 1) for r in (0, 1, 2)], ids=str)
def test_qformat_readings_against_a_python_integer_model(oracle, q):
    rng = np.random.default_rng(sum(q) * 7 + q[0])
    oracle.set_qformat(*q)
    try:
        for dt, ebits, qbits in ((np.int8, 8, 16), (np.int16, 16, 32), (np.int32, 32, 64), (np.int64, 64, 64)):
            info = np.iinfo(dt)
            x = rng.integers(info.min, info.max + 1, 300, dtype=dt)
            x[:6] = [info.min, info.max, 5, -5, 1000 % (info.max + 1), -1]
            for factor in (0.5, -0.5, 0.3, -0.3337, 1.0, -1.0, 0.0, 77.77):
                assert oracle.scale(x, factor, False).tolist() == _py_scale(x, factor, q, ebits, qbits), (dt, factor)
    finally:
        oracle.set_qformat()


@pytest.mark.parametrize("q", [(f, t, r) for f in (0, 1) for t in (0, 1) for r in (0, 1, 2)], ids=str)
def test_reference_rotate_scale_points_hold_under_every_reading(oracle, q):
    """math/TestRotate.cpp:50-53, math/TestScale.cpp:49-52 (tolerance 1) cannot tell the twelve readings apart: every one passes"""
    oracle.set_qformat(*q)
    try:
        for name in ("int8", "int16", "int32", "int64"):
            x = GOLD["rotate_in_" + name]
            for k, phase in enumerate([0.0, np.pi / 2, np.pi, 3 * np.pi / 2]):
                exp = GOLD["rotate_exp_%s_%d" % (name, k)]
                got = oracle.rotate(x, phase).astype(np.float64)
                if name == "int8":
                    exp = exp.astype(np.int64).astype(np.int8).astype(np.float64)
                    d = np.abs(got - exp)
                    d = np.minimum(d, 256 - d)
                else:
                    d = np.abs(got - exp)
                assert d.max() <= 1.0, (name, phase)
            xs = GOLD["scale_in_" + name]
            for k, factor in enumerate([-1.0, -0.5, 0.0, 0.5, 1.0]):
                exp = GOLD["scale_exp_%s_%d" % (name, k)]
                got = oracle.scale(xs, factor, False).astype(np.float64)
                if name == "int8":
                    exp = exp.astype(np.int64).astype(np.int8).astype(np.float64)
                    d = np.abs(got - exp)
                    d = np.minimum(d, 256 - d)
                else:
                    d = np.abs(got - exp)
                assert d.max() <= 1.0, (name, factor)
    finally:
        oracle.set_qformat()


def test_qformat_hand_computed_cases(oracle):
    """int16 scale: 0.5 x +-5 = +-2.5 -> floor 2/-3, toward zero 2/-2, nearest 3/-2; 0.3 x 1000 under the four tap quantisations"""
    cases = [((0, 0, 0), 0.5, [5, -5], [2, -3]), ((0, 0, 1), 0.5, [5, -5], [2, -2]), ((0, 0, 2), 0.5, [5, -5], [3, -2]),
             ((0, 0, 0), 0.3, [1000], [299]), ((0, 1, 0), 0.3, [1000], [300]), ((1, 0, 0), 0.3, [1000], [296]), ((1, 1, 0), 0.3, [1000], [300])]
    try:
        for q, factor, xin, want in cases:
            oracle.set_qformat(*q)
            assert oracle.scale(np.array(xin, np.int16), factor, False).tolist() == want, (q, factor)
    finally:
        oracle.set_qformat()


# ---- FreqDemod -----------------------------------------------------------------------------
def test_freqdemod_anchor(oracle):
    """polar(1, 0.3 i^2) -> 0, 0.3, 0.9, 1.5, 2.1, 2.7, -2.983185, -2.383185."""
    i = np.arange(8, dtype=np.float64)
    z = np.exp(1j * 0.3 * i * i).astype(np.complex64)
    got = oracle.FreqDemod(oracle.F32).work(z)
    want = np.array([0, 0.3, 0.9, 1.5, 2.1, 2.7, -2.983185, -2.383185])
    assert np.max(np.abs(got - want)) < 2e-6


def test_freqdemod_state_carries_and_resets(oracle):
    rng = np.random.default_rng(3)
    x = rand_stream(rng, oracle.F32, 1000, True)
    whole = oracle.FreqDemod(oracle.F32).work(x)
    blk = oracle.FreqDemod(oracle.F32)
    parts = np.concatenate([blk.work(x[:1]), blk.work(x[1:400]), blk.work(x[400:])])
    assert np.array_equal(whole, parts)
    blk.activate()
    assert np.array_equal(blk.work(x[:10]), whole[:10])


@pytest.mark.parametrize("name", list(TYPES))
def test_freqdemod_against_the_compiled_reference_pieces(oracle, name):
    """demod/FreqDemod.cpp:60-67 = std::complex<T> operator* + getAngle (FxptHelpers.hpp:14-29) + std::conj, _prev = 0 at activate():
    the fixture composes the COMPILED pieces (tests/golden/make_golden.py section 4); the oracle's loop must reproduce it bit for bit,
    in one call and with the state carried over calls -- the first sample against _prev = 0, the integer products wrapping in
    complex<intN> before getAngle truncates them to int16."""
    x, want = GOLD["freqdemod_in_" + name], GOLD["freqdemod_out_" + name]
    sc = oracle.scalar_code(x)
    assert np.array_equal(oracle.FreqDemod(sc).work(x), want)
    blk = oracle.FreqDemod(sc)
    parts = np.concatenate([blk.work(x[:1]), blk.work(x[1:777]), blk.work(x[777:])])
    assert np.array_equal(parts, want)
    assert want[0] == 0                       # anything times _prev = 0 has angle 0 (getAngle(0) = 0 for every type)


def test_fft_work_is_one_frame_per_call(oracle):
    """FFT.cpp:61-72: consume/produce exactly numBins whatever is queued."""
    import ctypes as C
    L = oracle.lib()
    h = L.orc_fft_create(oracle.F32, 4, 0)
    x = np.arange(16, dtype=np.float32).reshape(8, 2)
    y = np.zeros_like(x)
    c, p = C.c_size_t(), C.c_size_t()
    L.orc_fft_work(h, x.ctypes.data, y.ctypes.data, C.byref(c), C.byref(p))
    L.orc_fft_destroy(h)
    assert (c.value, p.value) == (4, 4) and not np.any(y[4:])


def test_factory_type_matrix(oracle):
    """unsupported combinations are rejected as the factories do (FIRFilter.cpp:383, FFT.cpp:92)."""
    with pytest.raises(ValueError):
        oracle.Fir(oracle.F32, False, True)       # COMPLEX taps on a real stream
    with pytest.raises(ValueError):
        oracle.fft(np.zeros((4, 2), np.int32), 4)  # only double/float/int16


# ---- against the compiled reference directly (build container only) -------------------------
needs_ref = pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(os.path.dirname(__file__)), "oracle", "_ref", "libpcx_ref.so")),
                               reason="oracle/_ref not built here")


@needs_ref
@pytest.mark.parametrize("n", [1, 2, 6, 7, 12, 25, 27, 30, 32, 49, 60, 121, 128, 360, 1000, 2048])
def test_fft_bit_exact_vs_compiled_reference(oracle, n):
    rng = np.random.default_rng(n)
    for inv in (False, True):
        x = rng.uniform(-1, 1, (3 * n, 2)).astype(np.float32)
        assert np.array_equal(oracle.fft(x, n, inv), oracle.ref_fft(x, n, inv))
        x = rng.uniform(-1, 1, (3 * n, 2))
        assert np.array_equal(oracle.fft(x, n, inv), oracle.ref_fft(x, n, inv))
        x = rng.integers(-32768, 32768, (3 * n, 2)).astype(np.int16)
        assert np.array_equal(oracle.fft(x, n, inv), oracle.ref_fft(x, n, inv))


@needs_ref
def test_atan2_exhaustive_slice_vs_compiled_reference(oracle):
    L, R = oracle.lib(), oracle.ref()
    rng = np.random.default_rng(0)
    ys, xs = rng.integers(-32768, 32768, 50000), rng.integers(-32768, 32768, 50000)
    assert all(L.orc_fxpt_atan2(int(y), int(x)) == R.ref_fxpt_atan2(int(y), int(x)) for y, x in zip(ys, xs))
    for v in range(-300, 300):   # near the axes and diagonals
        for w in (-32768, -1, 0, 1, 32767, v, -v):
            assert L.orc_fxpt_atan2(v, w) == R.ref_fxpt_atan2(v, w)
            assert L.orc_fxpt_atan2(w, v) == R.ref_fxpt_atan2(w, v)


def test_synthetic_stream_is_deterministic(oracle):
    a = oracle.fill_uniform_f32(1000, 2, 0)
    b = oracle.fill_uniform_f32(500, 2, 500)
    assert np.array_equal(a[500:], b) and a.min() >= -1 and a.max() < 1
    assert abs(float(a.mean())) < 0.1



def test_fft_int16_one_bin_against_compiled_reference(oracle):
    """numBins = 1 in Q15 is x * 32767/32768 rounded, not a copy (kiss_fft.c:202-235 with p = 1)"""
    x = np.array([[5836, -18035], [123, -1], [-32768, 32767], [1, 0]], np.int16)
    want = [[5836, -18034], [123, -1], [-32767, 32766], [1, 0]]
    assert oracle.fft(x, 1, False).tolist() == want
    if oracle.ref() is not None:
        assert oracle.ref_fft(x, 1, False).tolist() == want and oracle.ref_fft(x, 1, True).tolist() == want


def test_rotate_phasor_is_sincos(oracle):
    """std::polar(1.0, phase) in an optimised build = glibc sincos(); its sine differs from sin() in the last
    bit for this phase (found by the GPU soak run) -- the oracle pins the sincos() value"""
    import ctypes as C
    phase = 2.747554270528532
    y = oracle.rotate(np.array([[1.0, 0.0]]), phase)
    libm = C.CDLL("libm.so.6")
    libm.sincos.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    s, c = C.c_double(), C.c_double()
    libm.sincos(phase, C.byref(s), C.byref(c))
    assert y[0, 0] == c.value and y[0, 1] == s.value
    assert repr(float(y[0, 1])) == "0.3839204418969204"
