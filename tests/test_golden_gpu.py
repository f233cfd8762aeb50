"""GPU suite: the HIP path against the committed golden vectors DIRECTLY -- no oracle in between.

tests/golden/golden.npz holds the reference tests' own known-answer data and outputs of the compiled reference
(oracle/_ref: fft/kissfft.hh, fft/kiss_fft.c, functions/fxpt_atan2.cpp built from /root/reference, see
tests/golden/make_golden.py).  Integer results must be bit-identical; float32 / float64 transforms within the 1e-5
(north_star) / 1e-13 bars of the largest reference sample, float angles within 1e-5 of pi."""
import os

import numpy as np
import pytest

from tests.util import TOL, ang_err, nerr

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden.npz"))
TYPES = ["int8", "int16", "int32", "int64", "float32", "float64"]
FFT_SIZES = [2, 3, 4, 5, 8, 9, 15, 16, 20, 64, 100, 210, 256, 1024, 4096]


@pytest.mark.parametrize("inv", [0, 1])
@pytest.mark.parametrize("n", FFT_SIZES)
def test_fft_against_compiled_kissfft_outputs(dev, n, inv):
    """fft/kissfft.hh (float, double) and fft/kiss_fft.c (Q15) outputs on seeded frames"""
    seen = 0
    for kind, dtype, bar in (("f32", "complex_float32", TOL), ("f64", "complex_float64", 1e-13), ("i16", "complex_int16", 0)):
        key = "fft_%s_in_%d_%d" % (kind, n, inv)
        if key not in GOLD:
            continue
        seen += 1
        want = GOLD["fft_%s_out_%d_%d" % (kind, n, inv)]
        got = dev.Fft(dtype, n, bool(inv)).transform(GOLD[key])
        if bar == 0:
            assert np.array_equal(got, want), kind
        else:
            assert nerr(got, want) <= bar, kind
    assert seen


def test_fft_reference_test_known_answers(dev):
    """fft/TestFFT.cpp:14-29 (float) and :95-156 (int16: forward = DFT / N, inverse of the unscaled spectrum = input)"""
    x = GOLD["fft_kat_in"]
    want = GOLD["fft_kat_out"]
    got = dev.Fft("complex_float32", 4, False).transform(x.astype(np.float32))
    assert np.max(np.abs(got - want)) < 0.01                           # the reference test's own tolerance
    back = dev.Fft("complex_float32", 4, True).transform(want.astype(np.float32))
    assert np.max(np.abs(back - 4 * x)) < 0.01
    xi = (x * 1000).astype(np.int16)
    gi = dev.Fft("complex_int16", 4, False).transform(xi)
    assert np.array_equal(gi, np.array([[100, 550], [250, 350], [0, 150], [50, -450]], np.int16))
    bi = dev.Fft("complex_int16", 4, True).transform((want * 1000).astype(np.int16))
    assert np.max(np.abs(bi.astype(np.int32) - xi)) <= 1


def test_fxpt_atan2_through_int16_angle(dev):
    """functions/fxpt_atan2.cpp on 2048 (y, x) pairs: /comms/angle of complex_int16 is getAngle = fxpt_atan2(imag, real)
    reinterpreted as int16 (FxptHelpers.hpp:14-29)"""
    yx = GOLD["atan2_in"]
    z = np.ascontiguousarray(yx[:, ::-1])            # (re, im) = (x, y)
    got = dev.angle(z)
    assert np.array_equal(got.view(np.uint16), GOLD["atan2_out"])


@pytest.mark.parametrize("name", TYPES)
def test_angle_and_abs_against_compiled_reference_outputs(dev, name):
    z = GOLD["rand_in_" + name]
    ga = dev.angle(z)
    gc = dev.abs_(z, True)
    gr = dev.abs_(np.ascontiguousarray(z[:, 0]), False)
    if name.startswith("float"):
        assert ang_err(ga, GOLD["rand_angle_" + name]) <= TOL
        if name == "float64":
            assert nerr(gc, GOLD["rand_abs_cplx_" + name]) <= 4e-15       # device hypot vs glibc hypot: <= 2 ulp
        else:
            assert np.array_equal(gc, GOLD["rand_abs_cplx_" + name], equal_nan=True)
    else:
        assert np.array_equal(ga, GOLD["rand_angle_" + name])
        assert np.array_equal(gc, GOLD["rand_abs_cplx_" + name])
    assert np.array_equal(gr, GOLD["rand_abs_real_" + name])


@pytest.mark.parametrize("name", TYPES)
def test_reference_test_points_angle_abs(dev, name):
    """math/TestAngle.cpp:30-65 (13 points) and the inputs of math/TestAbs.cpp through the compiled getAngle / getAbs"""
    zin = GOLD["angle_in_" + name]
    g = dev.angle(zin)
    if name.startswith("float"):
        assert ang_err(g, GOLD["angle_ref_" + name]) <= TOL
    else:
        assert np.array_equal(g, GOLD["angle_ref_" + name])
    a = GOLD["abs_in_" + name]
    assert np.array_equal(dev.abs_(a, False), GOLD["abs_real_exp_" + name])
    zc = np.ascontiguousarray(a.reshape(-1, 2))
    gc = dev.abs_(zc, True)
    want = GOLD["abs_cplx_exp_" + name]
    if name == "float64":
        assert nerr(gc, want) <= 4e-15          # device hypot vs glibc hypot: <= 2 ulp
    else:
        assert np.array_equal(gc, want)         # getAbs compiled from the reference: float32 and the integers bit for bit


@pytest.mark.parametrize("name", ["float32", "float64"])
def test_rotate_scale_against_the_compiled_operators(dev, name):
    """math/Rotate.cpp:15-23 / math/Scale.cpp:15-23 on float types through the compiled std::complex and scalar multiplies
    (tests/golden/make_golden.py section 6): the device bit for bit, no oracle in between"""
    x = GOLD["rotscale_in_" + name]
    for k, phase in enumerate(GOLD["rotscale_phases"]):
        assert np.array_equal(dev.rotate(x, float(phase)), GOLD["rotate_out_%s_%d" % (name, k)]), phase
    for k, factor in enumerate(GOLD["rotscale_factors"]):
        assert np.array_equal(dev.scale(x, float(factor), True), GOLD["scale_out_%s_%d" % (name, k)]), factor


@pytest.mark.parametrize("key,ctaps", [("c0_63c_f32", True), ("c1_255c_f32", True), ("c4_127r_f32", False), ("2049c_f32", True), ("31c_f64", True)])
def test_fir_against_the_compiled_complex_multiply_accumulate(dev, key, ctaps):
    """filter/FIRFilter.cpp:294-300 as the compiled std::complex operator* / operator+= composed tap by tap in the loop's order
    (tests/golden/make_golden.py section 5; the tap sets of BASELINE configs[0], [1], [4]): the HIP path against that fixture directly --
    the EXACT kernel bit for bit, the frequency-domain and the FMA time-domain kernels within 1e-5 (float32) / 1e-13 (float64) of the
    largest reference sample"""
    x, taps, want = GOLD["fir_%s_in" % key], GOLD["fir_%s_taps" % key], GOLD["fir_%s_out" % key]
    n = want.shape[0]
    f = dev.FirFilter("complex_" + ("float32" if x.dtype == np.float32 else "float64"), "COMPLEX" if ctaps else "REAL")
    f.set_taps(taps)
    f.set_algo(dev._lib.FIR_EXACT)
    got, c, p = f.process(x, n)
    assert (c, p) == (n, n) and np.array_equal(got, want)
    bar = TOL if x.dtype == np.float32 else 1e-13
    for algo in (dev._lib.FIR_AUTO, dev._lib.FIR_DIRECT):
        f.set_algo(algo)
        got, c, p = f.process(x, n)
        assert (c, p) == (n, n) and nerr(got, want) <= bar, algo
        if algo == dev._lib.FIR_AUTO and x.dtype == np.float32:
            # the float32 fixtures are long enough to cross two block seams of the frequency-domain kernel (make_golden.py section 5):
            # it IS that kernel that was just compared with the reference operators' outputs, and the seams are inside the comparison
            assert f.last_algo == dev._lib.FIR_OLS_FFT
            K = len(taps)
            S = 4096 - (K - 1 + 15) // 16 * 16              # the block payload (fir_ols.hip launch_fir_cf32_ols4096)
            assert n > 2 * S, (n, S)
            for seam in (S, 2 * S):                          # and the samples either side of each seam, on their own (same absolute bar)
                assert np.max(np.abs(got[seam - 64:seam + 64].astype(np.float64) - want[seam - 64:seam + 64])) <= bar * np.max(np.abs(want)), seam


@pytest.mark.parametrize("name", TYPES)
def test_freqdemod_against_the_compiled_reference_pieces(dev, name):
    """demod/FreqDemod.cpp:60-67 as the compiled std::complex<T> operator* and the compiled getAngle (FxptHelpers.hpp:14-29) composed
    (tests/golden/make_golden.py section 4): the HIP kernel against that fixture directly -- integers bit for bit (the product wraps
    in complex<intN>, getAngle truncates it to int16), floats within 1e-5 of pi; in one call, and with _prev carried over three"""
    x, want = GOLD["freqdemod_in_" + name], GOLD["freqdemod_out_" + name]

    def close(got, ref):
        if name.startswith("int"):
            return np.array_equal(got, ref)
        return ang_err(got, ref) <= TOL
    assert close(dev.FreqDemod("complex_" + name).process(x), want)
    blk = dev.FreqDemod("complex_" + name)
    parts = np.concatenate([blk.process(x[:1]), blk.process(x[1:777]), blk.process(x[777:])])
    assert close(parts, want)
    assert parts[0] == 0                      # the first sample meets _prev = 0 (FreqDemod.cpp:44-47)
    blk.reset()
    assert close(blk.process(x[:100]), want[:100])


# ---- Rotate / Scale / Conjugate: the DEVICE against the reference tests' own known answers, with their own tolerances ----
def _close(got, exp, name):
    """POTHOS_TEST_CLOSE(out, expected, 1); the int8 expectation is std::complex<int8>(double), whose cast wraps"""
    got = np.asarray(got, np.float64)
    if name == "int8":
        exp = exp.astype(np.int64).astype(np.int8).astype(np.float64)
        d = np.abs(got - exp)
        return np.minimum(d, 256 - d)
    return np.abs(got - exp)


@pytest.mark.parametrize("name", TYPES)
def test_rotate_reference_test_points_on_the_device(dev, name):
    """math/TestRotate.cpp:28-32 (inputs), :50-53 (POTHOS_TEST_CLOSE(out, expected, 1) at phases 0, pi/2, pi, 3pi/2)"""
    x = GOLD["rotate_in_" + name]
    for k, phase in enumerate([0.0, np.pi / 2, np.pi, 3 * np.pi / 2]):
        d = _close(dev.rotate(x, phase), GOLD["rotate_exp_%s_%d" % (name, k)], name)
        assert np.max(d) <= 1.0, (name, phase)
        if name.startswith("float"):
            assert np.max(d) <= 1e-4 * 240          # what float arithmetic owes the exact rotation of +-240


@pytest.mark.parametrize("name", TYPES)
def test_scale_reference_test_points_on_the_device(dev, name):
    """math/TestScale.cpp:28-31 (inputs), :49-52 (POTHOS_TEST_CLOSE(out, expected, 1) at factors -1, -0.5, 0, 0.5, 1)"""
    x = GOLD["scale_in_" + name]
    for k, factor in enumerate([-1.0, -0.5, 0.0, 0.5, 1.0]):
        d = _close(dev.scale(x, factor, False), GOLD["scale_exp_%s_%d" % (name, k)], name)
        assert np.max(d) <= 1.0, (name, factor)


def test_conjugate_reference_test_points_on_the_device(dev):
    """math/TestConjugate.cpp:30-34 (inputs), :64-66: exact equality with std::conj, float and integer"""
    x = GOLD["conj_in"]
    got = dev.conj(x)
    assert np.array_equal(got[:, 0], x[:, 0]) and np.array_equal(got[:, 1], -x[:, 1])
    for dt in (np.int8, np.int16, np.int32, np.int64, np.float64):
        xi = x.astype(dt)
        gi = dev.conj(xi)
        assert np.array_equal(gi[:, 0], xi[:, 0]) and np.array_equal(gi[:, 1], (-xi[:, 1].astype(np.int64)).astype(dt) if dt != np.float64 else -xi[:, 1])
