"""GPU parity: the HIP path, called through the C ABI (include/pcx.h), against the CPU oracle
on the same seeded inputs.  Bars (north_star): bit-exact for integer work and for the float
maps whose arithmetic order is preserved (Rotate/Scale/Conjugate/Abs/EXACT FIR); 1e-5 of
max|ref| for the float kernels that reorder/fuse arithmetic (FIR direct/OLS, FFT); 1e-5*pi
angular for FreqDemod (libm vs device atan2).
"""
import os

import numpy as np
import pytest

from tests.util import NAMES, SCALARS, TOL, ang_err, nerr, rand_stream

pytestmark = pytest.mark.gpu


# --------------------------------------------------------------------------- #
# element-wise blocks
# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("scalar", SCALARS, ids=lambda s: NAMES[s])
@pytest.mark.parametrize("n", [0, 1, 13, 1000, 65537])
def test_conjugate(oracle, dev, scalar, n):
    rng = np.random.default_rng(n + scalar)
    x = rand_stream(rng, scalar, n, True)
    assert np.array_equal(dev.conj(x), oracle.conj(x))


@pytest.mark.parametrize("scalar", SCALARS, ids=lambda s: NAMES[s])
@pytest.mark.parametrize("phase", [None, 0.0, 0.7, np.pi / 2, np.pi, 3 * np.pi / 2, -2.5])
def test_rotate(oracle, dev, scalar, phase):
    rng = np.random.default_rng(7 + scalar)
    for n in (1, 13, 4099):
        x = rand_stream(rng, scalar, n, True)
        assert np.array_equal(dev.rotate(x, phase), oracle.rotate(x, phase)), (n, phase)


@pytest.mark.parametrize("scalar", SCALARS, ids=lambda s: NAMES[s])
@pytest.mark.parametrize("is_complex", [False, True])
@pytest.mark.parametrize("factor", [-1.0, -0.5, 0.0, 0.5, 1.0, 3.14159, 1e-3])
def test_scale(oracle, dev, scalar, is_complex, factor):
    rng = np.random.default_rng(11 + scalar)
    for n in (1, 13, 4099):
        x = rand_stream(rng, scalar, n, is_complex)
        assert np.array_equal(dev.scale(x, factor, is_complex), oracle.scale(x, factor, is_complex))


@pytest.mark.parametrize("scalar", SCALARS, ids=lambda s: NAMES[s])
@pytest.mark.parametrize("is_complex", [False, True])
def test_abs(oracle, dev, scalar, is_complex):
    rng = np.random.default_rng(13 + scalar)
    for n in (1, 100, 70001):
        x = rand_stream(rng, scalar, n, is_complex)
        got, ref = dev.abs_(x, is_complex), oracle.abs_(x, is_complex)
        if scalar == oracle.F64 and is_complex:
            assert nerr(got, ref) <= 1e-15 * 4     # device hypot vs glibc hypot: <= 2 ulp
        else:
            assert np.array_equal(got, ref)         # float32 magnitude is bit-exact (fp64 sqrt path)


def test_abs_cf32_special_values(oracle, dev):
    x = np.array([[1e-30, 1e-30], [3e38, 3e38], [1e-45, 0], [0, 0], [-0.0, 0.0], [np.inf, 1], [np.inf, np.nan],
                  [3, 4], [-5, 12]], np.float32)
    got, ref = dev.abs_(x, True), oracle.abs_(x, True)
    assert np.array_equal(got, ref, equal_nan=True)


@pytest.mark.parametrize("scalar", SCALARS, ids=lambda s: NAMES[s])
def test_freqdemod(oracle, dev, scalar):
    rng = np.random.default_rng(17 + scalar)
    n = 50001
    if scalar in (oracle.F64, oracle.F32):
        # FM-like signal with non-vanishing envelope so atan2 is well conditioned
        ph = np.cumsum(rng.uniform(-1.5, 1.5, n))
        x = np.stack([np.cos(ph), np.sin(ph)], 1) * rng.uniform(0.5, 1.5, (n, 1))
        x = x.astype(oracle.NP_SCALAR[scalar])
    else:
        x = rand_stream(rng, scalar, n, True)
    # several work() calls of ragged sizes: _prev must carry across calls
    ref_blk, gpu_blk = oracle.FreqDemod(scalar), dev.FreqDemod((scalar, True))
    cuts = [0, 1, 2, 7, 4096, 4097, 30000, n]
    for a, b in zip(cuts[:-1], cuts[1:]):
        ref, got = ref_blk.work(x[a:b]), gpu_blk.process(x[a:b])
        if scalar in (oracle.F64, oracle.F32):
            assert ang_err(got, ref) <= TOL, (a, b)
        else:
            assert np.array_equal(got, ref), (a, b)
    # activate() resets _prev
    ref_blk.activate(); gpu_blk.reset()
    ref, got = ref_blk.work(x[:100]), gpu_blk.process(x[:100])
    if scalar in (oracle.F64, oracle.F32):
        assert ang_err(got, ref) <= TOL
    else:
        assert np.array_equal(got, ref)


def test_freqdemod_first_sample_quadrants(oracle, dev):
    """_prev = 0 at activate: the first output is arg(+-0 +-0j), whose value depends on signs."""
    for re, im in [(1, 0), (-1, 0), (0, 1), (0, -1), (1, 1), (-1, 1), (-1, -1), (1, -1)]:
        x = np.array([[re, im], [0.3, 0.4]], np.float32)
        ref, got = oracle.FreqDemod(oracle.F32).work(x), dev.FreqDemod("complex_float32").process(x)
        assert ang_err(got, ref) <= TOL, (re, im, got, ref)


# --------------------------------------------------------------------------- #
# FIR
# --------------------------------------------------------------------------- #
def _taps(rng, n, cplx):
    t = rng.normal(size=n) / np.sqrt(n)
    return t + 1j * rng.normal(size=n) / np.sqrt(n) if cplx else t


@pytest.mark.parametrize("scalar", SCALARS, ids=lambda s: NAMES[s])
@pytest.mark.parametrize("kind", ["real-REAL", "complex-REAL", "complex-COMPLEX"])
@pytest.mark.parametrize("L,M", [(1, 1), (1, 3), (3, 1), (3, 2), (2, 3)])
def test_fir_all_types_polyphase(oracle, dev, scalar, kind, L, M):
    """Every combination FIRFilterFactory accepts (FIRFilter.cpp:369-384) with rational resampling."""
    is_complex, ctaps = kind != "real-REAL", kind == "complex-COMPLEX"
    rng = np.random.default_rng(1000 * scalar + 10 * L + M)
    ntaps = 21
    taps = _taps(rng, ntaps, ctaps) * (0.5 if scalar in (oracle.F64, oracle.F32) else 0.9)
    x = rand_stream(rng, scalar, 4096, is_complex)
    ref_blk = oracle.Fir(scalar, is_complex, ctaps)
    gpu_blk = dev.FirFilter((scalar, is_complex), "COMPLEX" if ctaps else "REAL")
    for b in (ref_blk, gpu_blk):
        b.set_taps(taps); b.set_decimation(M); b.set_interpolation(L)
    ref_blk.activate()
    out_cap = 3 * 4096
    ref, rc, rp, _ = ref_blk.work(x, out_cap)
    floats = scalar in (oracle.F64, oracle.F32)
    gpu_blk.set_algo(dev._lib.FIR_EXACT)
    got, gc, gp = gpu_blk.process(x, out_cap)
    assert (gc, gp) == (rc, rp)
    assert np.array_equal(got, ref)            # bit-exact: integer ring arithmetic / unfused float order
    if floats:
        gpu_blk.set_algo(dev._lib.FIR_DIRECT)  # same order with FMA
        got, gc, gp = gpu_blk.process(x, out_cap)
        assert (gc, gp) == (rc, rp)
        assert nerr(got, ref) <= (TOL if scalar == oracle.F32 else 1e-13)


def test_fir_anchors(oracle, dev):
    """SURVEY appendix A anchors: K=63 on 4096 -> 4034/4034; L=3, M=2, 21 taps -> 2730/4095."""
    rng = np.random.default_rng(5)
    x = rand_stream(rng, oracle.F32, 4096, True)
    f = dev.FirFilter("complex_float32", "COMPLEX")
    f.set_taps(_taps(rng, 63, True))
    _, c, p = f.process(x, 1 << 20)
    assert (c, p) == (4034, 4034)
    f = dev.FirFilter("complex_float32", "COMPLEX")
    f.set_taps(_taps(rng, 61, True)); f.set_interpolation(3); f.set_decimation(2)
    assert f.K == 21
    _, c, p = f.process(x, 4096)      # the probe's output buffer held 4096 elements
    assert (c, p) == (2730, 4095)


@pytest.mark.parametrize("algo", ["DIRECT", "OLS_FFT", "AUTO"])
@pytest.mark.parametrize("ntaps", [1, 2, 7, 8, 9, 63, 64, 127, 255, 256, 257, 1000, 2049, 2050, 3000, 4097, 4098, 6145, 6146, 8192, 8193])
def test_fir_cf32_fast_paths(oracle, dev, algo, ntaps):
    """LDS-tiled direct kernel and frequency-domain overlap-save kernel vs the oracle."""
    if algo == "DIRECT" and ntaps > 2049:
        pytest.skip("direct tile plan")
    rng = np.random.default_rng(ntaps)
    n = 3 * 4096 + 777 + ntaps + (40000 if ntaps > 2049 else 0)   # beyond 2049 taps: 2 ... 4 partitions, runs of several blocks
    x = rand_stream(rng, oracle.F32, n, True)
    taps = _taps(rng, ntaps, True)
    ref_blk = oracle.Fir(oracle.F32, True, True); ref_blk.set_taps(taps); ref_blk.activate()
    ref, rc, rp, _ = ref_blk.work(x, n)
    f = dev.FirFilter("complex_float32", "COMPLEX"); f.set_taps(taps)
    f.set_algo(getattr(dev._lib, "FIR_" + algo))
    got, gc, gp = f.process(x, n)
    assert (gc, gp) == (rc, rp) == (n - ntaps + 1, n - ntaps + 1)
    assert nerr(got, ref) <= TOL, f.last_algo
    if algo == "AUTO" and ntaps > 1:
        assert f.last_algo == dev._lib.FIR_OLS_FFT   # 4096-sample blocks; beyond 2049 taps the taps in partitions (fir_ols_part.hip)


def test_fir_cf32_real_taps_fast(oracle, dev):
    rng = np.random.default_rng(3)
    x = rand_stream(rng, oracle.F32, 20000, True)
    taps = _taps(rng, 127, False)
    ref_blk = oracle.Fir(oracle.F32, True, False); ref_blk.set_taps(taps); ref_blk.activate()
    ref, rc, rp, _ = ref_blk.work(x, 20000)
    for algo in ("FIR_DIRECT", "FIR_OLS_FFT"):
        f = dev.FirFilter("complex_float32", "REAL"); f.set_taps(taps); f.set_algo(getattr(dev._lib, algo))
        got, gc, gp = f.process(x, 20000)
        assert (gc, gp) == (rc, rp)
        assert nerr(got, ref) <= TOL


def test_fir_streaming_equivalence(oracle, dev):
    """Repeated work() calls with the K-1 tail left un-consumed == one big call (FIRFilter.cpp:305-308)."""
    rng = np.random.default_rng(9)
    ntaps, n = 255, 40000
    x = rand_stream(rng, oracle.F32, n, True)
    taps = _taps(rng, ntaps, True)
    f = dev.FirFilter("complex_float32", "COMPLEX"); f.set_taps(taps)
    whole, _, _ = f.process(x, n)
    pos, outs = 0, []
    for avail in (300, 5000, 254, 255, 256, 12345, n):
        chunk = x[pos:min(n, pos + avail)]
        y, c, p = f.process(chunk, 1 << 20)
        outs.append(y); pos += c
    y, c, p = f.process(x[pos:], 1 << 20)
    outs.append(y); pos += c
    got = np.concatenate(outs)
    assert got.shape == whole.shape
    assert nerr(got, whole) <= 2e-6   # block boundaries differ between calls (OLS), arithmetic does not


def test_fir_output_capacity_limits_consumption(oracle, dev):
    rng = np.random.default_rng(10)
    x = rand_stream(rng, oracle.F32, 5000, True)
    f = dev.FirFilter("complex_float32", "COMPLEX"); f.set_taps(_taps(rng, 31, True)); f.set_decimation(4)
    ref = oracle.Fir(oracle.F32, True, True); ref.set_taps(np.ones(31)); ref.set_decimation(4); ref.activate()
    for cap in (0, 1, 7, 100, 10000):
        _, rc, rp, _ = ref.work(x, cap)
        _, gc, gp = f.process(x, cap)
        assert (gc, gp) == (rc, rp)


def test_fir_errors(dev):
    with pytest.raises(ValueError):
        dev.FirFilter("float32", "COMPLEX")           # FIRFilterFactory: unsupported types
    f = dev.FirFilter("complex_float32", "COMPLEX")
    with pytest.raises(ValueError):
        f.set_taps([])                                 # "taps cannot be empty"
    with pytest.raises(ValueError):
        f.set_decimation(0)
    with pytest.raises(ValueError):
        f.set_interpolation(0)
    # default taps {1.0}: pass-through (ctor setTaps, FIRFilter.cpp:125)
    x = np.arange(20, dtype=np.float32).reshape(10, 2)
    y, c, p = f.process(x, 100)
    assert (c, p) == (10, 10) and np.array_equal(y, x)


# --------------------------------------------------------------------------- #
# FFT
# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("inverse", [False, True])
@pytest.mark.parametrize("nbins", [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384])
def test_fft_cf32(oracle, dev, nbins, inverse):
    rng = np.random.default_rng(nbins)
    nframes = 5 if nbins >= 1024 else 37
    x = rand_stream(rng, oracle.F32, nbins * nframes, True)
    ref = oracle.fft(x, nbins, inverse)
    got = dev.Fft("complex_float32", nbins, inverse).transform(x)
    assert nerr(got, ref) <= TOL


@pytest.mark.parametrize("inverse", [False, True])
@pytest.mark.parametrize("nbins", [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384])
def test_fft_cf64(oracle, dev, nbins, inverse):
    """16 ... 8192 bins: the radix-16 plan in double precision (fft_r16_f64.hip); frame counts that leave
    the last workgroup's group of frames ragged"""
    rng = np.random.default_rng(nbins + 1)
    nframes = 3 if nbins >= 1024 else 37
    x = rand_stream(rng, oracle.F64, nbins * nframes, True)
    ref = oracle.fft(x, nbins, inverse)
    got = dev.Fft("complex_float64", nbins, inverse).transform(x)
    assert nerr(got, ref) <= 1e-13


@pytest.mark.parametrize("inverse", [False, True])
@pytest.mark.parametrize("nbins", [2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384])
def test_fft_int16_bit_exact(oracle, dev, nbins, inverse):
    """kiss_fft -DFIXED_POINT=16: every Q15 rounding reproduced (scaled by 1/N overall).  4^s sizes run
    the fused-pair kernel, 2*4^s sizes the pass kernel."""
    rng = np.random.default_rng(nbins + 2)
    x = rand_stream(rng, oracle.I16, nbins * 3, True)
    ref = oracle.fft(x, nbins, inverse)
    got = dev.Fft("complex_int16", nbins, inverse).transform(x)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("nbins,nframes", [(16, 1000), (64, 333), (256, 77), (32, 515), (1024, 9)])
def test_fft_int16_many_frames(oracle, dev, nbins, nframes):
    """frame counts that do not fill the last workgroup (several frames share one for short transforms)"""
    rng = np.random.default_rng(nbins)
    x = rand_stream(rng, oracle.I16, nbins * nframes, True)
    assert np.array_equal(dev.Fft("complex_int16", nbins, False).transform(x), oracle.fft(x, nbins, False))


@pytest.mark.parametrize("inverse", [False, True])
@pytest.mark.parametrize("nbins,nframes", [(16, 1000), (16, 256), (32, 515), (64, 333), (64, 64), (128, 77)])
def test_fft_f32_short_frames_staged_io(oracle, dev, nbins, nframes, inverse):
    """numBins <= 64 copy 256 lanes' frames through LDS with lane-contiguous accesses: full and ragged groups"""
    rng = np.random.default_rng(nbins + nframes)
    x = rand_stream(rng, oracle.F32, nbins * nframes, True)
    assert nerr(dev.Fft("complex_float32", nbins, inverse).transform(x), oracle.fft(x, nbins, inverse)) <= TOL


def test_fft_kat_reference_vectors(dev):
    """fft/TestFFT.cpp:14-29 (float) and :95-105,131-132 (int16, result/N)."""
    x = np.array([[0.4, 0.6], [-0.7, 0.6], [-0.2, 0.8], [0.9, 0.2]], np.float32)
    want = np.array([[0.4, 2.2], [1.0, 1.4], [0.0, 0.6], [0.2, -1.8]], np.float32)
    got = dev.Fft("complex_float32", 4, False).transform(x)
    assert np.max(np.abs(got - want)) < 0.01
    back = dev.Fft("complex_float32", 4, True).transform(want)
    assert np.max(np.abs(back - 4 * x)) < 0.01
    xi = (x * 1000).astype(np.int16)
    goti = dev.Fft("complex_int16", 4, False).transform(xi)
    assert np.array_equal(goti, np.array([[100, 550], [250, 350], [0, 150], [50, -450]], np.int16))


def test_fft_many_frames_4096(oracle, dev):
    """more frames than the launch has workgroups: exercises the grid-stride frame loop"""
    rng = np.random.default_rng(4)
    nframes = 2048 + 300
    x = rand_stream(rng, oracle.F32, 4096 * nframes, True)
    got = dev.Fft("complex_float32", 4096, False).transform(x)
    # spot-check frames across the whole range against the oracle
    for f in (0, 1, 2047, 2048, 2049, nframes - 1):
        ref = oracle.fft(x[f * 4096:(f + 1) * 4096], 4096, False)
        assert nerr(got[f * 4096:(f + 1) * 4096], ref) <= TOL
    # round trip: ifft(fft(x)) = N x
    back = dev.Fft("complex_float32", 4096, True).transform(got)
    assert nerr(back / 4096.0, x) <= TOL


def test_fft_errors(dev):
    with pytest.raises(ValueError):
        dev.Fft("complex_int32", 64)       # FFTFactory: unsupported type
    with pytest.raises(NotImplementedError):
        dev.Fft("complex_float32", 3 << 26)   # beyond every plan (2^26 bins); complex_int16 at 65,536 bins was the last size refused


@pytest.mark.parametrize("inverse", [False, True])
@pytest.mark.parametrize("dtype,nbins,nframes", [("complex_float32", 10243, 3), ("complex_float32", 2 * 10243, 2), ("complex_float64", 5147, 2),
                                                 ("complex_float64", 2 * 10243, 1)])
def test_fft_chirp_z_plan_for_sizes_without_another(oracle, dev, dtype, nbins, nframes, inverse):
    """FFTFactory takes any numBins (FFT.cpp:83-93): a prime (or 2 x a prime) beyond one workgroup's LDS has no mixed-radix
    or four-step plan on the device and runs Bluestein's form on the power-of-two plans (fft_bluestein.hip)"""
    sc = oracle.F32 if dtype == "complex_float32" else oracle.F64
    x = rand_stream(np.random.default_rng(nbins + int(inverse)), sc, nbins * nframes, True)
    got = dev.Fft(dtype, nbins, inverse).transform(x)
    assert nerr(got, oracle.fft(x, nbins, inverse)) <= (TOL if sc == oracle.F32 else 1e-12)


@pytest.mark.parametrize("inverse", [False, True])
@pytest.mark.parametrize("dtype,nbins,nframes", [("complex_float32", 1 << 15, 3), ("complex_float32", 1 << 16, 2), ("complex_float32", 1 << 17, 2),
                                                 ("complex_float32", 1 << 20, 1), ("complex_float32", 1 << 22, 1), ("complex_float32", 1 << 23, 1),
                                                 ("complex_float64", 1 << 13, 3), ("complex_float64", 1 << 14, 3), ("complex_float64", 1 << 15, 2),
                                                 ("complex_float64", 1 << 16, 2), ("complex_float64", 1 << 17, 1), ("complex_float64", 1 << 20, 1),
                                                 ("complex_float64", 1 << 21, 1), ("complex_float64", 1 << 22, 1)])
def test_fft_four_step_sizes(oracle, dev, dtype, nbins, nframes, inverse):
    """Power-of-two transforms beyond one workgroup against kissfft's restatement; same 1e-5 bar (1e-13 for
    float64).  complex_float32: two passes at 2^15 / 2^16 (columns with strided tiles, rows with the transpose
    on their store), three to 2^22, the five-pass form beyond.  complex_float64: one workgroup to 2^13, the same two /
    three pass plans in double to 2^21 (fft_large_f64.hip), five passes beyond."""
    scalar = oracle.F32 if dtype.endswith("32") else oracle.F64
    rng = np.random.default_rng(nbins % 1000 + inverse)
    x = rand_stream(rng, scalar, nbins * nframes, True)
    ref = oracle.fft(x, nbins, inverse)
    got = dev.Fft(dtype, nbins, inverse).transform(x)
    assert nerr(got, ref) <= (TOL if scalar == oracle.F32 else 1e-13)


# --------------------------------------------------------------------------- #
# fused Rotate -> FIR -> FreqDemod
# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("algo", ["DIRECT", "OLS_FFT"])
@pytest.mark.parametrize("ntaps,ctaps", [(127, False), (63, True), (1, False), (2048, False)])
def test_fm_chain(oracle, dev, ntaps, ctaps, algo):
    from pothoscomms_amd import taps as tp
    rng = np.random.default_rng(ntaps)
    n = 3 * 2047 + 500 + ntaps + (9000 if ntaps > 1000 else 0)
    x = tp.fm_test_signal(n)
    taps = tp.lowpass(ntaps, 0.1) if not ctaps else tp.complex_bandpass(ntaps, 0.1, 0.03)
    phase = 0.7
    # the three reference blocks back to back (oracle)
    xr = oracle.rotate(x, phase)
    fir = oracle.Fir(oracle.F32, True, ctaps); fir.set_taps(taps); fir.activate()
    y, _, _, _ = fir.work(xr, n)
    ref = oracle.FreqDemod(oracle.F32).work(y)
    ch = dev.FmChain(); ch.set_phase(phase); ch.set_taps(taps, ctaps)
    ch.set_algo(getattr(dev._lib, "FIR_" + algo))
    got, c, p = ch.process(x, n)
    assert (c, p) == (n - ntaps + 1, n - ntaps + 1)
    assert ch.last_algo == getattr(dev._lib, "FIR_" + algo)
    assert ang_err(got, ref) <= TOL
    # split into two calls: carried state
    ch.reset()
    cut = 4000
    g1, c1, _ = ch.process(x[:cut], n)
    g2, _, _ = ch.process(x[c1:], n)
    assert ang_err(np.concatenate([g1, g2]), ref) <= TOL
    del rng


@pytest.mark.parametrize("ctaps", [False, True])
@pytest.mark.parametrize("ntaps", [2, 3, 16, 17, 18, 63, 255, 1000, 2049, 2050, 4097, 4098])
def test_fir_cf64_overlap_save(oracle, dev, ntaps, ctaps):
    """complex_float64, M = L = 1: the frequency-domain kernel in double (fir_ols_f64.hip) for 4 <= K <= 4097,
    the sliding-window kernel beyond; stream lengths that leave ragged first and last blocks"""
    rng = np.random.default_rng(7 * ntaps + ctaps)
    taps = _taps(rng, ntaps, ctaps)
    for n in (ntaps - 1 + 1, ntaps - 1 + 4095, 3 * 4096 + 1234 + ntaps):
        x = rand_stream(rng, oracle.F64, n, True)
        ref_blk = oracle.Fir(oracle.F64, True, ctaps)
        ref_blk.set_taps(taps); ref_blk.activate()
        ref, rc, rp, _ = ref_blk.work(x, n)
        f = dev.FirFilter((oracle.F64, True), "COMPLEX" if ctaps else "REAL")
        f.set_taps(taps)
        got, gc, gp = f.process(x, n)
        assert (gc, gp) == (rc, rp)
        assert f.last_algo == (dev._lib.FIR_OLS_FFT if 4 <= ntaps <= 4097 else dev._lib.FIR_DIRECT)
        assert nerr(got, ref) <= 1e-13, (ntaps, n)


@pytest.mark.parametrize("ctaps", [False, True])
@pytest.mark.parametrize("scalar_name", ["int16", "int8"])
@pytest.mark.parametrize("ntaps", [2, 63, 64, 95, 96, 255, 1000, 2049, 2050, 4097, 4098])
def test_fir_complex_integer_overlap_save_bit_exact(oracle, dev, ntaps, scalar_name, ctaps):
    """complex_int16 / complex_int8, M = L = 1: from 64 / 96 taps AUTO takes the double-precision overlap-save pipeline, whose
    rounded sums ARE the integer convolution (then the reference's wrap, fromQ shift and truncation): bit-exact against
    the oracle's ring arithmetic, random full-scale inputs and the extreme case (every sample -2^(bits-1), taps at +-0.4999)."""
    scalar = oracle.I16 if scalar_name == "int16" else oracle.I8
    rng = np.random.default_rng(11 * ntaps + ctaps)
    full = 32768 if scalar == oracle.I16 else 128
    for case in ("random", "extreme"):
        if case == "random":
            taps = _taps(rng, ntaps, ctaps) * 0.9
            n = 2 * 4096 + 777 + ntaps
            x = rng.integers(-full, full, size=(n, 2)).astype(np.int16 if scalar == oracle.I16 else np.int8)
        else:
            sgn = rng.choice([-1.0, 1.0], size=ntaps)
            taps = 0.4999 * sgn + (0.4999j * rng.choice([-1.0, 1.0], size=ntaps) if ctaps else 0)
            n = 4096 + 100 + ntaps
            x = np.full((n, 2), -full, dtype=np.int16 if scalar == oracle.I16 else np.int8)
            x[::7, 0] = full - 1
        ref_blk = oracle.Fir(scalar, True, ctaps)
        ref_blk.set_taps(taps); ref_blk.activate()
        ref, rc, rp, _ = ref_blk.work(x, n)
        f = dev.FirFilter((scalar, True), "COMPLEX" if ctaps else "REAL")
        f.set_taps(taps)
        got, gc, gp = f.process(x, n)
        assert (gc, gp) == (rc, rp)
        lo = 64 if scalar == oracle.I16 else 96
        if "PCX_OLS_INT_MIN" not in os.environ:   # the A/B switch moves the crossover
            assert f.last_algo == (dev._lib.FIR_OLS_FFT if lo <= ntaps <= 4097 else dev._lib.FIR_EXACT), (ntaps, f.last_algo)
        f.set_algo(dev._lib.FIR_OLS_FFT if ntaps <= 4097 else dev._lib.FIR_EXACT)     # and forced, below the crossover too
        got2, _, _ = f.process(x, n)
        assert np.array_equal(got2, ref), (case, ntaps, "forced")
        assert np.array_equal(got, ref), (case, ntaps)


@pytest.mark.parametrize("scalar_name", ["float64", "int16", "int8"])
@pytest.mark.parametrize("ntaps", [2, 23, 24, 47, 48, 255, 1000, 2049, 2050, 4097, 4098])
def test_fir_real_streams_on_the_double_pipeline(oracle, dev, ntaps, scalar_name):
    """REAL float64 / int16 / int8 streams (real taps, M = L = 1): two real blocks per double-precision transform from
    24 / 48 taps.  Integers bit-exact (random full-scale and all-extreme inputs), float64 within 1e-13; stream lengths
    that leave an odd number of blocks and ragged ends."""
    scalar = {"float64": oracle.F64, "int16": oracle.I16, "int8": oracle.I8}[scalar_name]
    rng = np.random.default_rng(13 * ntaps)
    npdt = {"float64": np.float64, "int16": np.int16, "int8": np.int8}[scalar_name]
    full = {"float64": 1, "int16": 32768, "int8": 128}[scalar_name]
    for case in ("random", "extreme"):
        for n in (ntaps, 3 * 4096 + 555 + ntaps, 2 * 3840 + ntaps - 1):
            if case == "random":
                taps = _taps(rng, ntaps, False) * 0.9
                x = rng.standard_normal(n) if scalar == oracle.F64 else rng.integers(-full, full, size=n)
            else:
                taps = 0.4999 * rng.choice([-1.0, 1.0], size=ntaps)
                x = np.full(n, -full, dtype=np.float64)
                x[::5] = full - (0 if scalar == oracle.F64 else 1)
            x = x.astype(npdt)
            ref_blk = oracle.Fir(scalar, False, False)
            ref_blk.set_taps(taps); ref_blk.activate()
            ref, rc, rp, _ = ref_blk.work(x, n)
            f = dev.FirFilter((scalar, False), "REAL")
            f.set_taps(taps)
            for forced in (False, True):
                if forced:
                    f.set_algo(dev._lib.FIR_OLS_FFT if 2 <= ntaps <= 4097 else dev._lib.FIR_EXACT)
                got, gc, gp = f.process(x, n)
                assert (gc, gp) == (rc, rp)
                if scalar == oracle.F64:
                    # normalised by the output scale the taps and samples can reach (a single cancelling output must not set it)
                    scale = max(float(np.max(np.abs(ref))) if rp else 0.0, 0.1 * float(np.sum(np.abs(taps))) * float(np.max(np.abs(x))))
                    assert rp == 0 or float(np.max(np.abs(got.astype(np.float64) - ref))) <= 1e-13 * scale, (case, ntaps, n, forced)
                else:
                    assert np.array_equal(got, ref), (case, ntaps, n, forced)
            if "PCX_OLS_REAL_MIN" not in os.environ:
                f2 = dev.FirFilter((scalar, False), "REAL"); f2.set_taps(taps); f2.process(x, n)
                lo = 24 if scalar == oracle.F64 else 48
                if rp > 0:
                    assert (f2.last_algo == dev._lib.FIR_OLS_FFT) == (lo <= ntaps <= 4097), (ntaps, f2.last_algo)


@pytest.mark.parametrize("ctaps", [False, True])
@pytest.mark.parametrize("M", [2, 4, 8, 16, 6, 10, 12, 48, 50, 100, 160])
@pytest.mark.parametrize("ntaps", [1, 2, 16, 17, 255, 1000, 2049])
def test_fir_cf32_decimating_folded_spectrum(oracle, dev, ntaps, M, ctaps):
    """complex_float32, interpolation 1, even decimation M = M1 * M2 (M1 = 16 / 8 / 4 / 2): one forward transform, the
    spectrum folded M1-fold, a 4096/M1-point inverse, one output in M2 stored (fir_ols_decim.hip).  Same outputs and consume/produce counts as the reference's decimator
    (FIRFilter.cpp:286-302) over stream lengths with ragged first / last blocks, chunked calls included."""
    rng = np.random.default_rng(17 * ntaps + M + ctaps)
    taps = _taps(rng, ntaps, ctaps)
    for n in (ntaps + M - 1, ntaps + 5 * M, 4096 + ntaps, 3 * 4096 + 777 + ntaps):
        x = rand_stream(rng, oracle.F32, n, True)
        ref_blk = oracle.Fir(oracle.F32, True, ctaps)
        ref_blk.set_taps(taps); ref_blk.set_decimation(M); ref_blk.activate()
        ref, rc, rp, _ = ref_blk.work(x, n)
        f = dev.FirFilter((oracle.F32, True), "COMPLEX" if ctaps else "REAL")
        f.set_taps(taps); f.set_decimation(M)
        got, gc, gp = f.process(x, n)
        assert (gc, gp) == (rc, rp), (ntaps, M, n)
        if rp:
            assert f.last_algo == dev._lib.FIR_OLS_FFT
            # the frequency-domain error is relative to the block's scale, not to one (possibly cancelling) output: a stream
            # that yields one or two outputs is normalised by at least a tenth of the typical output magnitude
            typical = float(np.sqrt(np.sum(np.abs(taps) ** 2)) * np.sqrt(np.mean(x.astype(np.float64) ** 2) * 2))
            scale = max(float(np.max(np.abs(ref))), 0.1 * typical)
            assert float(np.max(np.abs(got.astype(np.float64) - ref))) <= TOL * scale, (ntaps, M, n)


@pytest.mark.parametrize("ctaps", [False, True])
@pytest.mark.parametrize("L", [2, 4, 8, 16])
@pytest.mark.parametrize("ntaps", [1, 3, 16, 33, 255, 1000, 2049, 2500])
def test_fir_cf32_interpolating_replicated_spectrum(oracle, dev, ntaps, L, ctaps):
    """complex_float32, decimation 1, interpolation 2 / 4 / 8 / 16: a 4096/L-point forward transform, its spectrum
    replicated against H of the whole tap vector, the 4096-point inverse writing the interleaved output
    (fir_ols_decim.hip).  Same outputs and consume/produce counts as the reference's polyphase rows
    (FIRFilter.cpp:286-302, :341-350); tap vectors too long for the plan fall back to the polyphase kernel."""
    rng = np.random.default_rng(19 * ntaps + L + ctaps)
    taps = _taps(rng, ntaps, ctaps)
    K = -(-ntaps // L)
    for n in (K, K + 1, 4096 // L + K, 3 * (4096 // L) + 77 + K):
        x = rand_stream(rng, oracle.F32, n, True)
        ref_blk = oracle.Fir(oracle.F32, True, ctaps)
        ref_blk.set_taps(taps); ref_blk.set_interpolation(L); ref_blk.activate()
        ref, rc, rp, _ = ref_blk.work(x, n * L)
        f = dev.FirFilter((oracle.F32, True), "COMPLEX" if ctaps else "REAL")
        f.set_taps(taps); f.set_interpolation(L)
        got, gc, gp = f.process(x, n * L)
        assert (gc, gp) == (rc, rp), (ntaps, L, n)
        if rp:
            assert nerr(got, ref) <= TOL, (ntaps, L, n)
        # a short output buffer limits the iterations (FIRFilter.cpp:278): same counts again
        cap = (rp // 2) // L * L
        if cap:
            ref_blk2 = oracle.Fir(oracle.F32, True, ctaps)
            ref_blk2.set_taps(taps); ref_blk2.set_interpolation(L); ref_blk2.activate()
            ref2, rc2, rp2, _ = ref_blk2.work(x, cap)
            got2, gc2, gp2 = f.process(x, cap)
            assert (gc2, gp2) == (rc2, rp2) and nerr(got2, ref2) <= TOL, (ntaps, L, n, cap)


@pytest.mark.parametrize("cplx", [True, False])
@pytest.mark.parametrize("L,M", [(2, 1), (3, 1), (5, 3), (2, 3), (16, 1)])
@pytest.mark.parametrize("rowK", [2050, 4097, 5000, 8193])
def test_fir_f32_interpolating_with_long_rows(oracle, dev, rowK, L, M, cplx):
    """complex_float32 / float32, interpolation L (and decimation M) with polyphase rows of more than 2049 taps: every row through the
    partitioned kernel, then the interleaving pass.  Tap counts that leave the last rows one tap short; the reference's counts."""
    rng = np.random.default_rng(31 * rowK + 7 * L + M + cplx)
    ntaps = rowK * L - (L // 2)                                  # rows 0 .. L - L//2 - 1 have rowK taps, the others one fewer
    ctaps = cplx and bool(L % 2)
    taps = _taps(rng, ntaps, ctaps)
    for n in (rowK + M - 1, rowK + 4 * M + 2, rowK + 2048 * 2 + 77 * M):
        x = rand_stream(rng, oracle.F32, n, cplx)
        ref_blk = oracle.Fir(oracle.F32, cplx, ctaps)
        ref_blk.set_taps(taps); ref_blk.set_interpolation(L); ref_blk.set_decimation(M); ref_blk.activate()
        ref, rc, rp, _ = ref_blk.work(x, n * L)
        f = dev.FirFilter((oracle.F32, cplx), "COMPLEX" if ctaps else "REAL")
        f.set_taps(taps); f.set_interpolation(L); f.set_decimation(M)
        assert f.K == rowK
        got, gc, gp = f.process(x, n * L)
        assert (gc, gp) == (rc, rp), (rowK, L, M, n)
        if rp:
            assert f.last_algo == dev._lib.FIR_OLS_FFT
            typical = float(np.sqrt(np.sum(np.abs(taps) ** 2) / L) * np.sqrt(np.mean(x.astype(np.float64) ** 2) * (2 if cplx else 1)))
            assert float(np.abs(got - ref).max()) <= 2 * TOL * max(float(np.abs(ref).max()), 0.1 * typical), (rowK, L, M, n)


@pytest.mark.parametrize("cplx", [True, False])
@pytest.mark.parametrize("M", [2, 3, 8, 64, 1000])
@pytest.mark.parametrize("ntaps", [2050, 3000, 4097, 4098, 6145, 8193])
def test_fir_f32_long_decimating_filters(oracle, dev, ntaps, M, cplx):
    """complex_float32 / float32, decimation M, more than 2049 taps: the partitioned kernel at the full rate with a decimating store
    (fir_ols_part.hip; a real pair's halves split at a multiple of M).  Same kept outputs and consume / produce counts as the
    reference's loop (FIRFilter.cpp:286-302: the output with (n + 1) % M == 0); a short output buffer limits the iterations."""
    rng = np.random.default_rng(23 * ntaps + M + cplx)
    ctaps = cplx and bool(M % 2)
    taps = _taps(rng, ntaps, ctaps)
    for n in (ntaps + M - 1, ntaps + 5 * M + 3, ntaps + 2048 * 3 + 77, ntaps + 2048 * 11 + 5 * M + 1):
        x = rand_stream(rng, oracle.F32, n, cplx)
        ref_blk = oracle.Fir(oracle.F32, cplx, ctaps)
        ref_blk.set_taps(taps); ref_blk.set_decimation(M); ref_blk.activate()
        ref, rc, rp, _ = ref_blk.work(x, n)
        f = dev.FirFilter((oracle.F32, cplx), "COMPLEX" if ctaps else "REAL")
        f.set_taps(taps); f.set_decimation(M)
        got, gc, gp = f.process(x, n)
        assert (gc, gp) == (rc, rp), (ntaps, M, n)
        if rp:
            assert f.last_algo == dev._lib.FIR_OLS_FFT
            typical = float(np.sqrt(np.sum(np.abs(taps) ** 2)) * np.sqrt(np.mean(x.astype(np.float64) ** 2) * (2 if cplx else 1)))
            assert float(np.abs(got - ref).max()) <= 2 * TOL * max(float(np.abs(ref).max()), 0.1 * typical), (ntaps, M, n)
        cap = rp // 2
        if cap:
            ref_blk2 = oracle.Fir(oracle.F32, cplx, ctaps)
            ref_blk2.set_taps(taps); ref_blk2.set_decimation(M); ref_blk2.activate()
            ref2, rc2, rp2, _ = ref_blk2.work(x, cap)
            got2, gc2, gp2 = f.process(x, cap)
            assert (gc2, gp2) == (rc2, rp2), (ntaps, M, n, cap)
            assert float(np.abs(got2 - ref2).max()) <= 2 * TOL * max(float(np.abs(ref2).max()), 0.1 * typical), (ntaps, M, n, cap)


@pytest.mark.parametrize("scalar_name", ["float64", "int16", "int8"])
@pytest.mark.parametrize("M", [2, 3, 5, 8, 16, 100])
@pytest.mark.parametrize("ntaps", [2, 31, 32, 255, 2049, 4097])
def test_fir_complex_decimating_on_the_double_pipeline(oracle, dev, ntaps, M, scalar_name):
    """complex_float64 / complex_int16 / complex_int8 with decimation M (any M, interpolation 1): the double-precision
    overlap-save pipeline at full rate, one output in M stored -- the ones the reference's decimator keeps.  Integers
    bit-exact, float64 within 1e-13; consume/produce counts as the reference."""
    scalar = {"float64": oracle.F64, "int16": oracle.I16, "int8": oracle.I8}[scalar_name]
    rng = np.random.default_rng(23 * ntaps + M)
    taps = _taps(rng, ntaps, True) * 0.9
    full = {"float64": 1, "int16": 32768, "int8": 128}[scalar_name]
    npdt = {"float64": np.float64, "int16": np.int16, "int8": np.int8}[scalar_name]
    for n in (ntaps + M - 1, ntaps + 7 * M + 1, 2 * 4096 + 333 + ntaps):
        x = (rng.standard_normal((n, 2)) if scalar == oracle.F64 else rng.integers(-full, full, size=(n, 2))).astype(npdt)
        ref_blk = oracle.Fir(scalar, True, True)
        ref_blk.set_taps(taps); ref_blk.set_decimation(M); ref_blk.activate()
        ref, rc, rp, _ = ref_blk.work(x, n)
        f = dev.FirFilter((scalar, True), "COMPLEX")
        f.set_taps(taps); f.set_decimation(M)
        got, gc, gp = f.process(x, n)
        assert (gc, gp) == (rc, rp), (ntaps, M, n)
        if rp == 0:
            continue
        if scalar == oracle.F64:
            assert nerr(got, ref) <= 1e-13, (ntaps, M, n)
        else:
            assert np.array_equal(got, ref), (ntaps, M, n)
        lo = 16 if scalar == oracle.F64 else 32
        assert (f.last_algo == dev._lib.FIR_OLS_FFT) == (lo <= ntaps <= 4097), (ntaps, f.last_algo)


@pytest.mark.parametrize("scalar_name", ["float64", "float32", "int16", "int8"])
@pytest.mark.parametrize("M", [2, 3, 7, 16, 100])
@pytest.mark.parametrize("ntaps", [2, 15, 16, 255, 2049, 4097])
def test_fir_real_decimating_on_the_double_pipeline(oracle, dev, ntaps, M, scalar_name):   # (float32: the partitioned float kernel since late round 6)
    """REAL float64 / float32 / int16 / int8 streams with decimation M: two real blocks per double-precision transform at
    full rate, one output in M stored.  Integers bit-exact, float64 1e-13, float32 1e-5; counts as the reference."""
    scalar = {"float64": oracle.F64, "float32": oracle.F32, "int16": oracle.I16, "int8": oracle.I8}[scalar_name]
    rng = np.random.default_rng(29 * ntaps + M)
    taps = _taps(rng, ntaps, False) * 0.9
    full = {"float64": 1, "float32": 1, "int16": 32768, "int8": 128}[scalar_name]
    npdt = {"float64": np.float64, "float32": np.float32, "int16": np.int16, "int8": np.int8}[scalar_name]
    floats = scalar in (oracle.F64, oracle.F32)
    for n in (ntaps + M - 1, ntaps + 7 * M + 1, 3 * 4096 + 333 + ntaps):
        x = (rng.standard_normal(n) if floats else rng.integers(-full, full, size=n)).astype(npdt)
        ref_blk = oracle.Fir(scalar, False, False)
        ref_blk.set_taps(taps); ref_blk.set_decimation(M); ref_blk.activate()
        ref, rc, rp, _ = ref_blk.work(x, n)
        f = dev.FirFilter((scalar, False), "REAL")
        f.set_taps(taps); f.set_decimation(M)
        got, gc, gp = f.process(x, n)
        assert (gc, gp) == (rc, rp), (ntaps, M, n)
        if rp == 0:
            continue
        if floats:
            scale = max(float(np.max(np.abs(ref))), 0.1 * float(np.sum(np.abs(taps))) * float(np.max(np.abs(x))))
            assert float(np.max(np.abs(got.astype(np.float64) - ref))) <= (1e-13 if scalar == oracle.F64 else TOL) * scale, (ntaps, M, n)
        else:
            assert np.array_equal(got, ref), (ntaps, M, n)
        assert (f.last_algo == dev._lib.FIR_OLS_FFT) == (16 <= ntaps <= 4097), (ntaps, f.last_algo)


@pytest.mark.parametrize("scalar_name", ["float64", "int16", "int8"])
@pytest.mark.parametrize("L", [2, 3, 5, 8])
@pytest.mark.parametrize("M", [1, 2, 3, 7])
@pytest.mark.parametrize("ntaps", [7, 40, 161, 1000, 4000])
def test_fir_complex_interpolating_on_the_double_pipeline(oracle, dev, ntaps, L, M, scalar_name):
    """complex_float64 / complex_int16 / complex_int8 with interpolation L (and decimation M: rational resampling): every
    polyphase row through the double-precision overlap-save pipeline into a contiguous row, then the interleaving pass,
    which keeps one position in M.  Integers bit-exact, float64 1e-13; counts as the reference."""
    scalar = {"float64": oracle.F64, "int16": oracle.I16, "int8": oracle.I8}[scalar_name]
    rng = np.random.default_rng(31 * ntaps + L)
    taps = _taps(rng, ntaps, True) * 0.9
    full = {"float64": 1, "int16": 32768, "int8": 128}[scalar_name]
    npdt = {"float64": np.float64, "int16": np.int16, "int8": np.int8}[scalar_name]
    K = -(-ntaps // L)
    for n in (K, K + 3, 2 * 4096 + 99 + K):
        x = (rng.standard_normal((n, 2)) if scalar == oracle.F64 else rng.integers(-full, full, size=(n, 2))).astype(npdt)
        ref_blk = oracle.Fir(scalar, True, True)
        ref_blk.set_taps(taps); ref_blk.set_interpolation(L); ref_blk.set_decimation(M); ref_blk.activate()
        ref, rc, rp, _ = ref_blk.work(x, n * L)
        f = dev.FirFilter((scalar, True), "COMPLEX")
        f.set_taps(taps); f.set_interpolation(L); f.set_decimation(M)
        got, gc, gp = f.process(x, n * L)
        assert (gc, gp) == (rc, rp), (ntaps, L, M, n)
        if rp == 0:
            continue
        if scalar == oracle.F64:
            scale = max(float(np.max(np.abs(ref))), 0.1 * float(np.sqrt(np.sum(np.abs(taps) ** 2))))
            assert float(np.max(np.abs(got - ref))) <= 1e-13 * scale, (ntaps, L, M, n)
        else:
            assert np.array_equal(got, ref), (ntaps, L, M, n)
        assert (f.last_algo == dev._lib.FIR_OLS_FFT) == (16 <= K <= 2049), (ntaps, L, M, f.last_algo)


@pytest.mark.parametrize("scalar_name", ["float64", "float32", "int16", "int8"])
@pytest.mark.parametrize("L,M", [(2, 1), (3, 1), (8, 1), (3, 2), (5, 7)])
@pytest.mark.parametrize("ntaps", [7, 48, 161, 1000, 4000])
def test_fir_real_interpolating_on_the_double_pipeline(oracle, dev, ntaps, L, M, scalar_name):
    """REAL float64 / float32 / int16 / int8 streams with interpolation L (and decimation M): polyphase rows on the
    two-real-blocks-per-transform kernel, contiguous rows, interleaving pass.  Integers bit-exact, floats 1e-13 / 1e-5."""
    scalar = {"float64": oracle.F64, "float32": oracle.F32, "int16": oracle.I16, "int8": oracle.I8}[scalar_name]
    rng = np.random.default_rng(37 * ntaps + L + M)
    taps = _taps(rng, ntaps, False) * 0.9
    full = {"float64": 1, "float32": 1, "int16": 32768, "int8": 128}[scalar_name]
    npdt = {"float64": np.float64, "float32": np.float32, "int16": np.int16, "int8": np.int8}[scalar_name]
    floats = scalar in (oracle.F64, oracle.F32)
    K = -(-ntaps // L)
    for n in (K + M - 1, K + 3 * M, 2 * 4096 + 99 + K):
        x = (rng.standard_normal(n) if floats else rng.integers(-full, full, size=n)).astype(npdt)
        ref_blk = oracle.Fir(scalar, False, False)
        ref_blk.set_taps(taps); ref_blk.set_interpolation(L); ref_blk.set_decimation(M); ref_blk.activate()
        ref, rc, rp, _ = ref_blk.work(x, n * L)
        f = dev.FirFilter((scalar, False), "REAL")
        f.set_taps(taps); f.set_interpolation(L); f.set_decimation(M)
        got, gc, gp = f.process(x, n * L)
        assert (gc, gp) == (rc, rp), (ntaps, L, M, n)
        if rp == 0:
            continue
        if floats:
            scale = max(float(np.max(np.abs(ref))), 0.1 * float(np.sqrt(np.sum(taps ** 2))))
            assert float(np.max(np.abs(got.astype(np.float64) - ref))) <= (1e-13 if scalar == oracle.F64 else TOL) * scale, (ntaps, L, M, n)
        else:
            assert np.array_equal(got, ref), (ntaps, L, M, n)
        assert (f.last_algo == dev._lib.FIR_OLS_FFT) == (16 <= K <= 2049), (ntaps, L, M, f.last_algo)


# --------------------------------------------------------------------------- #
# FFT sizes that are not powers of two: kissfft's mixed-radix plan on the device
# --------------------------------------------------------------------------- #
MIXED_SIZES = [3, 5, 6, 7, 9, 10, 12, 15, 20, 25, 27, 30, 49, 60, 100, 121, 210, 360, 1000, 1009, 1920, 6000]


@pytest.mark.parametrize("inverse", [False, True])
@pytest.mark.parametrize("nbins", MIXED_SIZES)
def test_fft_mixed_radix_float(oracle, dev, nbins, inverse):
    rng = np.random.default_rng(nbins)
    x = rand_stream(rng, oracle.F32, nbins * 3, True)
    assert nerr(dev.Fft("complex_float32", nbins, inverse).transform(x), oracle.fft(x, nbins, inverse)) <= TOL
    if nbins <= 1000:
        xd = rand_stream(rng, oracle.F64, nbins * 2, True)
        assert nerr(dev.Fft("complex_float64", nbins, inverse).transform(xd), oracle.fft(xd, nbins, inverse)) <= 1e-13


@pytest.mark.parametrize("inverse", [False, True])
@pytest.mark.parametrize("dtype,nbins", [("complex_float32", 12000), ("complex_float32", 2 * 10223), ("complex_float32", 30030),
                                         ("complex_float32", 48000), ("complex_float32", 100000), ("complex_float32", 3 << 16),
                                         ("complex_float64", 6000), ("complex_float64", 2 * 5119), ("complex_float64", 48000)])
def test_fft_composite_beyond_one_workgroup(oracle, dev, dtype, nbins, inverse):
    """Sizes with odd factors too long for one workgroup's LDS: four-step N = n1 * n2 around two mixed-radix plans
    (kissfft takes any size, kissfft.hh:81-161).  2 x prime-beyond-the-LDS has no such split and is refused."""
    scalar = oracle.F32 if dtype.endswith("32") else oracle.F64
    rng = np.random.default_rng(nbins)
    x = rand_stream(rng, scalar, nbins * 2, True)
    got = dev.Fft(dtype, nbins, inverse).transform(x)
    assert nerr(got, oracle.fft(x, nbins, inverse)) <= (TOL if scalar == oracle.F32 else 1e-13)


@pytest.mark.parametrize("inverse", [False, True])
@pytest.mark.parametrize("nbins", [18, 45, 50, 75, 81, 90, 225, 243, 486, 625, 675, 720, 1125, 1200, 1536, 1875, 2025, 2400, 3125, 3645, 5625, 6144, 7680, 8100])
def test_fft_five_smooth_plan(oracle, dev, nbins, inverse):
    """complex_float32 (below 8192 bins) and complex_float64 (below 4096), 2^a 3^b 5^c bins: radix 16/8/4/2 passes, the pair passes 6, 15 (prime-factor) and 9
    (inner twiddles) and single 5s / 3s in every combination the planner produces; ragged frame groups"""
    rng = np.random.default_rng(nbins)
    nframes = 11 if nbins < 1000 else 3
    x = rand_stream(rng, oracle.F32, nbins * nframes, True)
    assert nerr(dev.Fft("complex_float32", nbins, inverse).transform(x), oracle.fft(x, nbins, inverse)) <= TOL
    if nbins < 4096:     # the same plan in double precision (256 ... 2047 bins; kissfft's order outside)
        xd = rand_stream(rng, oracle.F64, nbins * nframes, True)
        assert nerr(dev.Fft("complex_float64", nbins, inverse).transform(xd), oracle.fft(xd, nbins, inverse)) <= 1e-13


@pytest.mark.parametrize("inverse", [False, True])
@pytest.mark.parametrize("nbins", MIXED_SIZES)
def test_fft_mixed_radix_int16_bit_exact(oracle, dev, nbins, inverse):
    """radix 3 / 5 / generic butterflies of kiss_fft.c in Q15: every rounding reproduced"""
    rng = np.random.default_rng(nbins + 5)
    x = rand_stream(rng, oracle.I16, nbins * 3, True)
    assert np.array_equal(dev.Fft("complex_int16", nbins, inverse).transform(x), oracle.fft(x, nbins, inverse))


# --------------------------------------------------------------------------- #
# /comms/angle (first "next" sibling: shares getAngle with FreqDemod; math/TestAngle.cpp)
# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("scalar", SCALARS, ids=lambda s: NAMES[s])
def test_angle(oracle, dev, scalar):
    import os
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden.npz"))
    rng = np.random.default_rng(scalar + 40)
    x = rand_stream(rng, scalar, 30001, True)
    got, ref = dev.angle(x), oracle.angle(x)
    if scalar in (oracle.F64, oracle.F32):
        assert ang_err(got, ref) <= TOL
    else:
        assert np.array_equal(got, ref)
    # the reference test's own 13 points and the compiled reference's outputs on them
    zin = gold["angle_in_" + NAMES[scalar]]
    g = dev.angle(zin)
    if scalar in (oracle.F64, oracle.F32):
        assert np.max(np.abs(g - np.arctan2(zin[:, 1].astype(np.float64), zin[:, 0]))) <= np.pi / 500   # TestAngle.cpp:57
        assert ang_err(g, gold["angle_ref_" + NAMES[scalar]]) <= TOL
    else:
        assert np.array_equal(g, gold["angle_ref_" + NAMES[scalar]])


@pytest.mark.parametrize("ctaps", [True, False])
@pytest.mark.parametrize("L,M,ntaps", [(1, 2, 255), (1, 4, 64), (1, 10, 1000), (3, 1, 100), (3, 2, 61), (2, 3, 255),
                                       (5, 7, 333), (4, 4, 16), (1, 4096, 31), (8, 1, 2049 * 8)])
def test_fir_cf32_frequency_domain_resampling(oracle, dev, L, M, ntaps, ctaps):
    """interpolation / decimation on the overlap-save kernel (one launch per polyphase row) vs the oracle"""
    rng = np.random.default_rng(L * 100 + M + ntaps)
    n = 3 * 4096 + 1234 + (-(-ntaps // L))
    x = rand_stream(rng, oracle.F32, n, True)
    taps = _taps(rng, ntaps, ctaps)
    ref_blk = oracle.Fir(oracle.F32, True, ctaps)
    f = dev.FirFilter("complex_float32", "COMPLEX" if ctaps else "REAL")
    for b in (ref_blk, f):
        b.set_taps(taps); b.set_interpolation(L); b.set_decimation(M)
    ref_blk.activate()
    for cap in (n * L, 1000):
        ref, rc, rp, _ = ref_blk.work(x, cap)
        got, gc, gp = f.process(x, cap)
        assert f.last_algo == dev._lib.FIR_OLS_FFT
        assert (gc, gp) == (rc, rp)
        if rp:
            assert nerr(got, ref) <= TOL


@pytest.mark.parametrize("ntaps", [2, 31, 255, 1024, 2049, 2050, 3000, 4097, 4098, 6145, 8193])
def test_fir_real_f32_frequency_domain(oracle, dev, ntaps):
    """real float32 stream with REAL taps: two real blocks per complex transform; beyond 2049 taps the call's two halves side by side
    as one complex stream through the partitioned kernel (odd and even lengths, a second half shorter than the first, one block)"""
    rng = np.random.default_rng(ntaps + 77)
    for n in (ntaps + 5, 3842 * 2 + 100 + ntaps, 3842 * 5 + 7 + ntaps) + ((ntaps + 2048 * 9 + 31, ntaps + 70001) if ntaps > 2049 else ()):
        x = rand_stream(rng, oracle.F32, n, False)
        taps = _taps(rng, ntaps, False)
        ref_blk = oracle.Fir(oracle.F32, False, False); ref_blk.set_taps(taps); ref_blk.activate()
        ref, rc, rp, _ = ref_blk.work(x, n)
        f = dev.FirFilter("float32", "REAL"); f.set_taps(taps)
        got, gc, gp = f.process(x, n)
        assert f.last_algo == dev._lib.FIR_OLS_FFT
        assert (gc, gp) == (rc, rp) == (n - ntaps + 1, n - ntaps + 1)
        assert nerr(got, ref) <= TOL


def test_fir_host_path_large_buffer(oracle, dev):
    """a multi-megasample host buffer through the staging path (seams of the device blocks checked)"""
    rng = np.random.default_rng(123)
    ntaps, n = 255, 7 * (1 << 20) + 12345
    x = rand_stream(rng, oracle.F32, n + ntaps - 1, True)
    taps = _taps(rng, ntaps, True)
    f = dev.FirFilter("complex_float32", "COMPLEX"); f.set_taps(taps)
    got, c, p = f.process(x, n)
    assert (c, p) == (n, n)
    ref = oracle.Fir(oracle.F32, True, True); ref.set_taps(taps); ref.activate()
    for start in (0, (1 << 21) - 300, 3 * (1 << 21) - 300, n - 5000):     # chunk seams and both ends
        want, _, pp, _ = ref.work(x[start:start + 5000 + ntaps - 1], 5000)
        m = min(5000, n - start)
        assert nerr(got[start:start + m], want[:m]) <= TOL, start


@pytest.mark.parametrize("seed", range(int(os.environ.get("PCX_FUZZ_SEEDS", "24"))))
def test_fir_randomised_configurations(oracle, dev, seed):
    """random (type, taps kind, K, L, M, buffer sizes, output room) through AUTO vs the oracle's work();
    PCX_FUZZ_SEEDS=N widens the sweep (soak runs)"""
    rng = np.random.default_rng(1000 + seed)
    scalar = [oracle.F32, oracle.F32, oracle.F32, oracle.F64, oracle.I16, oracle.I8, oracle.I32, oracle.I64][seed % 8 if seed >= 24 else seed % 7]
    is_complex = bool(rng.integers(0, 4) > 0) or scalar != oracle.F32
    ctaps = is_complex and bool(rng.integers(0, 2))
    L, M = int(rng.integers(1, 5)), int(rng.integers(1, 6))
    if seed % 3 == 0:
        L = M = 1
    ntaps = int(rng.integers(1, 700 if scalar == oracle.F32 else 80))
    if seed >= 24 and seed % 11 == 0 and scalar == oracle.F32:
        ntaps = int(rng.integers(2000, 6000))      # the 8192 / 16384-sample overlap-save plans (M = L = 1) and their fallbacks
    if seed >= 24 and seed % 7 == 3 and scalar == oracle.F32:
        L, M = int(rng.integers(1, 13)), int(rng.integers(1, 41))   # many polyphase rows / sparse decimation
        ntaps = int(rng.integers(1, 4000))
    K = -(-ntaps // L)
    n_in = int(rng.integers(K, K + 30000))
    out_cap = int(rng.integers(1, 2 * n_in * L // M + 10))
    taps = _taps(rng, ntaps, ctaps) * (1.0 if scalar in (oracle.F32, oracle.F64) else 0.9)
    x = rand_stream(rng, scalar, n_in, is_complex, amp=1000 if scalar not in (oracle.I8,) else 100)
    ref = oracle.Fir(scalar, is_complex, ctaps)
    f = dev.FirFilter((scalar, is_complex), "COMPLEX" if ctaps else "REAL")
    for b in (ref, f):
        b.set_taps(taps); b.set_interpolation(L); b.set_decimation(M)
    ref.activate()
    want, rc, rp, _ = ref.work(x, out_cap)
    got, gc, gp = f.process(x, out_cap)
    assert (gc, gp) == (rc, rp), (scalar, is_complex, ctaps, L, M, ntaps, n_in, out_cap)
    if rp == 0:
        return
    if scalar in (oracle.F32, oracle.F64):
        assert nerr(got, want) <= (TOL if scalar == oracle.F32 else 1e-12)
    else:
        assert np.array_equal(got, want)


@pytest.mark.parametrize("big", [0, 1, 2])
def test_fir_int16_tap_magnitude_paths(oracle, dev, big):
    """int16 streams: complex Q16.16 taps below 0.5 in magnitude take the packed dot-product kernel, taps below
    128 the 24-bit multiply path, larger ones the 32-bit one; all wrap exactly like the oracle's ring arithmetic."""
    rng = np.random.default_rng(77 + big)
    n, K = 50001, 37
    x = rand_stream(rng, oracle.I16, n, True)
    taps = (rng.standard_normal(K) + 1j * rng.standard_normal(K)) * [0.4, 300.0, 0.1][big]
    if big == 2:
        taps = np.clip(taps.real, -0.49, 0.49) + 1j * np.clip(taps.imag, -0.49, 0.49)
        taps[3] = 0.4999 - 0.4999j         # extremes of the 16-bit range
    ref = oracle.Fir(oracle.I16, True, True); ref.set_taps(taps); ref.activate()
    want, rc, rp, _ = ref.work(x, n)
    f = dev.FirFilter("complex_int16", "COMPLEX"); f.set_taps(taps)
    got, gc, gp = f.process(x, n)
    assert (gc, gp) == (rc, rp)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("seed", range(int(os.environ.get("PCX_FUZZ_SEEDS", "24"))))
def test_fft_randomised_sizes(oracle, dev, seed):
    """random numBins (any factorisation, primes included), type, direction and frame count vs kissfft's restatement"""
    rng = np.random.default_rng(5000 + seed)
    scalar = [oracle.F32, oracle.F64, oracle.I16][seed % 3]
    nbins = int(rng.integers(1, 2500)) if seed % 4 else int(2 ** rng.integers(0, 13))
    inverse = bool(rng.integers(0, 2))
    nframes = int(rng.integers(1, 6))
    x = rand_stream(rng, scalar, nbins * nframes, True)
    ref = oracle.fft(x, nbins, inverse)
    got = dev.Fft((scalar, True), nbins, inverse).transform(x)
    if scalar == oracle.I16:
        assert np.array_equal(got, ref), nbins
    else:
        assert nerr(got, ref) <= (TOL if scalar == oracle.F32 else 1e-12), nbins


def test_fft_int16_one_bin_is_not_the_identity(oracle, dev):
    """kiss_fft Q15 with numBins = 1 runs kf_bfly_generic(p = 1): C_FIXDIV by 1 = x * 32767/32768 rounded
    (found by the randomised sweep; checked against the compiled reference on the CPU side)"""
    x = np.array([[5836, -18035], [123, -1], [-32768, 32767], [1, 0]], np.int16)
    for inverse in (False, True):
        want = oracle.fft(x, 1, inverse)
        assert want.tolist() == [[5836, -18034], [123, -1], [-32767, 32766], [1, 0]]
        assert np.array_equal(dev.Fft("complex_int16", 1, inverse).transform(x), want)
    xf = np.array([[1.5, -2.25]], np.float32)
    assert np.array_equal(dev.Fft("complex_float32", 1, False).transform(xf), xf)     # float: leaf copy


def test_documented_in_place_calls(oracle, dev):
    """include/pcx.h: same-size maps and the FFT accept out == in on device buffers"""
    import torch
    d = torch.device("cuda", 0)
    rng = np.random.default_rng(12)
    n = 70001
    x = rand_stream(rng, oracle.F32, n, True)
    t = torch.from_numpy(x).to(d)
    dev.rotate(t, 0.9, scalar=dev.F32, out=t, n=n)
    assert np.array_equal(t.cpu().numpy(), oracle.rotate(x, 0.9))
    t = torch.from_numpy(x).to(d)
    dev.conj(t, scalar=dev.F32, out=t, n=n)
    assert np.array_equal(t.cpu().numpy(), oracle.conj(x))
    for nbins, nframes in ((4096, 9), (256, 40), (1000, 5), (32768, 2)):
        xf = rand_stream(rng, oracle.F32, nbins * nframes, True)
        tf = torch.from_numpy(xf).to(d)
        dev.Fft("complex_float32", nbins, False).transform_dev(tf, tf, nframes)
        assert nerr(tf.cpu().numpy(), oracle.fft(xf, nbins, False)) <= TOL, nbins
    xi = rand_stream(rng, oracle.I16, 1024 * 7, True)
    ti = torch.from_numpy(xi).to(d)
    dev.Fft("complex_int16", 1024, False).transform_dev(ti, ti, 7)
    assert np.array_equal(ti.cpu().numpy(), oracle.fft(xi, 1024, False))


@pytest.mark.parametrize("nbins", [65536, 131072, 98304, 40000, 57344, 2 * 32771])
@pytest.mark.parametrize("inv", [False, True])
def test_q15_fft_beyond_one_workgroup_is_bit_exact(oracle, dev, nbins, inv):
    """complex_int16 frames that no workgroup's LDS holds (the last PCX_ERR_UNSUPPORTED): kf_work's stages one launch
    each over global memory (kiss_fft.c:237-302), every C_FIXDIV where the reference has it.  Powers of two (65,536 = 4^8,
    131,072 = 4^8 * 2), 3 * 2^15, 2^6 * 5^4, 7 * 2^13 (a generic-radix stage) and 2 * a prime beyond 2^15 (kf_bfly_generic over
    32,771 terms): bit for bit against the oracle and, where the compiled reference is present, against fft/kiss_fft.c itself."""
    rng = np.random.default_rng(nbins + int(inv))
    nframes = 2 if nbins < 100000 and nbins != 2 * 32771 else 1
    x = rng.integers(-32768, 32768, (nframes * nbins, 2)).astype(np.int16)
    got = dev.Fft("complex_int16", nbins, inv).transform(x)
    want = oracle.fft(x, nbins, inv)
    assert np.array_equal(got, want)
    if oracle.ref() is not None:
        assert np.array_equal(got, oracle.ref_fft(x, nbins, inv))


def test_q15_fft_beyond_one_workgroup_in_place_and_batched(oracle, dev):
    """the same plan with input and output in ONE device buffer (the gather needs the workspace as its destination)"""
    import torch
    nbins, nframes = 65536, 3
    rng = np.random.default_rng(5)
    x = rng.integers(-20000, 20000, (nframes * nbins, 2)).astype(np.int16)
    t = torch.from_numpy(x).cuda()
    dev.Fft("complex_int16", nbins, False).transform_dev(t, t, nframes)
    torch.cuda.synchronize()
    assert np.array_equal(t.cpu().numpy(), oracle.fft(x, nbins, False))
