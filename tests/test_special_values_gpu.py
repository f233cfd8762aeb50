"""GPU suite on IEEE special values: signed zeros, subnormals, the largest finite numbers, infinities and NaN through every
float element-wise path, FreqDemod and the reference-order FIR.  The reference runs libm (`hypotf`, `atan2f`) and IEEE
arithmetic on whatever it is fed; the device must give the same bits for everything that is not NaN and NaN exactly where
the reference gives NaN (payloads are not compared: x86 and the GPU quiet and propagate them differently)."""
import numpy as np
import pytest

from tests.util import TOL

pytestmark = pytest.mark.gpu


def _grid(dt):
    fi = np.finfo(dt)
    v = np.array([0.0, -0.0, fi.smallest_subnormal, -fi.smallest_subnormal * 3, fi.tiny, -fi.tiny, 1.0, -1.0, 0.37, -2.5e3,
                  fi.max, -fi.max, fi.max / 3, np.inf, -np.inf, np.nan], dtype=dt)
    re, im = np.meshgrid(v, v)
    return np.stack([re.ravel(), im.ravel()], 1).astype(dt)


def _same(got, want, what):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape and got.dtype == want.dtype, what
    gn, wn = np.isnan(got), np.isnan(want)
    assert np.array_equal(gn, wn), (what, "NaN where the reference has none (or the reverse)", np.flatnonzero(gn != wn)[:8])
    u = np.uint32 if got.dtype == np.float32 else np.uint64
    ok = gn | (got.view(u) == want.view(u))
    assert ok.all(), (what, np.flatnonzero(~ok)[:8], got[~ok][:4], want[~ok][:4])


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_bit_exact_maps_on_special_values(oracle, dev, dt):
    x = _grid(dt)
    _same(dev.conj(x), oracle.conj(x), "conj")
    for f in (0.37, -2.0, 0.0, 1e30):
        _same(dev.scale(x, f, True), oracle.scale(x, f, True), ("scale", f))
    for ph in (0.0, 0.7, np.pi / 2, -3.0):
        _same(dev.rotate(x, ph), oracle.rotate(x, ph), ("rotate", ph))
    y = np.roll(x, 37, axis=0)
    for op in ("ADD", "SUB", "MUL"):
        _same(dev.arith(op, x, y, True), oracle.arith(getattr(oracle, op), x, y, True), op)
    # DIV: libgcc's __divsc3 (computed in double, rounded once) / __divdc3 (the rescaling algorithm libgcc has shipped since GCC 12),
    # both with their Annex G slow paths: bit for bit, on the grid and on operands spread over the whole exponent range
    _same(dev.arith("DIV", x, y, True), oracle.arith(oracle.DIV, x, y, True), "DIV on the grid")
    rng = np.random.default_rng(12)
    fi = np.finfo(dt)
    n = 100000
    e = rng.integers(fi.minexp - 50, fi.maxexp, size=(n, 4))
    w = np.ldexp(rng.uniform(1, 2, size=(n, 4)) * rng.choice([-1.0, 1.0], size=(n, 4)), e).astype(dt)
    sp = np.array([0.0, -0.0, fi.smallest_subnormal, fi.tiny, -fi.tiny, 1.0, fi.max, -fi.max, fi.max / 2, fi.eps, 1 / fi.eps, np.inf, -np.inf, np.nan], dtype=dt)
    w = np.where(rng.random((n, 4)) < 0.25, sp[rng.integers(0, len(sp), size=(n, 4))], w).astype(dt)
    p, q = np.ascontiguousarray(w[:, :2]), np.ascontiguousarray(w[:, 2:])
    _same(dev.arith("DIV", p, q, True), oracle.arith(oracle.DIV, p, q, True), "DIV over the exponent range")
    _same(dev.arith("MUL", p, q, True), oracle.arith(oracle.MUL, p, q, True), "MUL over the exponent range")
    re, im = dev.split_complex(x)
    _same(re, np.ascontiguousarray(x[:, 0]), "split re"); _same(im, np.ascontiguousarray(x[:, 1]), "split im")
    _same(dev.combine_complex(re, im), x, "combine")


def test_abs_is_hypotf_on_special_values(oracle, dev):
    """std::abs(complex<float>) is hypotf: +inf whenever a part is infinite, even next to a NaN; no overflow for parts near FLT_MAX / sqrt 2;
    subnormal results exact"""
    x = _grid(np.float32)
    _same(dev.abs_(x, True), oracle.abs_(x, True), "abs complex_float32")
    r = np.ascontiguousarray(x[:, 0])
    _same(dev.abs_(r, False), oracle.abs_(r, False), "abs float32")


def test_angle_and_freq_demod_on_special_values(oracle, dev):
    """atan2f's special cases: signed zeros choose 0 / +-pi, infinities give multiples of pi/4, NaN propagates"""
    x = _grid(np.float32)
    want, got = oracle.angle(x), dev.angle(x)
    gn, wn = np.isnan(got), np.isnan(want)
    assert np.array_equal(gn, wn), np.flatnonzero(gn != wn)[:8]
    d = np.abs(got[~wn].astype(np.float64) - want[~wn].astype(np.float64))
    assert np.all(d <= TOL * np.pi), (x[~wn][d > TOL * np.pi][:6], got[~wn][d > TOL * np.pi][:6], want[~wn][d > TOL * np.pi][:6])
    zero_in = ~wn & (want == 0)
    assert np.array_equal(np.signbit(got[zero_in]), np.signbit(want[zero_in]))       # +0 and -0 as the reference
    # FreqDemod: in[i] * conj(in[i-1]) through the reference's complex multiply (slow path included), then atan2f
    fin = np.concatenate([x, x[::-1], np.roll(x, 5, axis=0)])
    want, got = oracle.FreqDemod(oracle.F32).work(fin), dev.FreqDemod("complex_float32").process(fin)
    gn, wn = np.isnan(got), np.isnan(want)
    assert np.array_equal(gn, wn)
    d = np.abs((got[~wn].astype(np.float64) - want[~wn] + np.pi) % (2 * np.pi) - np.pi)
    assert np.all(d <= TOL * np.pi), np.flatnonzero(d > TOL * np.pi)[:8]


def test_reference_order_fir_on_special_values(oracle, dev):
    """PCX_FIR_EXACT keeps the reference's operations and their order: same bits, same NaNs, subnormal partial sums included"""
    from pothoscomms_amd import _lib
    x = np.concatenate([_grid(np.float32)] * 3)
    rng = np.random.default_rng(4)
    for taps in (np.array([1.0]), np.array([0.5, -0.25 + 0.5j, 1e-30, 3.0]), rng.normal(size=17) + 1j * rng.normal(size=17)):
        ct = np.iscomplexobj(taps)
        f = dev.FirFilter("complex_float32", "COMPLEX" if ct else "REAL"); f.set_taps(taps); f.set_algo(_lib.FIR_EXACT)
        ref = oracle.Fir(oracle.F32, True, ct); ref.set_taps(taps); ref.activate()
        n = len(x) - len(taps) + 1
        want, _, p, _ = ref.work(x, n)
        got, _, gp = f.process(x, n)
        assert gp == p == n
        _same(got, want, ("fir exact", len(taps)))
    # the resampling kernel's reference-order mode (one output per lane) has the same second look
    taps = rng.normal(size=23) + 1j * rng.normal(size=23)
    for L, M in ((2, 1), (1, 3), (3, 2)):
        f = dev.FirFilter("complex_float32", "COMPLEX"); f.set_taps(taps); f.set_interpolation(L); f.set_decimation(M); f.set_algo(_lib.FIR_EXACT)
        ref = oracle.Fir(oracle.F32, True, True); ref.set_taps(taps); ref.set_interpolation(L); ref.set_decimation(M); ref.activate()
        cap = len(x) * L // M + 8
        want, rc, rp, _ = ref.work(x, cap)
        got, gc, gp = f.process(x, cap)
        assert (gc, gp) == (rc, rp) and rp > 0
        _same(got, want, ("fir exact resampling", L, M))


def test_double_precision_abs_and_angle_on_special_values(oracle, dev):
    """hypot and atan2 in double: the device's are not bit-identical to glibc's (bars 4e-15 / 1e-5 of pi), but every special case must
    be the reference's -- +inf next to a NaN, NaN propagation, 0 / +-pi on signed zeros, multiples of pi/4 on infinities"""
    x = _grid(np.float64)
    for name, got, want, tol in (("abs", dev.abs_(x, True), oracle.abs_(x, True), 4e-15), ("angle", dev.angle(x), oracle.angle(x), None)):
        gn, wn = np.isnan(got), np.isnan(want)
        assert np.array_equal(gn, wn), (name, np.flatnonzero(gn != wn)[:8])
        inf = ~wn & np.isinf(want)
        assert np.array_equal(got[inf], want[inf]), name
        zero = ~wn & (want == 0)
        assert np.array_equal(got[zero], want[zero]) and np.array_equal(np.signbit(got[zero]), np.signbit(want[zero])), name
        fin = ~wn & ~inf & ~zero
        if tol is not None:
            # subnormal magnitudes: a few ulps of the smallest subnormal are allowed on top of the relative bar
            assert np.all(np.abs(got[fin] - want[fin]) <= tol * np.abs(want[fin]) + 4 * np.finfo(np.float64).smallest_subnormal), name
        else:
            assert np.all(np.abs(got[fin] - want[fin]) <= TOL * np.pi), name
