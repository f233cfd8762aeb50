"""Randomised GPU parity sweeps (PCX_FUZZ_SEEDS sizes them; the default keeps the suite short):
element-wise maps on device buffers at arbitrary offsets, FreqDemod and the fused FM chain fed in
random work()-sized pieces with the state carried between calls, all against the oracle."""
import os

import numpy as np
import pytest

from tests.util import TOL, ang_err, d2h, diff_note, h2d, nerr, rand_stream

pytestmark = pytest.mark.gpu
SEEDS = range(int(os.environ.get("PCX_FUZZ_SEEDS", "12")))


@pytest.mark.parametrize("seed", SEEDS)
def test_maps_on_offset_device_buffers(oracle, dev, seed):
    """rotate / scale / conj / abs / angle / arithmetic through the *_dev entry points with the buffers
    starting at arbitrary element offsets (the 16-byte vector path must fall back cleanly) and odd lengths"""
    import torch
    d = torch.device("cuda", 0)
    rng = np.random.default_rng(9000 + seed)
    scalar = [oracle.F32, oracle.F64, oracle.I16, oracle.I8, oracle.I32, oracle.I64][seed % 6]
    TD = {oracle.F32: torch.float32, oracle.F64: torch.float64, oracle.I16: torch.int16, oracle.I8: torch.int8,
          oracle.I32: torch.int32, oracle.I64: torch.int64}[scalar]
    n = int(rng.integers(1, 70000))
    oi, oo = int(rng.integers(0, 5)), int(rng.integers(0, 5))
    x = rand_stream(rng, scalar, n, True)
    xin = torch.zeros((n + 8, 2), dtype=TD, device=d)
    xin[oi:oi + n] = h2d(x, d)
    xv = xin[oi:oi + n]
    outc = torch.zeros((n + 8, 2), dtype=TD, device=d)
    outr = torch.zeros((n + 8,), dtype=TD, device=d)
    oc, orr = outc[oo:oo + n], outr[oo:oo + n]
    phase = float(rng.uniform(-3, 3))
    dev.rotate(xv, phase, scalar=scalar, out=oc, n=n)
    assert np.array_equal(d2h(oc), oracle.rotate(x, phase))
    dev.scale(xv, 0.37, True, scalar=scalar, out=oc, n=n)
    assert np.array_equal(d2h(oc), oracle.scale(x, 0.37, True))
    dev.conj(xv, scalar=scalar, out=oc, n=n)
    assert np.array_equal(d2h(oc), oracle.conj(x))
    dev.abs_(xv, True, scalar=scalar, out=orr, n=n)
    ref = oracle.abs_(x, True)
    got = d2h(orr)
    assert (nerr(got, ref) <= 1e-15 * 4) if scalar == oracle.F64 else np.array_equal(got, ref)
    dev.angle(xv, scalar=scalar, out=orr, n=n)
    ref, got = oracle.angle(x), d2h(orr)
    assert (ang_err(got, ref) <= TOL) if scalar in (oracle.F32, oracle.F64) else np.array_equal(got, ref)
    y = rand_stream(rng, scalar, n, True)
    yin = torch.zeros((n + 8, 2), dtype=TD, device=d)
    yin[oo:oo + n] = h2d(y, d)
    for op in ("ADD", "SUB", "MUL"):
        dev.arith(op, xv, yin[oo:oo + n], True, scalar=scalar, out=oc, n=n)
        assert np.array_equal(d2h(oc).view(np.uint8), oracle.arith(getattr(oracle, op), x, y, True).view(np.uint8)), op
    # nothing outside the output windows was written
    assert not bool(outc[:oo].any()) and not bool(outc[oo + n:].any())
    assert not bool(outr[:oo].any()) and not bool(outr[oo + n:].any())


@pytest.mark.parametrize("seed", SEEDS)
def test_freqdemod_random_chunks(oracle, dev, seed):
    rng = np.random.default_rng(7000 + seed)
    scalar = [oracle.F32, oracle.I16, oracle.F64, oracle.I8, oracle.I32][seed % 5]
    n = int(rng.integers(1, 60000))
    if scalar in (oracle.F32, oracle.F64):
        ph = np.cumsum(rng.uniform(-1.5, 1.5, n))
        x = (np.stack([np.cos(ph), np.sin(ph)], 1) * rng.uniform(0.5, 1.5, (n, 1))).astype(oracle.NP_SCALAR[scalar])
    else:
        x = rand_stream(rng, scalar, n, True)
    ref_blk, gpu_blk = oracle.FreqDemod(scalar), dev.FreqDemod((scalar, True))
    cuts = sorted(set([0, n] + [int(c) for c in rng.integers(0, n + 1, int(rng.integers(0, 8)))]))
    for a, b in zip(cuts[:-1], cuts[1:]):
        ref, got = ref_blk.work(x[a:b]), gpu_blk.process(x[a:b])
        if scalar in (oracle.F32, oracle.F64):
            assert ang_err(got, ref) <= TOL, (a, b)
        else:
            assert np.array_equal(got, ref), (a, b)


@pytest.mark.parametrize("seed", SEEDS)
def test_fm_chain_random_chunks(oracle, dev, seed):
    """the fused kernel fed in random pieces (state = conj of the last FIR output carried on the device)
    equals Rotate -> FIR -> FreqDemod of the oracle over the whole stream"""
    from pothoscomms_amd import taps as tp
    rng = np.random.default_rng(8000 + seed)
    ntaps = int(rng.integers(1, 400))
    if seed % 9 == 4:
        ntaps = int(rng.integers(1500, 9000))      # both fused plans' limits and the unfused long-filter path
    ctaps = bool(rng.integers(0, 2))
    n = int(rng.integers(ntaps + 1, ntaps + 40000))
    x = tp.fm_test_signal(n, seed=seed)
    taps = tp.complex_bandpass(ntaps, 0.1, 0.03) if ctaps else tp.lowpass(ntaps, 0.1)
    if ntaps < 3:
        taps = np.ones(ntaps) * (1 + 0.5j if ctaps else 1.0)
    phase = float(rng.uniform(-3, 3))
    xr = oracle.rotate(x, phase)
    fir = oracle.Fir(oracle.F32, True, ctaps); fir.set_taps(taps); fir.activate()
    y, _, produced, _ = fir.work(xr, n)
    ref = oracle.FreqDemod(oracle.F32).work(y)
    ch = dev.FmChain(); ch.set_phase(phase); ch.set_taps(taps, ctaps)
    algo = [dev._lib.FIR_AUTO, dev._lib.FIR_OLS_FFT, dev._lib.FIR_DIRECT][seed % 3]
    if ntaps > 2048:
        algo = dev._lib.FIR_AUTO
    ch.set_algo(algo)
    xp = x.view(np.float32).reshape(-1, 2) if np.iscomplexobj(x) else x
    pos, outs = 0, []
    ends = sorted(set([n] + [int(c) for c in rng.integers(ntaps, n + 1, int(rng.integers(0, 5)))]))
    for end in ends:
        g, c, p = ch.process(xp[pos:end], n)
        outs.append(g); pos += c
    got = np.concatenate(outs)
    assert len(got) == produced == n - ntaps + 1
    assert ang_err(got, ref) <= TOL, (seed, ntaps, ends, diff_note(np.round(got, 3), np.round(ref, 3)))


@pytest.mark.parametrize("seed", SEEDS)
def test_fir_block_random_streaming_with_bursts(oracle, seed):
    """/comms/fir_filter block driven work() by work() with random window growth, output room, resampling
    and frame-start/end labels: consumed / produced / reserve and the samples follow the oracle's work()"""
    from pothoscomms_amd import blocks as B
    rng = np.random.default_rng(6000 + seed)
    scalar = [oracle.F32, oracle.I16, oracle.F64][seed % 3]
    name = {oracle.F32: "float32", oracle.I16: "int16", oracle.F64: "float64"}[scalar]
    ctaps = bool(rng.integers(0, 2))
    ntaps = int(rng.integers(1, 120))
    L, M = (1, 1) if seed % 2 else (int(rng.integers(1, 4)), int(rng.integers(1, 4)))
    taps = (rng.normal(size=ntaps) + (1j * rng.normal(size=ntaps) if ctaps else 0)) / max(1.0, np.sqrt(ntaps))
    n = int(rng.integers(2000, 30000))
    x = rand_stream(rng, scalar, n, True, amp=1000)
    blk = B.make("/comms/fir_filter", "complex_" + name, "COMPLEX" if ctaps else "REAL")
    ref = oracle.Fir(scalar, True, ctaps)
    blk.call("setTaps", taps); ref.set_taps(taps)
    blk.call("setInterpolation", L); ref.set_interpolation(L)
    blk.call("setDecimation", M); ref.set_decimation(M)
    mode = seed % 4          # 0/1: plain stream, 2: frame start only, 3: start and end labels
    if mode == 2:
        blk.call("setFrameStartId", "S"); ref.set_frame_ids(True, False)
    elif mode == 3:
        blk.call("setFrameStartId", "S"); blk.call("setFrameEndId", "E"); ref.set_frame_ids(True, True)
    blk.activate(); ref.activate()
    labels_at = []
    if mode >= 2:
        pos = int(rng.integers(0, n // 4))
        while pos < n - 10:
            blen = int(rng.integers(50, 3000))
            labels_at.append(("S", pos, blen if mode == 2 else None))
            if mode == 3:
                labels_at.append(("E", min(n - 1, pos + blen - 1), None))
            pos += blen + int(rng.integers(0, 2000))
    out_cap = int(rng.integers(64, 20000))
    pos, avail_end, guard, total_p, calls = 0, 0, 0, 0, 0
    while pos < n and guard < 400:
        guard += 1
        avail_end = min(n, avail_end + int(rng.integers(1, 6000)))
        win = x[pos:avail_end]
        labs = [B.Label(i, idx - pos, d) for (i, idx, d) in labels_at if pos <= idx < avail_end]
        rlabs = [(i, idx - pos, 1, d) for (i, idx, d) in labels_at if pos <= idx < avail_end]
        y, c, p, r, _ = blk.work(win, out_cap, labs)
        ry, rc, rp, rr = ref.work(win, out_cap, rlabs)
        assert (c, p, r) == (rc, rp, rr), (seed, pos, avail_end)
        if p:
            if scalar == oracle.I16:
                assert np.array_equal(y, ry)
            else:
                assert nerr(y, ry) <= (TOL if scalar == oracle.F32 else 1e-12) or float(np.abs(ry).max()) == 0.0, \
                    (seed, calls, diff_note(np.round(y, 2), np.round(ry, 2)), y.ctypes.data % 4096, y.itemsize)
        pos += c
        total_p += p
        calls += 1
        if c == 0 and avail_end == n:
            break
    assert calls >= 1
    if mode < 2:
        assert total_p > 0
        if guard < 400:      # not cut short by the call budget (tiny output room): a plain stream drains to its history
            assert pos >= n - (-(-ntaps // L)) - M


@pytest.mark.parametrize("seed", SEEDS)
def test_label_driven_rotate_scale_blocks(oracle, seed):
    """Rotate / Scale blocks with setLabelId: the coefficient changes at every matching label inside the
    buffer, work() stops in front of the next one (Rotate.cpp:105-123, Scale.cpp:104-122)"""
    from pothoscomms_amd import blocks as B
    rng = np.random.default_rng(4000 + seed)
    scalar = [oracle.F32, oracle.I16, oracle.F64, oracle.I32][seed % 4]
    name = {oracle.F32: "float32", oracle.I16: "int16", oracle.F64: "float64", oracle.I32: "int32"}[scalar]
    rotate = bool(seed % 2)
    n = int(rng.integers(10, 20000))
    x = rand_stream(rng, scalar, n, True, amp=1000)
    blk = B.make("/comms/rotate" if rotate else "/comms/scale", "complex_" + name)
    value = float(rng.uniform(-3, 3))
    blk.call("setPhase" if rotate else "setFactor", value)
    blk.call("setLabelId", "coef")
    marks = sorted(set(int(v) for v in rng.integers(0, n, int(rng.integers(0, 6)))))
    vals = [float(rng.uniform(-3, 3)) for _ in marks]
    others = [B.Label("other", int(v), 1.0) for v in rng.integers(0, n, 2)]
    pos, out = 0, []
    cur = value
    want = np.zeros_like(x)
    edges = marks + [n]
    # expected: segment before the first label uses `value`, each label's value applies from its index on
    seg_start = 0
    for k, e in enumerate(edges):
        seg = x[seg_start:e]
        if len(seg):
            want[seg_start:e] = oracle.rotate(seg, cur) if rotate else oracle.scale(seg, cur, True)
        if k < len(marks):
            cur = vals[k]
        seg_start = e
    guard = 0
    while pos < n and guard < 50:
        guard += 1
        labs = [B.Label("coef", m - pos, v) for m, v in zip(marks, vals) if m >= pos] + [B.Label(l.id, l.index - pos, l.data) for l in others if l.index >= pos]
        labs.sort(key=lambda l: l.index)
        y, c, p, _, _ = blk.work(x[pos:], n, labs)
        assert c == p and c > 0
        out.append(y); pos += c
    got = np.concatenate(out)
    assert pos == n and np.array_equal(got, want)


@pytest.mark.parametrize("seed", SEEDS)
def test_scale_rotate_extreme_coefficients(oracle, dev, seed):
    """factors / phases across many magnitudes, including ones whose Q-format image overflows the Q type
    (floatToQ's double -> integer conversion out of range) and denormal-range floats"""
    rng = np.random.default_rng(3000 + seed)
    scalar = [oracle.I16, oracle.F32, oracle.I8, oracle.I32, oracle.I64, oracle.F64][seed % 6]
    cplx = bool(rng.integers(0, 2))
    n = int(rng.integers(1, 5000))
    x = rand_stream(rng, scalar, n, cplx)
    mag = 10.0 ** float(rng.uniform(-12, 12))
    factor = mag * (1 if rng.integers(0, 2) else -1)
    assert np.array_equal(dev.scale(x, factor, cplx), oracle.scale(x, factor, cplx)), (scalar, factor)
    if cplx:
        phase = float(rng.uniform(-1, 1)) * 10.0 ** float(rng.uniform(-8, 6))
        assert np.array_equal(dev.rotate(x, phase), oracle.rotate(x, phase)), (scalar, phase)


ARITH_NP = [np.int8, np.int16, np.int32, np.int64, np.uint8, np.uint16, np.uint32, np.uint64, np.float32, np.float64]


@pytest.mark.parametrize("seed", SEEDS)
def test_arith_split_combine_random(oracle, dev, seed):
    rng = np.random.default_rng(2000 + seed)
    dt = np.dtype(ARITH_NP[seed % 10])
    cplx = bool(rng.integers(0, 2))
    n = int(rng.integers(1, 40000))
    shape = (n, 2) if cplx else (n,)
    op = ["ADD", "SUB", "MUL", "DIV"][int(rng.integers(0, 4))]
    if dt.kind == "f":
        a = (rng.standard_normal(shape) * 10.0 ** rng.uniform(-3, 3)).astype(dt)
        b = (rng.standard_normal(shape) * 10.0 ** rng.uniform(-3, 3)).astype(dt)
        if op == "DIV":
            b = np.where(np.abs(b) < 1e-3, dt.type(1.5), b).astype(dt)
    else:
        info = np.iinfo(dt)
        a = rng.integers(info.min, info.max, size=shape, dtype=dt, endpoint=True)
        b = rng.integers(info.min, info.max, size=shape, dtype=dt, endpoint=True)
        if op == "DIV":
            small = rng.integers(-9 if info.min < 0 else 1, 10, size=shape)
            b = np.where(small == 0, 3, small).astype(dt)
            a = (a // 4).astype(dt)
    got = dev.arith(op, a, b, cplx)
    ref = oracle.arith(getattr(oracle, op), a, b, cplx)
    # every operation and type bit for bit -- float division included (libgcc's __divsc3 / __divdc3 restated on the device,
    # tests/test_special_values_gpu.py covers the exponent range and the special values)
    assert np.array_equal(got.view(np.uint8), ref.view(np.uint8)), (dt, cplx, op)
    if dt.kind != "u":
        re, im = a.reshape(-1)[:n], b.reshape(-1)[:n]
        z = dev.combine_complex(re, im)
        assert np.array_equal(z.view(np.uint8), oracle.combine_complex(re, im).view(np.uint8))
        r2, i2 = dev.split_complex(z)
        assert np.array_equal(r2.view(np.uint8), np.ascontiguousarray(re).view(np.uint8)) and np.array_equal(i2.view(np.uint8), np.ascontiguousarray(im).view(np.uint8))


@pytest.mark.parametrize("seed", SEEDS)
def test_fir_edge_geometries(oracle, dev, seed):
    """few samples, K close to or beyond the buffer, long taps on every plan boundary, tiny output room"""
    rng = np.random.default_rng(1000 + 7 * seed)
    K = int([1, 2, 15, 16, 17, 255, 256, 257, 2048, 2049, 2050, 4096, 4097, 4098, 8192, 8193, 8194][seed % 17])
    extra = int(rng.integers(0, 3)) * int(rng.integers(0, 9000))
    n_in = max(1, K - 1 + int(rng.integers(-2, 3)) + extra)
    out_cap = int(rng.integers(0, 3)) * int(rng.integers(1, 10000)) + int(rng.integers(0, 2))
    ctaps = bool(rng.integers(0, 2))
    taps = (rng.normal(size=K) + (1j * rng.normal(size=K) if ctaps else 0)) / np.sqrt(K)
    x = rand_stream(rng, oracle.F32, n_in, True)
    ref = oracle.Fir(oracle.F32, True, ctaps); ref.set_taps(taps); ref.activate()
    f = dev.FirFilter("complex_float32", "COMPLEX" if ctaps else "REAL"); f.set_taps(taps)
    want, rc, rp, _ = ref.work(x, out_cap)
    got, gc, gp = f.process(x, out_cap)
    assert (gc, gp) == (rc, rp), (K, n_in, out_cap)
    if rp:
        # with a handful of outputs max|ref| can be a single heavily cancelled value (seed 430: one output of
        # 0.007 from 255 unit-scale products) and the 1e-5 bar degenerates; floor the scale at 10 % of the
        # filter's typical output level sqrt(sum |h|^2) * rms(x)
        typical = float(np.sqrt(np.sum(np.abs(taps) ** 2)) * np.sqrt(np.mean(x.astype(np.float64) ** 2) * 2))
        scale = max(float(np.abs(want).max()), 0.1 * typical)
        # thousands of taps: the reference's sequential float32 sum is itself 1e-5..5e-5 away from the exact
        # convolution (seed 3771: 4.3e-5 at K = 8192, the device 2.7e-6), so the device is held to the 1e-5 bar
        # against a float64 sum on a few outputs and to the reference within the reference's own error
        idx = np.unique(np.linspace(0, rp - 1, min(rp, 24)).astype(int))
        h = taps.astype(np.complex64).astype(np.complex128) if ctaps else taps.astype(np.float32).astype(np.float64)
        xc = x[:, 0].astype(np.float64) + 1j * x[:, 1].astype(np.float64)
        exact = np.array([np.dot(h, xc[n + K - 1 - np.arange(K)]) for n in idx])
        ex = np.stack([exact.real, exact.imag], 1)
        ref_noise = float(np.abs(want[idx] - ex).max())
        if K > 8193:
            # beyond every frequency-domain plan AUTO runs the reference's own operation order: its rounding noise
            # (seed 10029: above 1e-5 of float64) is reproduced bit for bit
            assert f.last_algo == dev._lib.FIR_EXACT and np.array_equal(got, want), (K, n_in, out_cap, diff_note(got, want), got.ctypes.data % 4096)
        else:
            assert float(np.abs(got[idx] - ex).max()) <= TOL * scale, (K, n_in, out_cap)
            assert float(np.abs(got - want).max()) <= TOL * scale + 2.0 * ref_noise, (K, n_in, out_cap)


@pytest.mark.parametrize("seed", SEEDS)
def test_fft_and_demod_blocks_random_chunks(oracle, seed):
    """/comms/fft consumes and produces whole frames only, whatever the buffer sizes (FFT.cpp:61-72);
    /comms/freq_demod carries _prev across work() calls (FreqDemod.cpp:49-71)"""
    from pothoscomms_amd import blocks as B
    rng = np.random.default_rng(500 + seed)
    nbins = int([8, 64, 100, 256, 1000, 1024, 4096][seed % 7])
    inverse = bool(rng.integers(0, 2))
    nframes = int(rng.integers(1, 12))
    x = rand_stream(rng, oracle.F32, nbins * nframes + int(rng.integers(0, nbins)), True)
    blk = B.make("/comms/fft", "complex_float32", nbins, inverse)
    blk.activate()
    pos, avail, outs, guard = 0, 0, [], 0
    while guard < 200:
        guard += 1
        avail = min(len(x), avail + int(rng.integers(1, 3 * nbins)))
        room = int(rng.integers(0, 4 * nbins))
        y, c, p, _, _ = blk.work(x[pos:avail], room)
        whole = min(avail - pos, room) // nbins * nbins
        assert c == p == whole, (nbins, avail - pos, room)
        outs.append(y); pos += c
        if avail == len(x) and len(x) - pos < nbins:
            break
    got = np.concatenate(outs) if outs else np.zeros((0, 2), np.float32)
    nf = len(got) // nbins
    assert nf == nframes or guard >= 200
    if nf:
        assert nerr(got, oracle.fft(x[:nf * nbins], nbins, inverse)) <= TOL
    # FreqDemod block on an int16 stream: bit-exact across arbitrary cuts
    n = int(rng.integers(1, 20000))
    xi = rand_stream(rng, oracle.I16, n, True)
    dm, ref = B.make("/comms/freq_demod", "complex_int16"), oracle.FreqDemod(oracle.I16)
    dm.activate(); ref.activate()
    cuts = sorted(set([0, n] + [int(c) for c in rng.integers(0, n + 1, int(rng.integers(0, 6)))]))
    for a, b in zip(cuts[:-1], cuts[1:]):
        y, c, p, _, _ = dm.work(xi[a:b], b - a)
        assert (c, p) == (b - a, b - a)
        assert np.array_equal(y, ref.work(xi[a:b]))


# ---- guard bands: nothing outside [out, out + produced) may be written, nothing outside the declared input read ----
GUARD = 4096   # elements of poison on either side
GSEEDS = range(2 * len(SEEDS))


def _guarded(torch, d, n_elems, width, dtype, fill):
    """A device buffer with GUARD poisoned elements on either side of an n_elems window; returns (whole, window)."""
    whole = torch.full(((n_elems + 2 * GUARD) * width,), fill, dtype=dtype, device=d)
    win = whole[GUARD * width:(GUARD + n_elems) * width]
    return whole, (win.view(-1, width) if width > 1 else win)


def _bands_intact(whole, n_elems, width, fill):
    lo, hi = whole[:GUARD * width], whole[(GUARD + n_elems) * width:]
    if fill != fill:      # NaN poison
        return bool(lo.isnan().all()) and bool(hi.isnan().all())
    return bool((lo == fill).all()) and bool((hi == fill).all())


@pytest.mark.parametrize("seed", GSEEDS)
def test_fir_writes_only_its_outputs_and_reads_only_its_inputs(oracle, dev, seed):
    """Every FIR pipeline with range-checked stores (overlap-save 4096 / partitioned long taps / decimating / interpolating / polyphase,
    the direct tile, the sliding window): NaN guard bands around input AND output.  A read past the declared input would
    pull a NaN into the last outputs; a store outside [0, produced) would overwrite the poison."""
    import torch
    from pothoscomms_amd import _lib
    d = torch.device("cuda", 0)
    rng = np.random.default_rng(9100 + seed)
    nan = float("nan")
    geoms = [(1, 1), (1, 1), (1, 2), (1, 8), (2, 1), (4, 1), (3, 1), (3, 2), (1, 1), (1, 5)]
    L, M = geoms[seed % len(geoms)]
    ntaps = int(rng.choice([2, 17, 63, 255, 1000, 3000, 5000, 8193])) if L == 1 else int(rng.choice([8, 63, 255])) * L   # (L = 1, M > 1 beyond 2049 taps: the partitioned kernel's decimating store)
    algo = [_lib.FIR_AUTO, _lib.FIR_OLS_FFT, _lib.FIR_DIRECT, _lib.FIR_EXACT][seed % 4] if (L, M) == (1, 1) else _lib.FIR_AUTO
    h = (rng.normal(size=ntaps) + 1j * rng.normal(size=ntaps)) / ntaps
    f = dev.FirFilter("complex_float32", "COMPLEX")
    f.set_taps(h); f.set_decimation(M); f.set_interpolation(L); f.set_algo(algo)
    K = f.K
    n_iter = int(rng.integers(1, 40000)) // M * M + M
    n_in, n_out = n_iter + K - 1, n_iter // M * L
    xw, x = _guarded(torch, d, n_in, 2, torch.float32, nan)
    yw, y = _guarded(torch, d, n_out, 2, torch.float32, nan)
    xh = rng.uniform(-1, 1, (n_in, 2)).astype(np.float32)
    x.copy_(torch.from_numpy(xh).to(d))
    c, p = f.process_dev(x, y, n_in, n_out)
    torch.cuda.synchronize()
    assert (c, p) == (n_iter, n_out)
    assert _bands_intact(yw, n_out, 2, nan), (L, M, ntaps, algo)
    got = y.cpu().numpy()
    assert np.isfinite(got).all(), (L, M, ntaps, algo)        # no NaN from beyond the input window
    blk = oracle.Fir(oracle.F32, True, True)
    blk.set_taps(h); blk.set_decimation(M); blk.set_interpolation(L); blk.activate()
    ref, rc, rp, _ = blk.work(xh, n_out)
    assert (rc, rp) == (c, p)
    assert nerr(got, ref) <= TOL


@pytest.mark.parametrize("seed", GSEEDS)
def test_real_fir_writes_only_its_outputs_and_reads_only_its_inputs(oracle, dev, seed):
    """real float32 streams: the two-blocks-per-transform kernel and, beyond 2049 taps or decimating, the call's two halves side by side
    through the partitioned kernel (the second half reads at an offset of half the outputs: whatever lies behind the input must stay unread).
    NaN guard bands around input and output; tiny, odd and large calls."""
    import torch
    from pothoscomms_amd import _lib
    d = torch.device("cuda", 0)
    rng = np.random.default_rng(9700 + seed)
    nan = float("nan")
    ntaps = int(rng.choice([2, 63, 255, 2049, 2050, 3000, 4097, 5000, 8193]))
    M = int(rng.choice([1, 1, 2, 7, 64, 1000]))
    h = rng.normal(size=ntaps) / np.sqrt(ntaps)
    f = dev.FirFilter("float32", "REAL")
    f.set_taps(h); f.set_decimation(M)
    n_out = int(rng.choice([1, 2, 31, 33, 2047, 2049, int(rng.integers(1, 60000 // M + 2))]))
    n_in = n_out * M + ntaps - 1
    xw, x = _guarded(torch, d, n_in, 1, torch.float32, nan)
    yw, y = _guarded(torch, d, n_out, 1, torch.float32, nan)
    xh = rng.uniform(-1, 1, n_in).astype(np.float32)
    x.copy_(torch.from_numpy(xh).to(d))
    c, p = f.process_dev(x, y, n_in, n_out)
    torch.cuda.synchronize()
    assert (c, p) == (n_out * M, n_out)
    assert f.last_algo == _lib.FIR_OLS_FFT or (M > 1 and ntaps < 16)
    assert _bands_intact(yw, n_out, 1, nan), (ntaps, n_out, M)
    got = y.cpu().numpy()
    assert np.isfinite(got).all(), (ntaps, n_out)             # no NaN from beyond the input window
    blk = oracle.Fir(oracle.F32, False, False)
    blk.set_taps(h); blk.set_decimation(M); blk.activate()
    ref, rc, rp, _ = blk.work(xh, n_out)
    assert (rc, rp) == (c, p)
    typical = float(np.sqrt(np.sum(h ** 2)) * np.sqrt(np.mean(xh.astype(np.float64) ** 2)))
    assert float(np.abs(got - ref).max()) <= TOL * max(float(np.abs(ref).max()), 0.1 * typical) * (3.0 if ntaps > 2049 else 1.0), (ntaps, n_out, M)


@pytest.mark.parametrize("seed", GSEEDS)
def test_fft_writes_only_its_frames(oracle, dev, seed):
    import torch
    d = torch.device("cuda", 0)
    rng = np.random.default_rng(9300 + seed)
    nbins = int(rng.choice([4, 60, 256, 1000, 4096, 4096, 8192, 3 * 1024, 65536]))
    dtype, td, sc = [("complex_float32", torch.float32, oracle.F32), ("complex_float64", torch.float64, oracle.F64),
                     ("complex_int16", torch.int16, oracle.I16)][seed % 3]
    nframes = int(rng.integers(1, 9)) if nbins >= 4096 else int(rng.integers(1, 70))
    fill = float("nan") if sc != oracle.I16 else 12345
    xw, x = _guarded(torch, d, nbins * nframes, 2, td, fill)
    yw, y = _guarded(torch, d, nbins * nframes, 2, td, fill)
    xh = rand_stream(rng, sc, nbins * nframes, True) if sc != oracle.I16 else rng.integers(-3000, 3000, (nbins * nframes, 2)).astype(np.int16)
    x.copy_(torch.from_numpy(xh).to(d))
    inv = bool(seed & 1)
    dev.Fft(dtype, nbins, inv).transform_dev(x, y, nframes)
    torch.cuda.synchronize()
    assert _bands_intact(yw, nbins * nframes, 2, fill), (nbins, dtype)
    got, ref = y.cpu().numpy(), oracle.fft(xh, nbins, inv)
    if sc == oracle.I16:
        assert np.array_equal(got, ref)
    else:
        assert np.isfinite(got).all()
        assert nerr(got, ref) <= (TOL if sc == oracle.F32 else 1e-12)


@pytest.mark.parametrize("seed", GSEEDS)
def test_fm_chain_writes_only_its_outputs(oracle, dev, seed):
    import torch
    from pothoscomms_amd import _lib
    d = torch.device("cuda", 0)
    rng = np.random.default_rng(9500 + seed)
    ntaps = int(rng.choice([1, 9, 127, 500, 2048]))
    n = int(rng.integers(1, 50000))
    h = rng.normal(size=ntaps) / ntaps
    nan = float("nan")
    xw, x = _guarded(torch, d, n + ntaps - 1, 2, torch.float32, nan)
    yw, y = _guarded(torch, d, n, 1, torch.float32, nan)
    ph = np.cumsum(rng.uniform(-0.5, 0.5, n + ntaps - 1))
    xh = np.stack([np.cos(ph), np.sin(ph)], 1).astype(np.float32)
    x.copy_(torch.from_numpy(xh).to(d))
    ch = dev.FmChain(); ch.set_phase(0.3); ch.set_taps(h, False)
    ch.set_algo([_lib.FIR_AUTO, _lib.FIR_DIRECT, _lib.FIR_OLS_FFT][seed % 3])
    assert ch.process_dev(x, y, n + ntaps - 1, n) == (n, n)
    torch.cuda.synchronize()
    assert _bands_intact(yw, n, 1, nan), ntaps
    got = y.cpu().numpy()
    assert np.isfinite(got).all()
    fir = oracle.Fir(oracle.F32, True, False); fir.set_taps(h); fir.activate()
    yy, _, p, _ = fir.work(oracle.rotate(xh, 0.3), n)
    # (ill-conditioned where the filtered envelope vanishes: compare where it does not)
    ref = oracle.FreqDemod(oracle.F32).work(yy)
    mag = np.hypot(yy[:, 0], yy[:, 1])
    ok = np.minimum(mag, np.concatenate([[1.0], mag[:-1]])) > 3e-2 * mag.max()   # the angle of a product of two samples: error ~ FIR error / |y|
    assert ang_err(got[ok], ref[ok]) <= 2 * TOL
