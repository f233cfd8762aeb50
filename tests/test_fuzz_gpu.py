"""Randomised GPU parity sweeps (PCX_FUZZ_SEEDS sizes them; the default keeps the suite short):
element-wise maps on device buffers at arbitrary offsets, FreqDemod and the fused FM chain fed in
random work()-sized pieces with the state carried between calls, all against the oracle."""
import os

import numpy as np
import pytest

from tests.util import TOL, ang_err, nerr, rand_stream

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
    xin[oi:oi + n] = torch.from_numpy(x).to(d)
    xv = xin[oi:oi + n]
    outc = torch.zeros((n + 8, 2), dtype=TD, device=d)
    outr = torch.zeros((n + 8,), dtype=TD, device=d)
    oc, orr = outc[oo:oo + n], outr[oo:oo + n]
    phase = float(rng.uniform(-3, 3))
    dev.rotate(xv, phase, scalar=scalar, out=oc, n=n)
    assert np.array_equal(oc.cpu().numpy(), oracle.rotate(x, phase))
    dev.scale(xv, 0.37, True, scalar=scalar, out=oc, n=n)
    assert np.array_equal(oc.cpu().numpy(), oracle.scale(x, 0.37, True))
    dev.conj(xv, scalar=scalar, out=oc, n=n)
    assert np.array_equal(oc.cpu().numpy(), oracle.conj(x))
    dev.abs_(xv, True, scalar=scalar, out=orr, n=n)
    ref = oracle.abs_(x, True)
    got = orr.cpu().numpy()
    assert (nerr(got, ref) <= 1e-15 * 4) if scalar == oracle.F64 else np.array_equal(got, ref)
    dev.angle(xv, scalar=scalar, out=orr, n=n)
    ref, got = oracle.angle(x), orr.cpu().numpy()
    assert (ang_err(got, ref) <= TOL) if scalar in (oracle.F32, oracle.F64) else np.array_equal(got, ref)
    y = rand_stream(rng, scalar, n, True)
    yin = torch.zeros((n + 8, 2), dtype=TD, device=d)
    yin[oo:oo + n] = torch.from_numpy(y).to(d)
    for op in ("ADD", "SUB", "MUL"):
        dev.arith(op, xv, yin[oo:oo + n], True, scalar=scalar, out=oc, n=n)
        assert np.array_equal(oc.cpu().numpy().view(np.uint8), oracle.arith(getattr(oracle, op), x, y, True).view(np.uint8)), op
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
    ctaps = bool(rng.integers(0, 2))
    n = int(rng.integers(ntaps + 1, 40000))
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
    ch.set_algo(algo)
    xp = x.view(np.float32).reshape(-1, 2) if np.iscomplexobj(x) else x
    pos, outs = 0, []
    ends = sorted(set([n] + [int(c) for c in rng.integers(ntaps, n + 1, int(rng.integers(0, 5)))]))
    for end in ends:
        g, c, p = ch.process(xp[pos:end], n)
        outs.append(g); pos += c
    got = np.concatenate(outs)
    assert len(got) == produced == n - ntaps + 1
    assert ang_err(got, ref) <= TOL
