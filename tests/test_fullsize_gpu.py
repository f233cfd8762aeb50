"""GPU suite at BASELINE.json's full sizes (device-resident, through the *_dev entry points).
The oracle cannot filter 64 Mi samples in seconds, so these check size-independent properties
plus oracle parity on windows (first / last 64 Ki outputs and block seams)."""
import numpy as np
import pytest

from tests.util import TOL, nerr

pytestmark = pytest.mark.gpu
C1 = 64 * 1024 * 1024


@pytest.fixture(scope="module")
def torch_dev():
    import torch
    assert torch.cuda.is_available()
    return torch, torch.device("cuda", 0)


def test_fir255_64Mi_windows_and_cross_check(oracle, dev, torch_dev):
    """configs[1]: 255 taps, 64 Mi samples.  OLS vs oracle on windows; OLS vs direct everywhere."""
    torch, d = torch_dev
    from pothoscomms_amd import _lib, taps as tp
    h = tp.c1_taps()
    K = len(h)
    x = torch.empty((C1 + K - 1, 2), dtype=torch.float32, device=d)
    dev.fill_uniform_f32_dev(x, seed=2, offset=0)
    y_ols = torch.empty((C1, 2), dtype=torch.float32, device=d)
    y_dir = torch.empty((C1, 2), dtype=torch.float32, device=d)
    for algo, y in ((_lib.FIR_OLS_FFT, y_ols), (_lib.FIR_DIRECT, y_dir)):
        f = dev.FirFilter("complex_float32", "COMPLEX"); f.set_taps(h); f.set_algo(algo)
        c, p = f.process_dev(x, y)
        assert (c, p) == (C1, C1)
    torch.cuda.synchronize()
    # the two kernels (frequency domain / time domain) agree over the WHOLE stream
    scale = float(y_dir.abs().max())
    assert float((y_ols - y_dir).abs().max()) / scale <= TOL
    # oracle on windows: start, end, and around overlap-save block seams (S = 4096 - 254)
    ref = oracle.Fir(oracle.F32, True, True); ref.set_taps(h); ref.activate()
    S = 4096 - (K - 1)
    for start in (0, S - 300, 1000 * S - 300, C1 - 65536):
        n = 65536 if start in (0, C1 - 65536) else 600
        win = x[start:start + n + K - 1].cpu().numpy()
        want, _, p, _ = ref.work(win, n)
        assert p == n
        assert nerr(y_ols[start:start + n].cpu().numpy(), want) <= TOL, start
    # the device stream equals the oracle's generator bit for bit (same counter hash)
    assert np.array_equal(x[:1000].cpu().numpy().ravel(), oracle.fill_uniform_f32(2000, 2, 0))


@pytest.mark.parametrize("K", [4097, 6145, 8193])
def test_long_tap_fir_64Mi_windows_at_run_and_block_seams(oracle, dev, torch_dev, K):
    """The taps in partitions (fir_ols_part.hip): 64 Mi samples, 32768 blocks of 2048 outputs in 512 runs of 64.  The oracle on windows:
    the start of the stream (the windows in front of the buffer), the first run seam (where a workgroup computes its own history
    spectra), seams deep in the stream and the ragged end; an impulse in front of a run seam returns the taps across it."""
    torch, d = torch_dev
    from pothoscomms_amd import _lib, taps as tp
    h = tp.complex_bandpass(K, 0.05, 0.05)
    n = C1 + 1234                                         # a ragged last block
    x = torch.empty((n + K - 1, 2), dtype=torch.float32, device=d)
    dev.fill_uniform_f32_dev(x, seed=5, offset=0)
    y = torch.empty((n, 2), dtype=torch.float32, device=d)
    f = dev.FirFilter("complex_float32", "COMPLEX"); f.set_taps(h)
    assert f.process_dev(x, y) == (n, n) and f.last_algo == _lib.FIR_OLS_FFT
    torch.cuda.synchronize()
    ref = oracle.Fir(oracle.F32, True, True); ref.set_taps(h); ref.activate()
    nblocks = -(-n // 2048)
    run = nblocks // 512 + 1                               # the longer runs come first (a balanced partition)
    for start in (0, 2048 - 200, run * 2048 - 300, 7 * run * 2048 - 300, 300 * 2048 * 64 + 1800, n - 3000):
        m = 3000 if start in (0, n - 3000) else 600
        win = x[start:start + m + K - 1].cpu().numpy()
        want, _, p, _ = ref.work(win, m)
        assert p == m and nerr(y[start:start + m].cpu().numpy(), want) <= TOL, (K, start)
    # an impulse: the taps come back, across block and run seams
    x.zero_()
    at = run * 2048 - 1000 + K - 1
    x[at, 0] = 1.0
    assert f.process_dev(x, y) == (n, n)
    torch.cuda.synchronize()
    got = y[at - (K - 1):at + 1].cpu().numpy()
    hh = np.stack([h.real, h.imag], axis=1).astype(np.float32)
    assert np.abs(got - hh).max() <= 1e-5 * np.abs(hh).max() + 1e-7
    assert float(y[:at - (K - 1)].abs().max()) <= 1e-6 and float(y[at + 1:].abs().max()) <= 1e-6


@pytest.mark.parametrize("K", [4097, 8193])
def test_long_tap_real_fir_64Mi_windows(oracle, dev, torch_dev, K):
    """real float32, long taps: the call's two halves ride the partitioned kernel side by side as one complex stream.  The oracle on
    windows at the start, around the seam between the halves (the second half's history is the end of the first), at run seams of
    both halves and at the ragged end."""
    torch, d = torch_dev
    from pothoscomms_amd import _lib, taps as tp
    h = tp.lowpass(K, 0.07)
    n = C1 + 4321
    x = torch.empty((n + K - 1 + 1) // 2 * 2, dtype=torch.float32, device=d)
    dev.fill_uniform_f32_dev(x.view(-1, 2), seed=6, offset=0)
    x = x[:n + K - 1]
    y = torch.empty(n, dtype=torch.float32, device=d)
    f = dev.FirFilter("float32", "REAL"); f.set_taps(h)
    assert f.process_dev(x, y) == (n, n) and f.last_algo == _lib.FIR_OLS_FFT
    torch.cuda.synchronize()
    ref = oracle.Fir(oracle.F32, False, False); ref.set_taps(h); ref.activate()
    half = ((n + 1) // 2 + 31) // 32 * 32
    run = -(-half // 2048) // 512 + 1
    for start in (0, 2048 - 200, run * 2048 - 300, half - 1500, half + run * 2048 - 300, n - 3000):
        m = 3000
        win = x[start:start + m + K - 1].cpu().numpy()
        want, _, p, _ = ref.work(win, m)
        assert p == m and nerr(y[start:start + m].cpu().numpy(), want) <= TOL, (K, start)


def test_fir_linearity_and_impulse_full_size(dev, torch_dev):
    """FIR(a x1 + x2) = a FIR(x1) + FIR(x2); an impulse returns the taps."""
    torch, d = torch_dev
    from pothoscomms_amd import taps as tp
    h = tp.c1_taps()
    K, n = len(h), 8 * 1024 * 1024
    f = dev.FirFilter("complex_float32", "COMPLEX"); f.set_taps(h)
    x1 = torch.empty((n + K - 1, 2), dtype=torch.float32, device=d); dev.fill_uniform_f32_dev(x1, seed=11)
    x2 = torch.empty_like(x1); dev.fill_uniform_f32_dev(x2, seed=12)
    y1, y2, y3 = (torch.empty((n, 2), dtype=torch.float32, device=d) for _ in range(3))
    f.process_dev(x1, y1); f.process_dev(x2, y2)
    f.process_dev(0.5 * x1 + x2, y3)
    assert float((y3 - (0.5 * y1 + y2)).abs().max()) / float(y3.abs().max()) <= TOL
    imp = torch.zeros((n + K - 1, 2), dtype=torch.float32, device=d)
    pos = 5 * 3842 + 17
    imp[K - 1 + pos, 0] = 1.0
    f.process_dev(imp, y1)
    got = y1[pos:pos + K].cpu().numpy()
    want = np.stack([h.real, h.imag], 1).astype(np.float32)
    assert np.max(np.abs(got - want)) <= 1e-6
    assert float(y1[:pos].abs().max()) <= 1e-6 and float(y1[pos + K:].abs().max()) <= 1e-6


def test_fft4096_65536_frames_roundtrip_and_parseval(oracle, dev, torch_dev):
    """configs[2]: 65,536 frames of 4096.  ifft(fft(x)) = N x, Parseval per frame, oracle on 3 frames."""
    torch, d = torch_dev
    nframes = 65536
    x = torch.empty((nframes * 4096, 2), dtype=torch.float32, device=d)
    dev.fill_uniform_f32_dev(x, seed=3)
    X = torch.empty_like(x); xb = torch.empty_like(x)
    dev.Fft("complex_float32", 4096, False).transform_dev(x, X, nframes)
    dev.Fft("complex_float32", 4096, True).transform_dev(X, xb, nframes)
    torch.cuda.synchronize()
    assert float((xb / 4096.0 - x).abs().max()) <= 5e-6
    e_t = (x.double() ** 2).view(nframes, -1).sum(1)
    e_f = (X.double() ** 2).view(nframes, -1).sum(1) / 4096.0
    assert float(((e_t - e_f).abs() / e_t).max()) <= 1e-5
    for fr in (0, 31337, nframes - 1):
        want = oracle.fft(x[fr * 4096:(fr + 1) * 4096].cpu().numpy(), 4096)
        assert nerr(X[fr * 4096:(fr + 1) * 4096].cpu().numpy(), want) <= TOL


def _fm_stream(torch, d, n, K):
    """64 Mi samples of the C4 test signal: 1 Mi samples generated on the host (taps.fm_test_signal), tiled on the device.
    (The phase jumps at the tile seams are part of the stream both sides see.)"""
    from pothoscomms_amd import taps as tp
    xs = tp.fm_test_signal(1 << 20)
    reps = (n + K - 1 + (1 << 20) - 1) // (1 << 20)
    return torch.from_numpy(xs.view(np.float32).reshape(-1, 2)).to(d).repeat(reps, 1)[:n + K - 1].contiguous()


def test_fm_chain_64Mi_against_the_oracle_chain(oracle, dev, torch_dev):
    """configs[4] at full size, checked against the ORACLE's Rotate -> FIR(127 real taps) -> FreqDemod on windows: the
    first and last 64 Ki outputs, the overlap-save block seams (multiples of S = 4096 - 128), the seam between two
    work() calls (FreqDemod's carried prev, FreqDemod.cpp:63-65), and a tile seam of the test signal."""
    torch, d = torch_dev
    from pothoscomms_amd import taps as tp
    from tests.util import ang_err
    n, h, phase = C1, tp.c4_taps(), tp.C4_PHASE
    K = len(h)
    x = _fm_stream(torch, d, n, K)
    got = torch.empty(n, dtype=torch.float32, device=d)
    ch = dev.FmChain(); ch.set_phase(phase); ch.set_taps(h, False)
    n1 = 40 * 1024 * 1024 + 12345                        # two calls: the second continues from the carried state
    assert ch.process_dev(x, got, n1 + K - 1, n1) == (n1, n1)
    assert ch.process_dev(x[n1:], got[n1:], n - n1 + K - 1, n - n1) == (n - n1, n - n1)
    torch.cuda.synchronize()

    def oracle_window(start, cnt):
        first = max(start - 1, 0)                        # one FIR output in front seeds the demodulator
        m = start - first + cnt
        win = x[first:first + m + K - 1].cpu().numpy()
        fir = oracle.Fir(oracle.F32, True, False); fir.set_taps(h); fir.activate()
        y, _, p, _ = fir.work(oracle.rotate(win, phase), m)
        assert p == m
        return oracle.FreqDemod(oracle.F32).work(y)[start - first:]

    S = 4096 - 128
    windows = [(0, 65536), (n - 65536, 65536), (S - 300, 600), (1000 * S - 300, 600), (16000 * S - 300, 600),
               (n1 - 300, 600), ((1 << 20) - 400, 800), (n1 + ((n - n1) // S // 2) * S - 300, 600)]
    for start, cnt in windows:
        want = oracle_window(start, cnt)
        assert ang_err(got[start:start + cnt].cpu().numpy(), want) <= TOL, start


def test_fm_chain_full_size_against_separate_blocks(dev, torch_dev):
    """configs[4], 64 Mi samples: the fused kernel equals Rotate -> FIR -> FreqDemod run as three device calls, over the
    WHOLE stream (the oracle comparison above covers windows)."""
    torch, d = torch_dev
    from pothoscomms_amd import _lib, taps as tp
    n, h, phase = C1, tp.c4_taps(), tp.C4_PHASE
    K = len(h)
    x = _fm_stream(torch, d, n, K)
    fused = torch.empty(n, dtype=torch.float32, device=d)
    ch = dev.FmChain(); ch.set_phase(phase); ch.set_taps(h, False)
    assert ch.process_dev(x, fused, n + K - 1, n) == (n, n)
    xr = torch.empty_like(x)
    dev.rotate(x, phase, scalar=dev.F32, out=xr, n=n + K - 1)
    f = dev.FirFilter("complex_float32", "REAL"); f.set_taps(h); f.set_algo(_lib.FIR_DIRECT)
    y = torch.empty((n, 2), dtype=torch.float32, device=d)
    f.process_dev(xr, y)
    sep = torch.empty(n, dtype=torch.float32, device=d)
    dev.FreqDemod("complex_float32").process_dev(y, sep, n)
    torch.cuda.synchronize()
    del xr, y
    dd = (fused.double() - sep.double() + np.pi) % (2 * np.pi) - np.pi
    assert float(dd.abs().max()) / np.pi <= TOL


def test_sharded_step_split_equals_single_call(dev, torch_dev):
    """The N>1 pass (body while the halo is in flight, then the head) gives the same outputs as
    one call -- exercised on one GPU by making the ring believe it has a neighbour."""
    torch, d = torch_dev
    from pothoscomms_amd import taps as tp
    from pothoscomms_amd.stream import ShardedFir
    C = 1 << 20
    sf = ShardedFir(tp.c1_taps(), C, d)
    dev.fill_uniform_f32_dev(sf.buf, seed=9)
    whole = sf.step().clone()
    sf.ring.world = 2                       # take the multi-rank path; rank 0 has nothing to receive
    sf.ring.rank = 0
    sf.ring.start = lambda buf: []
    sf.out.zero_()
    split = sf.step()
    torch.cuda.synchronize()
    assert float((split - whole).abs().max()) / float(whole.abs().max()) <= 2e-6


def test_dynamic_dealing_is_deterministic_and_complete(dev, torch_dev):
    """The three dealt kernels (pcx_sched.hpp) hand blocks to whichever workgroup draws first: every launch must still produce
    every block exactly once.  200 launches each at full size, outputs poisoned before every launch, compared BIT FOR BIT with
    the first launch (a block's arithmetic does not depend on who runs it) -- a skipped block leaves NaNs, a block computed from
    a stale draw differs."""
    torch, d = torch_dev
    from pothoscomms_amd import taps as tp
    n = C1
    # FIR
    h = tp.c1_taps()
    K = len(h)
    x = torch.empty((n + K - 1, 2), dtype=torch.float32, device=d)
    dev.fill_uniform_f32_dev(x, seed=2)
    y0 = torch.empty((n, 2), dtype=torch.float32, device=d)
    y = torch.empty_like(y0)
    f = dev.FirFilter("complex_float32", "COMPLEX"); f.set_taps(h)
    f.process_dev(x, y0)
    assert bool(torch.isfinite(y0).all())
    for i in range(200):
        y.fill_(float("nan"))
        f.process_dev(x, y)
        if i % 20 == 19 or i < 3:
            assert torch.equal(y, y0), i
    assert torch.equal(y, y0)
    # ragged sizes around the pair structure (odd block counts, a last chunk of one block)
    for m in (3840 * 2049, 3840 * 2050 + 17, 3840 * 4097 - 1, 3840 * 2048 + 1):
        f.process_dev(x, y0, m + K - 1, m)
        y.fill_(float("nan"))
        f.process_dev(x, y, m + K - 1, m)
        assert torch.equal(y[:m], y0[:m]) and bool(torch.isfinite(y[:m]).all()) and bool(torch.isnan(y[m:]).all()), m
    del x, y, y0
    # FFT 4096
    nframes = 65536
    xf = torch.empty((nframes * 4096, 2), dtype=torch.float32, device=d)
    dev.fill_uniform_f32_dev(xf, seed=3)
    X0 = torch.empty_like(xf); X = torch.empty_like(xf)
    t = dev.Fft("complex_float32", 4096, False)
    t.transform_dev(xf, X0, nframes)
    for i in range(60):
        X.fill_(float("nan"))
        t.transform_dev(xf, X, nframes)
        if i % 10 == 9:
            assert torch.equal(X, X0), i
    for nf in (2049, 2050, 4097, 65535):
        t.transform_dev(xf, X0, nf)
        X.fill_(float("nan"))
        t.transform_dev(xf, X, nf)
        assert torch.equal(X[:nf * 4096], X0[:nf * 4096]) and bool(torch.isnan(X[nf * 4096:]).all()), nf
    del xf, X, X0
    # fused chain
    hc = tp.c4_taps()
    Kc = len(hc)
    xc = _fm_stream(torch, d, n, Kc)
    z0 = torch.empty(n, dtype=torch.float32, device=d); z = torch.empty_like(z0)
    ch = dev.FmChain(); ch.set_phase(tp.C4_PHASE); ch.set_taps(hc, False)
    ch.process_dev(xc, z0, n + Kc - 1, n)
    for i in range(100):
        z.fill_(float("nan"))
        ch.reset()
        ch.process_dev(xc, z, n + Kc - 1, n)
        if i % 10 == 9:
            assert torch.equal(z, z0), i


@pytest.mark.parametrize("dtype,cplx,n,K", [("complex_int16", True, 24 << 20, 255), ("complex_int8", True, 24 << 20, 255), ("int16", False, 40 << 20, 255),
                                            ("complex_float64", True, 12 << 20, 255), ("float64", False, 24 << 20, 255),
                                            ("complex_int16", True, 6 << 20, 2049), ("complex_int16", True, 12 << 20, 17), ("int16", False, 12 << 20, 2049)])
def test_double_pipeline_fir_at_a_size_that_is_dealt(oracle, dev, torch_dev, dtype, cplx, n, K):
    """round 6: the double-precision overlap-save kernels draw their blocks from 512 persistent workgroups once a call is long enough
    (more than 2,048 blocks: fir_ols_f64.hip launch_ip / launch_real_ip).  Every block exactly once: the WHOLE stream equals the
    time-domain kernel's (EXACT: the reference's own operation order on the device) -- bit for bit for the integer types, to 1e-12
    for double --, and the oracle on windows at both ends and across block seams; the element count behind the last whole block
    is odd on purpose (a ragged last block, an empty fetch-ahead behind it)."""
    torch, d = torch_dev
    from pothoscomms_amd import _lib
    rng = np.random.default_rng(6)
    n = n + 12345
    scalar, _ = dev.parse_dtype(dtype)
    h = (rng.normal(size=K) + (1j * rng.normal(size=K) if cplx else 0)) / np.sqrt(K) * (0.5 if "int" in dtype else 1.0)
    shape = (n + K - 1, 2) if cplx else (n + K - 1,)
    tdt = {"complex_int16": torch.int16, "int16": torch.int16, "complex_int8": torch.int8, "complex_float64": torch.float64, "float64": torch.float64}[dtype]
    if "int" in dtype:
        amp = 100 if "int8" in dtype else 20000
        x = torch.randint(-amp, amp, shape, device=d).to(tdt)
    else:
        x = torch.rand(shape, dtype=tdt, device=d) - 0.5
    oshape = (n, 2) if cplx else (n,)
    y_ols = torch.empty(oshape, dtype=tdt, device=d)
    y_td = torch.empty(oshape, dtype=tdt, device=d)
    for algo, y in ((_lib.FIR_OLS_FFT, y_ols), (_lib.FIR_EXACT, y_td)):
        f = dev.FirFilter(dtype, "COMPLEX" if cplx else "REAL"); f.set_taps(h); f.set_algo(algo)
        c, p = f.process_dev(x, y)
        assert (c, p) == (n, n)
        assert f.last_algo == algo
    torch.cuda.synchronize()
    S = 4096 - (K - 1 + 15) // 16 * 16
    assert -(-n // S) // (1 if cplx else 2) > 2048              # the call IS long enough to be dealt
    if "int" in dtype:
        assert torch.equal(y_ols, y_td)
    else:
        assert float((y_ols - y_td).abs().max()) / float(y_td.abs().max()) <= 1e-12
    ref = oracle.Fir(scalar, cplx, cplx); ref.set_taps(h); ref.activate()
    for start in (0, S - 300, 2500 * S - 300, n - 20000):
        m = (20000 if K <= 255 else 3000) if start in (0, n - 20000) else 600
        ref.activate()
        want, _, p, _ = ref.work(x[start:start + m + K - 1].cpu().numpy(), m)
        got = y_ols[start:start + m].cpu().numpy()
        assert p == m
        if "int" in dtype:
            assert np.array_equal(got, want), start
        else:
            assert nerr(got, want) <= 1e-12, start
