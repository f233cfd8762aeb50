"""GPU suite beyond 2^32: one call over more than 4 Gi SAMPLES (34 GB in, 34 GB out of the 288 GB of HBM3E).  Every index,
byte offset and grid computation of the path has to be 64-bit clean: the windows checked against the oracle sit at the start,
on either side of the 2^32-BYTE, 2^31- and 2^32-SAMPLE marks, and at the very end.  Inputs are generated on the device by the
counter hash the oracle shares, so the oracle regenerates any window from its offset alone."""
import numpy as np
import pytest

from tests.util import TOL, ang_err, nerr

pytestmark = pytest.mark.gpu
N = (1 << 32) + (1 << 20) + 70001            # samples per call


@pytest.fixture(scope="module")
def big(dev):
    """one input stream of N + 4096 complex_float32 samples and room for N outputs, shared by the tests of this file"""
    import torch
    d = torch.device("cuda", 0)
    free, _ = torch.cuda.mem_get_info(d)
    if free < 90 * (1 << 30):
        pytest.skip("needs 90 GB of free device memory")
    x = torch.empty((N + 4096, 2), dtype=torch.float32, device=d)
    # the fill kernel itself is under test here: 8.6 G scalars in one launch
    dev.fill_uniform_f32_dev(x, seed=11, offset=0)
    y = torch.empty((N, 2), dtype=torch.float32, device=d)
    torch.cuda.synchronize()
    yield torch, d, x, y
    del x, y
    torch.cuda.empty_cache()


def _marks(n, w):
    """window starts: stream start, around 2^29 samples (= 2^32 bytes of complex_float32), 2^30, 2^31, 2^32, stream end"""
    m = [0, (1 << 29) - w // 2, (1 << 30) - w // 2, (1 << 31) - w // 2, (1 << 32) - w // 2, n - w]
    return [s for s in m if 0 <= s and s + w <= n]


def _window(oracle, start, cnt):
    return oracle.fill_uniform_f32(2 * cnt, 11, 2 * start).reshape(-1, 2)


def test_the_generator_is_64_bit_clean(oracle, big):
    torch, d, x, y = big
    for s in _marks(N + 4096, 2048):
        assert np.array_equal(x[s:s + 2048].cpu().numpy(), _window(oracle, s, 2048)), s


def test_maps_over_more_than_2_32_samples(oracle, dev, big):
    torch, d, x, y = big
    dev.rotate(x, 0.7, scalar=oracle.F32, out=y, n=N)
    torch.cuda.synchronize()
    for s in _marks(N, 4096):
        assert np.array_equal(y[s:s + 4096].cpu().numpy(), oracle.rotate(_window(oracle, s, 4096), 0.7)), s
    dev.conj(x, scalar=oracle.F32, out=y, n=N)
    torch.cuda.synchronize()
    for s in _marks(N, 4096):
        assert np.array_equal(y[s:s + 4096].cpu().numpy(), oracle.conj(_window(oracle, s, 4096))), s
    mag = y.view(-1)[:N]                                   # float32 outputs in the same storage
    dev.abs_(x, True, scalar=oracle.F32, out=mag, n=N)
    torch.cuda.synchronize()
    for s in _marks(N, 4096):
        assert np.array_equal(mag[s:s + 4096].cpu().numpy(), oracle.abs_(_window(oracle, s, 4096), True)), s


def test_fir_255_taps_over_more_than_2_32_samples(oracle, dev, big):
    torch, d, x, y = big
    from pothoscomms_amd import taps as tp
    h = tp.c1_taps()
    K = len(h)
    f = dev.FirFilter("complex_float32", "COMPLEX"); f.set_taps(h)
    assert f.process_dev(x, y, N + K - 1, N) == (N, N)
    torch.cuda.synchronize()
    ref = oracle.Fir(oracle.F32, True, True); ref.set_taps(h); ref.activate()
    S = 4096 - 256
    marks = _marks(N, 2048) + [(1 << 32) // S * S - 1024, ((1 << 31) // S + 1) * S - 1024]    # and block seams next to the marks
    for s in marks:
        want, _, p, _ = ref.work(_window(oracle, s, 2048 + K - 1), 2048)
        assert p == 2048
        assert nerr(y[s:s + 2048].cpu().numpy(), want) <= TOL, s


def test_fft_4096_over_more_than_2_20_frames(oracle, dev, big):
    torch, d, x, y = big
    nframes = (1 << 20) + 17                              # 2^32 + 69632 samples
    t = dev.Fft("complex_float32", 4096, False)
    t.transform_dev(x, y, nframes)
    torch.cuda.synchronize()
    for fr in (0, (1 << 17) - 1, 1 << 17, (1 << 19) - 1, 1 << 19, (1 << 20) - 1, 1 << 20, nframes - 1):
        got = y[fr * 4096:(fr + 1) * 4096].cpu().numpy()
        assert nerr(got, oracle.fft(_window(oracle, fr * 4096, 4096), 4096, False)) <= TOL, fr


def test_freq_demod_and_fm_chain_over_more_than_2_32_samples(oracle, dev, big):
    torch, d, x, y = big
    out = y.view(-1)[:N]
    dm = dev.FreqDemod("complex_float32")
    dm.process_dev(x, out, N)
    torch.cuda.synchronize()
    for s in _marks(N, 4096):
        first = max(s - 1, 0)
        want = oracle.FreqDemod(oracle.F32).work(_window(oracle, first, s - first + 4096))[s - first:]
        got = out[s:s + 4096].cpu().numpy()
        if s == 0:
            assert got[0] == 0.0
        assert ang_err(got, want) <= TOL, s
    from pothoscomms_amd import taps as tp
    h, phase = tp.c4_taps(), tp.C4_PHASE
    K = len(h)
    ch = dev.FmChain(); ch.set_phase(phase); ch.set_taps(h, False)
    assert ch.process_dev(x, out, N + K - 1, N) == (N, N)
    torch.cuda.synchronize()
    for s in _marks(N, 2048):
        first = max(s - 1, 0)
        m = s - first + 2048
        fir = oracle.Fir(oracle.F32, True, False); fir.set_taps(h); fir.activate()
        yy, _, p, _ = fir.work(oracle.rotate(_window(oracle, first, m + K - 1), phase), m)
        assert p == m
        want = oracle.FreqDemod(oracle.F32).work(yy)[s - first:]
        # uniform noise through a narrow low-pass: small |y| now and then, where the angle is ill-conditioned -- compare
        # where the FIR output is not tiny (as tests/test_fuzz_gpu.py does for the guard-band case)
        mag = np.hypot(yy[s - first:, 0], yy[s - first:, 1])
        ok = mag > 3e-2 * mag.max()
        assert ang_err(out[s:s + 2048].cpu().numpy()[ok], want[ok]) <= 2 * TOL, s


def test_long_tap_plan_and_resamplers_over_more_than_2_32_samples(oracle, dev, big):
    torch, d, x, y = big
    from pothoscomms_amd import taps as tp
    # 3000 taps: the radix-16 family's 8192-sample overlap-save plan
    h = tp.complex_bandpass(3000, 0.1, 0.03)
    K = len(h)
    f = dev.FirFilter("complex_float32", "COMPLEX"); f.set_taps(h)
    assert f.process_dev(x, y, N + K - 1, N) == (N, N)
    torch.cuda.synchronize()
    ref = oracle.Fir(oracle.F32, True, True); ref.set_taps(h); ref.activate()
    for s in _marks(N, 512):
        want, _, p, _ = ref.work(_window(oracle, s, 512 + K - 1), 512)
        assert p == 512 and nerr(y[s:s + 512].cpu().numpy(), want) <= TOL, s
    # decimation by 8 folded into the spectrum: N inputs -> N/8 outputs
    h = tp.complex_bandpass(255, 0.05 / 8, 0.05 / 8)
    K = len(h)
    f = dev.FirFilter("complex_float32", "COMPLEX"); f.set_taps(h); f.set_decimation(8)
    n_in = (N + K - 1) // 8 * 8
    c, p = f.process_dev(x, y, n_in, N)
    torch.cuda.synchronize()
    assert p == c // 8 and p >= (1 << 29)
    ref = oracle.Fir(oracle.F32, True, True); ref.set_taps(h); ref.set_decimation(8); ref.activate()
    for s in _marks(p, 256):                               # output index s reads inputs from 8 s
        want, _, rp, _ = ref.work(_window(oracle, 8 * s, 8 * 256 + K - 1), 256)
        assert rp == 256 and nerr(y[s:s + 256].cpu().numpy(), want) <= TOL, s
    # interpolation by 4: N/4 inputs -> more than 2^32 outputs
    h = tp.complex_bandpass(255 * 4, 0.05 / 4, 0.05 / 4) * 4
    f = dev.FirFilter("complex_float32", "COMPLEX"); f.set_taps(h); f.set_interpolation(4)
    Kp = f.K
    n_in = N // 4 + Kp
    c, p = f.process_dev(x, y, n_in, N)
    torch.cuda.synchronize()
    assert p == 4 * c and p > (1 << 32)
    ref = oracle.Fir(oracle.F32, True, True); ref.set_taps(h); ref.set_interpolation(4); ref.activate()
    for s in _marks(p // 4, 256):                          # input index s yields outputs from 4 s
        want, _, rp, _ = ref.work(_window(oracle, s, 256 + Kp - 1), 4 * 256)
        assert rp == 4 * 256 and nerr(y[4 * s:4 * s + 4 * 256].cpu().numpy(), want) <= TOL, s


def test_other_fft_plans_over_more_than_2_32_samples(oracle, dev, big):
    torch, d, x, y = big
    for nbins, nframes in ((1024, (1 << 22) + 5), (16384, (1 << 18) + 3), (1000, (1 << 32) // 1000 + 60)):
        assert nbins * nframes <= N and nbins * nframes > (1 << 32)
        t = dev.Fft("complex_float32", nbins, False)
        t.transform_dev(x, y, nframes)
        torch.cuda.synchronize()
        per = (1 << 32) // nbins
        for fr in (0, per // 8 - 1, per // 8, per // 2, per - 1, per, nframes - 1):
            got = y[fr * nbins:(fr + 1) * nbins].cpu().numpy()
            assert nerr(got, oracle.fft(_window(oracle, fr * nbins, nbins), nbins, False)) <= TOL, (nbins, fr)


def test_complex_int16_fir_over_more_than_2_32_samples(oracle, dev, big):
    torch, d, x, y = big
    from pothoscomms_amd import taps as tp
    xi = torch.empty((N + 256, 2), dtype=torch.int16, device=d)
    step = 1 << 28
    for a in range(0, N + 256, step):                      # float stream -> int16, in slices (no 34 GB temporary)
        b = min(N + 256, a + step)
        xi[a:b] = (x[a:b] * 3000.0).to(torch.int16)
    yi = torch.empty((N, 2), dtype=torch.int16, device=d)
    h = tp.c1_taps() * 0.9
    K = len(h)
    f = dev.FirFilter("complex_int16", "COMPLEX"); f.set_taps(h)
    assert f.process_dev(xi, yi, N + K - 1, N) == (N, N)
    torch.cuda.synchronize()
    ref = oracle.Fir(oracle.I16, True, True); ref.set_taps(h); ref.activate()
    for s in _marks(N, 2048):
        win = (_window(oracle, s, 2048 + K - 1) * np.float32(3000.0)).astype(np.int16)
        assert np.array_equal(xi[s:s + 2048 + K - 1].cpu().numpy(), win), s
        want, _, p, _ = ref.work(win, 2048)
        assert p == 2048 and np.array_equal(yi[s:s + 2048].cpu().numpy(), want), s
    del xi, yi


def test_remaining_maps_over_more_than_2_32_samples(oracle, dev, big):
    """scale, angle, arithmetic (complex MUL of the stream with itself one sample on), split and combine"""
    torch, d, x, y = big
    W = 4096
    dev.scale(x, 0.37, True, scalar=oracle.F32, out=y, n=N)
    torch.cuda.synchronize()
    for s in _marks(N, W):
        assert np.array_equal(y[s:s + W].cpu().numpy(), oracle.scale(_window(oracle, s, W), 0.37, True)), s
    plane = y.view(-1)
    dev.angle(x, scalar=oracle.F32, out=plane[:N], n=N)
    torch.cuda.synchronize()
    for s in _marks(N, W):
        assert ang_err(plane[s:s + W].cpu().numpy(), oracle.angle(_window(oracle, s, W))) <= TOL, s
    dev.arith("MUL", x, x[1:], True, scalar=oracle.F32, out=y, n=N)
    torch.cuda.synchronize()
    for s in _marks(N, W):
        a, b = _window(oracle, s, W), _window(oracle, s + 1, W)
        assert np.array_equal(y[s:s + W].cpu().numpy().view(np.uint8), oracle.arith(oracle.MUL, a, b, True).view(np.uint8)), s
    re, im = plane[:N], plane[N:2 * N]
    dev.split_complex(x, scalar=oracle.F32, re=re, im=im, n=N)
    torch.cuda.synchronize()
    for s in _marks(N, W):
        w = _window(oracle, s, W)
        assert np.array_equal(re[s:s + W].cpu().numpy(), w[:, 0]) and np.array_equal(im[s:s + W].cpu().numpy(), w[:, 1]), s
    z = torch.empty((N, 2), dtype=torch.float32, device=d)
    dev.combine_complex(re, im, scalar=oracle.F32, out=z, n=N)
    torch.cuda.synchronize()
    for s in _marks(N, W):
        assert np.array_equal(z[s:s + W].cpu().numpy(), _window(oracle, s, W)), s
    del z


def test_time_domain_fir_kernels_over_more_than_2_32_samples(oracle, dev, big):
    """the LDS-tiled direct kernel (63 taps) and the reference-order sliding-window kernel (EXACT, 31 taps: bit-identical)"""
    torch, d, x, y = big
    from pothoscomms_amd import _lib, taps as tp
    for algo, ntaps in ((_lib.FIR_DIRECT, 63), (_lib.FIR_EXACT, 31)):
        h = tp.complex_bandpass(ntaps, 0.1, 0.05)
        f = dev.FirFilter("complex_float32", "COMPLEX"); f.set_taps(h); f.set_algo(algo)
        assert f.process_dev(x, y, N + ntaps - 1, N) == (N, N)
        torch.cuda.synchronize()
        assert f.last_algo == algo
        ref = oracle.Fir(oracle.F32, True, True); ref.set_taps(h); ref.activate()
        for s in _marks(N, 4096):
            want, _, p, _ = ref.work(_window(oracle, s, 4096 + ntaps - 1), 4096)
            got = y[s:s + 4096].cpu().numpy()
            assert np.array_equal(got, want) if algo == _lib.FIR_EXACT else nerr(got, want) <= TOL, (algo, s)


def test_four_step_and_chirp_z_fft_over_more_than_2_32_samples(oracle, dev, big):
    """65536 bins (the large power-of-two plan) and 2 x 10243 bins (chirp-z: no other float plan)"""
    torch, d, x, y = big
    for nbins in (65536, 1 << 20, 2 * 10243):
        nframes = (1 << 32) // nbins + 1
        assert nbins * nframes > (1 << 32) and nbins * nframes <= N
        t = dev.Fft("complex_float32", nbins, False)
        t.transform_dev(x, y, nframes)
        torch.cuda.synchronize()
        per8 = (1 << 29) // nbins
        for fr in sorted(set([0, per8, 4 * per8, nframes - 2, nframes - 1])):
            got = y[fr * nbins:(fr + 1) * nbins].cpu().numpy()
            assert nerr(got, oracle.fft(_window(oracle, fr * nbins, nbins), nbins, False)) <= TOL, (nbins, fr)


def test_batched_workspace_paths_over_more_than_2_32_samples(oracle, dev, big):
    """the paths that go through a workspace take a long call in batches (1 GiB of workspace whatever the call): interpolation
    by 3 (polyphase rows + interleave), the fused chain beyond 2048 taps (FIR, then FreqDemod with the carried state walking
    through the batches), complex_int16 interpolation by 3 on the double pipeline.  Windows sit on the batch seams too."""
    torch, d, x, y = big
    from pothoscomms_amd import taps as tp
    CAP = 1 << 30
    # complex_float32, L = 3
    L = 3
    h = tp.complex_bandpass(255 * L, 0.05 / L, 0.05 / L) * L
    f = dev.FirFilter("complex_float32", "COMPLEX"); f.set_taps(h); f.set_interpolation(L)
    Kp = f.K
    n_in = N // L + Kp - 1
    c, p = f.process_dev(x, y, n_in, N)
    torch.cuda.synchronize()
    assert p == L * c and p > (1 << 32)
    ref = oracle.Fir(oracle.F32, True, True); ref.set_taps(h); ref.set_interpolation(L); ref.activate()
    nb = CAP // (L * 8)
    for s in _marks(c, 256) + [nb - 128, 7 * nb - 128, 31 * nb - 128]:
        want, _, rp, _ = ref.work(_window(oracle, s, 256 + Kp - 1), L * 256)
        assert rp == L * 256 and nerr(y[L * s:L * s + L * 256].cpu().numpy(), want) <= TOL, s
    # fused chain, 3000 taps: the unfused long-filter path
    h = tp.lowpass(3000, 0.1)
    K, phase = len(h), 0.4
    out = y.view(-1)[:N]
    ch = dev.FmChain(); ch.set_phase(phase); ch.set_taps(h, False)
    assert ch.process_dev(x, out, N + K - 1, N) == (N, N)
    torch.cuda.synchronize()
    nb = CAP // 8
    for s in _marks(N, 512) + [nb - 256, 17 * nb - 256]:
        first = max(s - 1, 0)
        m = s - first + 512
        fir = oracle.Fir(oracle.F32, True, False); fir.set_taps(h); fir.activate()
        yy, _, rp, _ = fir.work(oracle.rotate(_window(oracle, first, m + K - 1), phase), m)
        want = oracle.FreqDemod(oracle.F32).work(yy)[s - first:]
        mag = np.hypot(yy[s - first:, 0], yy[s - first:, 1])
        ok = mag > 3e-2 * mag.max()
        assert ang_err(out[s:s + 512].cpu().numpy()[ok], want[ok]) <= 2 * TOL, s
    # complex_int16, L = 3: rows on the double pipeline
    n16 = N // L + 1024
    xi = torch.empty((n16, 2), dtype=torch.int16, device=d)
    step = 1 << 28
    for a in range(0, n16, step):
        b = min(n16, a + step)
        xi[a:b] = (x[a:b] * 3000.0).to(torch.int16)
    yi = y.view(torch.int16).view(-1, 2)[:N]
    h = tp.complex_bandpass(63 * L, 0.1 / L, 0.05 / L) * L * 0.9
    f = dev.FirFilter("complex_int16", "COMPLEX"); f.set_taps(h); f.set_interpolation(L)
    Kp = f.K
    c, p = f.process_dev(xi, yi, N // L + Kp - 1, N)
    torch.cuda.synchronize()
    assert p == L * c and p > (1 << 32)
    ref = oracle.Fir(oracle.I16, True, True); ref.set_taps(h); ref.set_interpolation(L); ref.activate()
    nb = CAP // (L * 4)
    for s in _marks(c, 256) + [nb - 128, 5 * nb - 128]:
        win = (_window(oracle, s, 256 + Kp - 1) * np.float32(3000.0)).astype(np.int16)
        want, _, rp, _ = ref.work(win, L * 256)
        assert rp == L * 256 and np.array_equal(yi[L * s:L * s + L * 256].cpu().numpy(), want), s
    del xi
