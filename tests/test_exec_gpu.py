"""GPU suite: the execution model of the C ABI (pcx_host.hpp ExecCtx, include/pcx.h "Conventions") -- what round 1's review
found missing: page-locked host buffers processed in place, calls of one handle on DIFFERENT streams ordered behind each
other (carried state, tables), setters between asynchronous calls, reset enqueued, two handles on two threads."""
import ctypes as C
import threading

import numpy as np
import pytest

from tests.util import TOL, ang_err, nerr

pytestmark = pytest.mark.gpu


def _pinned(shape, dtype):
    from pothoscomms_amd import _lib
    L = _lib.load()
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = C.c_void_p()
    _lib.check(L.pcx_host_alloc(C.byref(p), nbytes))
    arr = np.ctypeslib.as_array((C.c_char * nbytes).from_address(p.value)).view(dtype).reshape(shape)
    return arr, p


def _free(p):
    from pothoscomms_amd import _lib
    _lib.check(_lib.load().pcx_host_free(p))


def test_page_locked_buffers_are_processed_in_place(oracle, dev):
    """FIR, FFT, FreqDemod, the fused chain and the maps on pcx_host_alloc memory -- including pointers INTO an allocation
    (a port buffer is a window of a slab) and a pinned input with a pageable output"""
    from pothoscomms_amd import taps as tp
    rng = np.random.default_rng(5)
    h = tp.c1_taps()
    K, n = len(h), 70001
    slab, ps = _pinned((4096 + n + K - 1 + 100, 2), np.float32)
    out, po = _pinned((n + 300, 2), np.float32)
    try:
        x = slab[4096 + 37:4096 + 37 + n + K - 1]          # an odd offset inside the slab
        x[:] = rng.uniform(-1, 1, x.shape).astype(np.float32)
        y = out[123:123 + n]
        out[:] = np.nan
        f = dev.FirFilter("complex_float32", "COMPLEX"); f.set_taps(h)
        from pothoscomms_amd import _lib
        c, p = C.c_size_t(), C.c_size_t()
        _lib.check(_lib.load().pcx_fir_process(f._h, x.ctypes.data, n + K - 1, y.ctypes.data, n, C.byref(c), C.byref(p)))
        ref_blk = oracle.Fir(oracle.F32, True, True); ref_blk.set_taps(h); ref_blk.activate()
        ref, _, rp, _ = ref_blk.work(x.copy(), n)
        assert (c.value, p.value) == (n, n) and nerr(y, ref) <= TOL
        assert np.isnan(out[:123]).all() and np.isnan(out[123 + n:]).all()      # nothing outside the window was written
        # pinned in, pageable out (and the reverse)
        got, _, _ = f.process(x, n)
        assert nerr(got, ref) <= TOL
        y[:] = 0
        xc = x.copy()      # (kept alive across the call: .ctypes.data of a temporary dangles)
        _lib.check(_lib.load().pcx_fir_process(f._h, xc.ctypes.data, n + K - 1, y.ctypes.data, n, C.byref(c), C.byref(p)))
        assert nerr(y, ref) <= TOL
        # maps in place on pinned memory (out == in is part of the contract)
        z = x[:50000]
        want = oracle.conj(z.copy())
        _lib.check(_lib.load().pcx_conj(_lib.F32, z.ctypes.data, z.ctypes.data, 50000))
        assert np.array_equal(z, want)
        # FFT and FreqDemod
        xf = slab[:4096 * 3]
        xf[:] = rng.uniform(-1, 1, xf.shape).astype(np.float32)
        yf = out[:4096 * 3]
        t = dev.Fft("complex_float32", 4096, False)
        _lib.check(_lib.load().pcx_fft_transform(t._h, xf.ctypes.data, yf.ctypes.data, 3))
        assert nerr(yf, oracle.fft(xf.copy(), 4096, False)) <= TOL
    finally:
        _free(ps); _free(po)


def test_calls_on_different_streams_are_ordered_behind_each_other(oracle, dev):
    """FreqDemod carries prev on the device: chunk i on stream A, chunk i+1 on stream B, no host synchronisation between --
    the result must be the single-stream one.  Same for the fused chain, and for a FIR whose taps change between calls."""
    import torch
    d = torch.device("cuda", 0)
    sA, sB = torch.cuda.Stream(d), torch.cuda.Stream(d)
    rng = np.random.default_rng(9)
    n = 1 << 20
    ph = np.cumsum(rng.uniform(-1.0, 1.0, n))
    xh = np.stack([np.cos(ph), np.sin(ph)], 1).astype(np.float32)
    x = torch.from_numpy(xh).to(d)
    y = torch.empty(n, dtype=torch.float32, device=d)
    torch.cuda.synchronize()
    dm = dev.FreqDemod("complex_float32")
    cuts = [0, 1000, 1001, 300000, 300017, 800000, n]
    for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
        dm.process_dev(x[a:b], y[a:b], b - a, stream=(sA, sB)[i & 1])
    torch.cuda.synchronize()
    assert ang_err(y.cpu().numpy(), oracle.FreqDemod(oracle.F32).work(xh)) <= TOL
    # reset is enqueued behind the previous call and ahead of the next one, whatever their streams
    dm.process_dev(x[:5000], y[:5000], 5000, stream=sA)
    dm.reset()
    dm.process_dev(x[5000:9000], y[5000:9000], 4000, stream=sB)
    torch.cuda.synchronize()
    assert ang_err(y[5000:9000].cpu().numpy(), oracle.FreqDemod(oracle.F32).work(xh[5000:9000])) <= TOL
    # FIR: new taps between two asynchronous calls on two streams -- the first call must finish with the OLD tables
    from pothoscomms_amd import taps as tp
    h1, h2 = tp.c1_taps(), tp.c0_taps()
    big = 8 << 20
    xb = torch.empty((big + 254, 2), dtype=torch.float32, device=d)
    dev.fill_uniform_f32_dev(xb, seed=3)
    y1 = torch.empty((big, 2), dtype=torch.float32, device=d)
    y2 = torch.empty((big, 2), dtype=torch.float32, device=d)
    torch.cuda.synchronize()
    f = dev.FirFilter("complex_float32", "COMPLEX"); f.set_taps(h1)
    f.process_dev(xb, y1, big + 254, big, stream=sA)
    f.set_taps(h2)                                    # host returns at once; the kernel above is still running
    f.process_dev(xb[254 - 62:], y2, big + 62, big, stream=sB)
    torch.cuda.synchronize()
    xw = xb[:70000 + 254].cpu().numpy()
    r1 = oracle.Fir(oracle.F32, True, True); r1.set_taps(h1); r1.activate()
    r2 = oracle.Fir(oracle.F32, True, True); r2.set_taps(h2); r2.activate()
    assert nerr(y1[:70000].cpu().numpy(), r1.work(xw, 70000)[0]) <= TOL
    assert nerr(y2[:70000].cpu().numpy(), r2.work(xw[254 - 62:], 70000)[0]) <= TOL
    # ... and its tail, which the first launch reaches last
    xt = xb[big - 70000:].cpu().numpy()
    assert nerr(y1[big - 70000:].cpu().numpy(), r1.work(xt, 70000)[0]) <= TOL


def test_two_blocks_on_two_threads(oracle, dev):
    """two handles, two host threads, host buffers: each handle runs on its own stream; results are each block's own"""
    from pothoscomms_amd import taps as tp
    rng = np.random.default_rng(3)
    hs = [tp.c1_taps(), tp.c0_taps()]
    xs = [rng.uniform(-1, 1, (300000 + len(h) - 1, 2)).astype(np.float32) for h in hs]
    outs = [None, None]

    def actor(i):
        f = dev.FirFilter("complex_float32", "COMPLEX"); f.set_taps(hs[i])
        for _ in range(5):
            outs[i] = f.process(xs[i], 300000)[0]
    th = [threading.Thread(target=actor, args=(i,)) for i in range(2)]
    [t.start() for t in th]; [t.join() for t in th]
    for i in range(2):
        r = oracle.Fir(oracle.F32, True, True); r.set_taps(hs[i]); r.activate()
        assert nerr(outs[i], r.work(xs[i], 300000)[0]) <= TOL


def test_the_library_ignores_the_environment(oracle):
    """round 1 shipped timing-only kernel variants behind PCX_OLS_VARIANT: the product library must not read it"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import numpy as np\nfrom oracle import oracle as o\nfrom pothoscomms_amd import device, taps as tp\n"
            "h = tp.c1_taps(); x = np.random.default_rng(0).uniform(-1, 1, (90000 + 254, 2)).astype(np.float32)\n"
            "f = device.FirFilter('complex_float32', 'COMPLEX'); f.set_taps(h); got = f.process(x, 90000)[0]\n"
            "r = o.Fir(o.F32, True, True); r.set_taps(h); r.activate(); ref = r.work(x, 90000)[0]\n"
            "assert np.max(np.abs(got - ref)) / np.max(np.abs(ref)) <= 1e-5\nprint('ok')\n")
    env = dict(os.environ, PCX_OLS_VARIANT="11", PCX_OLS_DIAG="2", PCX_FFT_MIXED_DIAG="1", PCX_OLS_SLOTS="7", PCX_OLS_ALIGN="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


def test_process_dev_can_be_captured_into_a_graph(oracle, dev):
    """include/pcx.h: *_dev calls only enqueue (once the tables are uploaded), so a steady-state loop can be captured into a
    hipGraph and replayed -- the dealt kernels reset their own block counters at the end of every launch."""
    import torch
    from pothoscomms_amd import taps as tp
    d = torch.device("cuda", 0)
    h = tp.c1_taps()
    K, n = len(h), 8 << 20
    x = torch.empty((n + K - 1, 2), dtype=torch.float32, device=d)
    dev.fill_uniform_f32_dev(x, seed=4)
    y = torch.empty((n, 2), dtype=torch.float32, device=d)
    f = dev.FirFilter("complex_float32", "COMPLEX"); f.set_taps(h)
    s = torch.cuda.Stream(d)
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        f.process_dev(x, y)                 # first call on this stream: uploads the tables (not capturable), binds the stream
    torch.cuda.synchronize()
    want = y.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        f.process_dev(x, y)
        f.process_dev(x, y)
    for _ in range(5):
        y.fill_(float("nan"))
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(y, want)
    ref = oracle.Fir(oracle.F32, True, True); ref.set_taps(h); ref.activate()
    xw = x[:60000 + K - 1].cpu().numpy()
    assert nerr(y[:60000].cpu().numpy(), ref.work(xw, 60000)[0]) <= TOL
    # the other launch shapes: decimating and interpolating filters (batched stages), the FFT, the fused chain with its carried state
    fd = dev.FirFilter("complex_float32", "COMPLEX"); fd.set_taps(h); fd.set_decimation(8)
    fi = dev.FirFilter("complex_float32", "COMPLEX"); fi.set_taps(tp.complex_bandpass(255 * 8, 0.05 / 8, 0.05 / 8) * 8); fi.set_interpolation(8)
    t = dev.Fft("complex_float32", 4096, False)
    ch = dev.FmChain(); ch.set_phase(0.3); ch.set_taps(tp.c4_taps(), False)
    nd = (n // 8) * 8
    yd = torch.empty((nd // 8, 2), dtype=torch.float32, device=d)
    ni = n // 8
    yi = torch.empty((ni * 8, 2), dtype=torch.float32, device=d)
    yf = torch.empty((n, 2), dtype=torch.float32, device=d)
    yc = torch.empty(n, dtype=torch.float32, device=d)
    Ki, Kc = fi.K, len(tp.c4_taps())

    def run():
        fd.process_dev(x, yd, nd + K - 1, nd // 8)
        fi.process_dev(x, yi, ni + Ki - 1, ni * 8)
        t.transform_dev(x, yf, n // 4096)
        ch.reset()
        ch.process_dev(x, yc, n + Kc - 1, n)
    with torch.cuda.stream(s):
        run()
    torch.cuda.synchronize()
    wants = [v.clone() for v in (yd, yi, yf, yc)]
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=s):
        run()
    for _ in range(3):
        for v in (yd, yi, yf, yc):
            v.fill_(float("nan"))
        g2.replay()
        torch.cuda.synchronize()
        for v, w in zip((yd, yi, yf, yc), wants):
            assert torch.equal(v, w)


def test_a_new_handle_is_ready_when_create_returns(oracle, dev):
    """State a handle zeroes at create (FreqDemod's _prev, FreqDemod.cpp:46; the block dealer's books) must be zero when
    create returns, not whenever a fill queued on the null stream gets its turn: the handle's own stream does not wait for
    the null stream.  A long fill keeps the null stream busy while handles are created on recycled memory and used at once."""
    import torch
    d = torch.device("cuda", 0)
    rng = np.random.default_rng(77)
    busy = torch.empty(1 << 28, dtype=torch.float32, device=d)      # 1 GiB
    x = (rng.standard_normal((3000, 2)) + 0.1).astype(np.float32)
    ref = oracle.FreqDemod(oracle.F32).work(x)
    for rep in range(12):
        warm = dev.FreqDemod("complex_float32")
        warm.process(x)                                              # leaves conj(x[-1]) where the next handle's state will live
        del warm
        for _ in range(8):
            busy.fill_(float(rep))                                   # torch's default stream is the null stream
        blk = dev.FreqDemod("complex_float32")
        got0 = blk.process(x[:1500])
        got1 = blk.process(x[1500:])
        assert got0[0] == 0.0, rep
        assert ang_err(np.concatenate([got0, got1]), ref) <= TOL, rep
    torch.cuda.synchronize()
    # the dealer's counter pair: a FIR handle created and run at once behind the same busy null stream
    from pothoscomms_amd import taps as tp
    t = tp.complex_bandpass(255, 0.1, 0.03)
    n = 4096 * 3000
    xs = (rng.standard_normal((n, 2))).astype(np.float32)
    xd = torch.from_numpy(xs).to(d)
    want = None
    for rep in range(4):
        # the output is poisoned and that fill drained BEFORE the null stream is made busy: the handle's stream does not wait for
        # the null stream, so a fill of yd queued behind the busy work would land on top of the filter's output
        yd = torch.full((n, 2), float("nan"), dtype=torch.float32, device=d)
        torch.cuda.synchronize()
        for _ in range(8):
            busy.fill_(1.0)
        f = dev.FirFilter("complex_float32", "COMPLEX"); f.set_taps(t)
        s = torch.cuda.Stream(device=d)
        c, p = f.process_dev(xd, yd, n, n, stream=s)
        s.synchronize()
        assert p == n - 254, (c, p)
        y = yd[:p].cpu().numpy()
        assert not np.isnan(y).any(), rep
        if want is None:
            want = y
        assert np.array_equal(y, want), rep


def test_captured_stateful_calls_reset_or_pairs(oracle, dev):
    """include/pcx.h: a captured reset() + call starts every replay from zero state; a captured PAIR of calls carries the state from
    one replay into the next (the handle alternates between two device slots, and a graph replays the slots it was captured with)"""
    import torch
    d = torch.device("cuda", 0)
    n = 1 << 16
    rng = np.random.default_rng(3)
    ph = np.cumsum(rng.uniform(-1.0, 1.0, 2 * n))
    xh = np.stack([np.cos(ph), np.sin(ph)], 1).astype(np.float32)
    x = torch.from_numpy(xh).to(d)
    y = torch.empty(2 * n, dtype=torch.float32, device=d)
    s = torch.cuda.Stream(d)
    # reset + one call: every replay is the stream from its start
    dm = dev.FreqDemod("complex_float32")
    with torch.cuda.stream(s):
        dm.reset(); dm.process_dev(x[:n], y[:n], n, stream=s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        dm.reset(); dm.process_dev(x[:n], y[:n], n, stream=s)
    want = oracle.FreqDemod(oracle.F32).work(xh[:n])
    for _ in range(4):
        y.fill_(float("nan"))
        junk = (x * 2).sum()                      # other work in between (a captured hipMemsetAsync did not survive this)
        g.replay(); torch.cuda.synchronize()
        assert ang_err(y[:n].cpu().numpy(), want) <= TOL
    # a pair of calls: replay r continues from replay r-1
    dm2 = dev.FreqDemod("complex_float32")
    with torch.cuda.stream(s):
        dm2.process_dev(x[:n], y[:n], n, stream=s); dm2.process_dev(x[n:], y[n:], n, stream=s)
    torch.cuda.synchronize()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=s):
        dm2.process_dev(x[:n], y[:n], n, stream=s); dm2.process_dev(x[n:], y[n:], n, stream=s)
    ref = oracle.FreqDemod(oracle.F32)
    ref.work(xh)                                    # the eager pass above
    for _ in range(3):
        want2 = ref.work(xh)                        # the reference keeps streaming: xh again, state carried
        y.fill_(float("nan"))
        g2.replay(); torch.cuda.synchronize()
        assert ang_err(y.cpu().numpy(), want2) <= TOL


def test_a_handle_can_be_destroyed_with_its_work_still_in_flight(oracle, dev):
    """*_dev calls only enqueue; destroying the handle right behind them must not pull the tables from under the running kernel
    (hipFree waits for the device) -- a Pothos topology tears blocks down without draining the device first"""
    import torch
    from pothoscomms_amd import taps as tp
    d = torch.device("cuda", 0)
    h = tp.c1_taps()
    K, n = len(h), 32 << 20
    x = torch.empty((n + K - 1, 2), dtype=torch.float32, device=d)
    dev.fill_uniform_f32_dev(x, seed=8)
    y = torch.full((n, 2), float("nan"), dtype=torch.float32, device=d)
    yc = torch.full((n,), float("nan"), dtype=torch.float32, device=d)
    s = torch.cuda.Stream(d)
    torch.cuda.synchronize()
    for _ in range(3):
        f = dev.FirFilter("complex_float32", "COMPLEX"); f.set_taps(h)
        ch = dev.FmChain(); ch.set_phase(0.2); ch.set_taps(tp.c4_taps(), False)
        f.process_dev(x, y, n + K - 1, n, stream=s)
        ch.process_dev(x, yc, n + len(tp.c4_taps()) - 1, n, stream=s)
        del f, ch                                   # destroy with both kernels queued or running
    torch.cuda.synchronize()
    assert bool(torch.isfinite(y).all()) and bool(torch.isfinite(yc).all())
    ref = oracle.Fir(oracle.F32, True, True); ref.set_taps(h); ref.activate()
    for start in (0, n - 50000):
        xw = x[start:start + 50000 + K - 1].cpu().numpy()
        assert nerr(y[start:start + 50000].cpu().numpy(), ref.work(xw, 50000)[0]) <= TOL


def test_clock_probe_reports_a_plausible_shader_clock():
    """pcx_clock_probe_dev (a measurement aid of the C ABI): one wave spins for spin_us and reports shader cycles per 100 MHz tick in
    MHz -- on an idle device the boost clock, beside a running workload the clock the power cap leaves it (bench.py roofline.valu)"""
    import torch

    from pothoscomms_amd import _lib, device
    out = torch.zeros((4,), dtype=torch.float32, device="cuda:0")
    side = torch.cuda.Stream()
    device.clock_probe(out, 200, side)
    torch.cuda.synchronize()
    mhz = float(out[0].item())
    assert 500.0 < mhz < 3500.0, mhz
    with pytest.raises(_lib.InvalidArgument):
        device.clock_probe(out, 0, side)
