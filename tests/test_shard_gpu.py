"""GPU suite: the NATIVE multi-device driver of include/pcx.h (pcx_shard_*): one process, per-device FIR handles and
streams, the K-1 halo moved by RCCL send/recv -- or by peer copies, which lets this one-GPU box run several shards.
The concatenated shard outputs must equal the oracle's single-stream result (no seam), and a one-device RCCL run must
be bit-identical to pcx_fir_process_dev on the whole stream."""
import ctypes as C

import numpy as np
import pytest

from tests.util import TOL, nerr

pytestmark = pytest.mark.gpu


def _oracle_fir(o, taps, x, n):
    blk = o.Fir(o.F32, True, True)
    blk.set_taps(taps)
    blk.activate()
    ref, c, p, _ = blk.work(x, n)
    assert p == n
    return ref


def _poison_halos(ns):
    """NaN into every halo slot but shard 0's: only a working exchange can make the result right."""
    from pothoscomms_amd import _lib
    L = _lib.load()
    nan = np.full((ns.K - 1, 2), np.nan, np.float32)
    for g in range(1, ns.nshards):
        i, _, s, _ = ns.buffers(g)
        _lib.check(L.pcx_memcpy_h2d(C.c_void_p(i), nan.ctypes.data_as(C.c_void_p), nan.nbytes, C.c_void_p(s)))
        _lib.check(L.pcx_stream_sync(C.c_void_p(s)))


@pytest.mark.parametrize("G,Cs", [(2, 40000), (4, 9000), (3, 3000), (2, 254)])
def test_peer_copy_shards_on_one_device_have_no_seam(oracle, G, Cs):
    from pothoscomms_amd import device, taps as tp
    h = tp.c1_taps()
    K = len(h)
    x = oracle.fill_uniform_f32(2 * (K - 1 + G * Cs), 2, 0).reshape(-1, 2)
    ns = device.NodeStream([0] * G, device.NodeStream.PEER_COPY)
    ns.set_taps(h)
    ns.configure(Cs)
    ns.scatter(x)
    for rep in range(2):          # the second pass re-receives into a halo slot the first pass's head kernel read
        _poison_halos(ns)
        ns.step()
        got = ns.gather()
        ref = _oracle_fir(oracle, h, x, G * Cs)
        assert np.isfinite(got).all()
        assert nerr(got, ref) <= TOL


def test_rccl_one_device_equals_the_plain_call(oracle):
    """RCCL transport with a communicator of one: exercises the rccl.h path (dlopen, ncclCommInitAll, destroy) on this
    box; the pass itself must be bit-identical to pcx_fir_process_dev on the whole stream."""
    import torch

    from pothoscomms_amd import device, taps as tp
    h = tp.c1_taps()
    K, n = len(h), 100000
    x = oracle.fill_uniform_f32(2 * (K - 1 + n), 2, 0).reshape(-1, 2)
    ns = device.NodeStream([0], device.NodeStream.RCCL)
    ns.set_taps(h)
    ns.configure(n)
    ns.scatter(x)
    ns.step()
    got = ns.gather()
    f = device.FirFilter("complex_float32", "COMPLEX")
    f.set_taps(h)
    # the same placement as the shard buffer: the samples (not the history) on a 128-byte line
    lead = (-(K - 1)) % 16
    xa = torch.zeros((lead + K - 1 + n, 2), dtype=torch.float32, device="cuda:0")
    xa[lead:] = torch.from_numpy(x).cuda()
    y = torch.empty((n, 2), dtype=torch.float32, device="cuda:0")
    c, p = f.process_dev(xa[lead:], y)
    assert (c, p) == (n, n)
    assert np.array_equal(got, y.cpu().numpy())
    assert nerr(got, _oracle_fir(oracle, h, x, n)) <= TOL


def test_rccl_refuses_two_shards_on_one_device():
    from pothoscomms_amd import _lib, device
    with pytest.raises(_lib.InvalidArgument):
        device.NodeStream([0, 0], device.NodeStream.RCCL)


def test_real_taps_and_retap_between_passes(oracle):
    """REAL taps go in as complex taps with zero imaginary parts; new taps of another length re-lay the buffers."""
    from pothoscomms_amd import _lib, device, taps as tp
    ns = device.NodeStream([0, 0], device.NodeStream.PEER_COPY)
    h = tp.c4_taps()
    ns.set_taps(h, complex_taps=False)
    ns.configure(20000)
    x = oracle.fill_uniform_f32(2 * (len(h) - 1 + 2 * 20000), 5, 0).reshape(-1, 2)
    ns.scatter(x)
    ns.step()
    blk = oracle.Fir(oracle.F32, True, False)
    blk.set_taps(h)
    blk.activate()
    ref, _, p, _ = blk.work(x, 2 * 20000)
    assert nerr(ns.gather(), ref) <= TOL
    ns.set_taps(tp.c0_taps())
    with pytest.raises(_lib.InvalidArgument):
        ns.step()                     # buffers were dropped with the old halo size: configure again
    ns.configure(5000)
    h0 = tp.c0_taps()
    x0 = oracle.fill_uniform_f32(2 * (len(h0) - 1 + 2 * 5000), 1, 0).reshape(-1, 2)
    ns.scatter(x0)
    ns.step()
    assert nerr(ns.gather(), _oracle_fir(oracle, h0, x0, 2 * 5000)) <= TOL


def _ngpus():
    import torch
    return torch.cuda.device_count()


@pytest.mark.skipif("_ngpus() < 2", reason="needs two GPUs")
def test_rccl_two_devices_no_seam(oracle):
    """the real thing on a multi-GPU node: one shard per device, halo over RCCL send/recv"""
    from pothoscomms_amd import device, taps as tp
    G = min(_ngpus(), 8)
    h = tp.c1_taps()
    K, Cs = len(h), 50000
    x = oracle.fill_uniform_f32(2 * (K - 1 + G * Cs), 2, 0).reshape(-1, 2)
    ns = device.NodeStream(list(range(G)), device.NodeStream.RCCL)
    ns.set_taps(h)
    ns.configure(Cs)
    ns.scatter(x)
    for _ in range(2):
        _poison_halos(ns)
        ns.step()
        got = ns.gather()
        assert np.isfinite(got).all()
        assert nerr(got, _oracle_fir(oracle, h, x, G * Cs)) <= TOL


@pytest.mark.skipif("_ngpus() < 2", reason="needs two GPUs")
def test_handles_stay_on_the_device_they_were_created_on(oracle):
    """create on device 1, call from a thread whose current device is 0 (a Pothos actor thread's default)"""
    import threading

    from pothoscomms_amd import _lib, device, taps as tp
    L = _lib.load()
    h = tp.c1_taps()
    x = oracle.fill_uniform_f32(2 * (len(h) - 1 + 30000), 3, 0).reshape(-1, 2)
    xf = oracle.fill_uniform_f32(2 * 4096 * 3, 4, 0).reshape(-1, 2)
    _lib.check(L.pcx_set_device(1))
    try:
        f = device.FirFilter("complex_float32", "COMPLEX"); f.set_taps(h)
        t = device.Fft("complex_float32", 4096, False)
    finally:
        _lib.check(L.pcx_set_device(0))
    out = {}

    def actor():
        _lib.check(L.pcx_set_device(0))
        out["fir"] = f.process(x, 30000)[0]
        out["fft"] = t.transform(xf)
        out["conj"] = device.conj(x)
    th = threading.Thread(target=actor); th.start(); th.join()
    assert nerr(out["fir"], _oracle_fir(oracle, h, x, 30000)) <= TOL
    assert nerr(out["fft"], oracle.fft(xf, 4096, False)) <= TOL
    assert np.array_equal(out["conj"], oracle.conj(x))


def test_scatter_and_gather_on_page_locked_and_pageable_memory_agree(oracle):
    """page-locked caller memory goes to the DMA engine as it is (scatter returns before the copy has run: the buffer must
    stay alive until the pass is synchronised); pageable memory goes through the shard's bounce buffers"""
    from pothoscomms_amd import _lib, device, taps as tp
    h = tp.c1_taps()
    K, G, Cs = len(h), 3, 50000
    x = oracle.fill_uniform_f32(2 * (K - 1 + G * Cs), 2, 0).reshape(-1, 2)
    ns = device.NodeStream([0] * G, device.NodeStream.PEER_COPY)
    ns.set_taps(h)
    ns.configure(Cs)
    ns.scatter(x)
    ns.step()
    pageable = ns.gather()
    L = _lib.load()
    pin, pout = C.c_void_p(), C.c_void_p()
    _lib.check(L.pcx_host_alloc(C.byref(pin), x.nbytes))
    _lib.check(L.pcx_host_alloc(C.byref(pout), G * Cs * 8))
    try:
        xin = np.ctypeslib.as_array((C.c_float * x.size).from_address(pin.value)).reshape(x.shape)
        yout = np.ctypeslib.as_array((C.c_float * (G * Cs * 2)).from_address(pout.value)).reshape(-1, 2)
        xin[:] = x
        yout[:] = 0
        ns.scatter(xin)
        ns.step()
        got = ns.gather(out=yout)
        assert np.array_equal(got, pageable)
        assert nerr(got, _oracle_fir(oracle, h, x, G * Cs)) <= TOL
    finally:
        ns.sync()
        del ns
        _lib.check(L.pcx_host_free(pin)); _lib.check(L.pcx_host_free(pout))


def test_call_order_and_argument_errors():
    """every misuse of include/pcx.h's call order (create -> set_taps -> configure -> scatter/step/gather) and every bad argument is
    refused with PCX_ERR_ARG and a message; nothing is launched and the handle stays usable"""
    from pothoscomms_amd import _lib, device, taps as tp
    L = _lib.load()
    h = C.c_void_p()
    for n, devs, tr, frag in ((0, None, 1, "0 shards"), (65, None, 1, "65 shards"), (1, [0], 7, "unknown transport"), (1, [99], 1, "on device 99")):
        arr = (C.c_int * len(devs))(*devs) if devs else None
        assert L.pcx_shard_create(n, arr, tr, C.byref(h)) == _lib.ERR_ARG
        assert frag in L.pcx_last_error().decode(), (frag, L.pcx_last_error())
    ns = device.NodeStream([0, 0], device.NodeStream.PEER_COPY)
    x = np.zeros((1000, 2), np.float32)
    with pytest.raises(_lib.PcxError, match="set the taps first"):
        ns.configure(100)
    with pytest.raises(_lib.PcxError, match="configure first"):
        ns.step()
    with pytest.raises(_lib.PcxError, match="configure first"):
        ns.scatter(x)
    ns.set_taps(tp.c1_taps())
    with pytest.raises(_lib.PcxError, match="empty shard"):
        ns.configure(0)
    with pytest.raises(_lib.PcxError, match="shorter than the 254-sample halo"):
        ns.configure(100)
    ns.configure(400)
    with pytest.raises(_lib.PcxError, match="expected K-1 \\+ shards\\*C = 1054"):
        ns.scatter(x)
    y = np.zeros((801, 2), np.float32)
    assert L.pcx_shard_gather(ns._h, y.ctypes.data_as(C.c_void_p), 801) == _lib.ERR_ARG
    assert L.pcx_shard_step(None) == _lib.ERR_ARG and L.pcx_shard_destroy(None) == 0
    # ... and the handle still works
    xs = np.random.default_rng(0).standard_normal((254 + 800, 2)).astype(np.float32)
    ns.scatter(xs); ns.step()
    assert ns.gather().shape == (800, 2)


# ---- round 3: one launch per shard and pass (the gate), long filters, the fused chain ----

def test_the_gate_holds_the_first_block_until_it_is_signalled(oracle):
    """pcx_fir_process_dev_gated: ONE launch over a shard whose halo has not arrived.  The launch is queued with NaN in the
    halo slot and the gate closed; 50 ms later another stream writes the halo and signals.  Had block 0 run ahead of the gate the
    first outputs would be NaN; the result must be bit-identical to the plain call on the completed buffer."""
    import time

    import torch

    from pothoscomms_amd import device, taps as tp
    h = tp.c1_taps()
    K = len(h)
    n = 2100 * 3840                       # > 2048 blocks: the dealt kernel, which is the one that has the gate
    dev = torch.device("cuda", 0)
    lead = (-(K - 1)) % 16
    xa = torch.empty((lead + K - 1 + n, 2), dtype=torch.float32, device=dev)
    x = xa[lead:]
    device.fill_uniform_f32_dev(x, seed=9, offset=0)
    halo = x[:K - 1].clone()
    f = device.FirFilter("complex_float32", "COMPLEX")
    f.set_taps(h)
    want = torch.empty((n, 2), dtype=torch.float32, device=dev)
    assert f.process_dev(x, want) == (n, n)
    torch.cuda.synchronize()
    gate = torch.zeros((64,), dtype=torch.int32, device=dev)
    # the signal is queued BEHIND the launch here (that is the point of the test): its stream must not share the launch's hardware
    # queue, and a stream of another priority never does (include/pcx.h, the gate's contract; tools/gate_queue_probe.py)
    side = torch.cuda.Stream(device=dev, priority=-1)
    for value in (1, 2, 0x7FFFFFFF + 3):               # a pass counter, also across the sign bit (signed distance)
        if value > 2:
            gate.fill_((value - 1) - (1 << 32) if value - 1 >= (1 << 31) else value - 1)     # the word as int32
        x[:K - 1] = float("nan")
        got = torch.full((n, 2), float("nan"), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        c, p, gated = f.process_dev_gated(x, got, gate, value)
        assert gated and (c, p) == (n, n)
        time.sleep(0.05)                                # the launch is running (or done, all but block 0) by now
        with torch.cuda.stream(side):
            x[:K - 1] = halo
            device.gate_signal(gate, value, side.cuda_stream)
        torch.cuda.synchronize()
        assert torch.isfinite(got).all()
        assert torch.equal(got, want)
    # a short call has no gated kernel: nothing is queued, the caller is told
    c, p, gated = f.process_dev_gated(x[:K - 1 + 50000], got, gate, 5, out_cap=50000)
    assert not gated and (c, p) == (0, 0)
    ref = _oracle_fir(oracle, h, x[:K - 1 + 8192].cpu().numpy(), 8192)
    assert nerr(want[:8192].cpu().numpy(), ref) <= TOL


@pytest.mark.parametrize("G", [2, 3])
def test_long_shards_take_one_gated_launch_each_and_have_no_seam(oracle, G):
    """shards of more than 2048 blocks: pcx_shard_step queues ONE launch per shard; poisoned halos, two passes; every shard's output
    is bit-identical to a plain call on its completed buffer, and the seams agree with the oracle"""
    import torch

    from pothoscomms_amd import _lib, device, taps as tp
    L = _lib.load()
    h = tp.c1_taps()
    K, Cs = len(h), 2080 * 3840
    ns = device.NodeStream([0] * G, device.NodeStream.PEER_COPY)
    ns.set_taps(h)
    ns.configure(Cs)
    for g in range(G):
        i, o, s, d = ns.buffers(g)
        _lib.check(L.pcx_fill_uniform_f32_dev(C.c_void_p(i), 2 * (K - 1 + Cs), 4, 2 * g * Cs, C.c_void_p(s)))
    f = device.FirFilter("complex_float32", "COMPLEX")
    f.set_taps(h)
    for rep in range(2):
        _poison_halos(ns)
        ns.step()
        ns.sync()
        for g in range(G):
            i, o, s, d = ns.buffers(g)
            xin = np.empty((K - 1 + Cs, 2), np.float32)
            yout = np.empty((Cs, 2), np.float32)
            _lib.check(L.pcx_memcpy_d2h(xin.ctypes.data_as(C.c_void_p), C.c_void_p(i), xin.nbytes, None))
            _lib.check(L.pcx_memcpy_d2h(yout.ctypes.data_as(C.c_void_p), C.c_void_p(o), yout.nbytes, None))
            assert np.isfinite(yout).all(), "shard %d pass %d" % (g, rep)
            # the seam against the oracle, and the whole shard against a plain call on the same (completed) buffer
            assert nerr(yout[:6000], _oracle_fir(oracle, h, xin[:K - 1 + 6000], 6000)) <= TOL
            lead = (-(K - 1)) % 16
            xa = torch.zeros((lead + K - 1 + Cs, 2), dtype=torch.float32, device="cuda:0")
            xa[lead:] = torch.from_numpy(xin).cuda()
            y = torch.empty((Cs, 2), dtype=torch.float32, device="cuda:0")
            assert f.process_dev(xa[lead:], y) == (Cs, Cs)
            assert np.array_equal(yout, y.cpu().numpy()), "shard %d pass %d" % (g, rep)
            if g > 0:      # the halo really is the left neighbour's tail
                ip, _, _, _ = ns.buffers(g - 1)
                tail = np.empty((K - 1, 2), np.float32)
                _lib.check(L.pcx_memcpy_d2h(tail.ctypes.data_as(C.c_void_p), C.c_void_p(ip + 8 * Cs), tail.nbytes, None))
                assert np.array_equal(xin[:K - 1], tail)


@pytest.mark.parametrize("K", [5000, 9000])
def test_filters_longer_than_the_old_head_split(oracle, K):
    """the two-launch split used a head of 4096 outputs whatever K, so with K-1 > 4096 the body launch read
    halo samples that had not arrived.  Two shards, two passes with DIFFERENT data (a stale halo of pass 1 must show in pass 2)."""
    from pothoscomms_amd import device, taps as tp
    rng = np.random.default_rng(K)
    h = (rng.standard_normal(K) + 1j * rng.standard_normal(K)) / K
    G, Cs = 2, 20000
    ns = device.NodeStream([0] * G, device.NodeStream.PEER_COPY)
    ns.set_taps(h)
    ns.configure(Cs)
    for seed in (11, 12):
        x = oracle.fill_uniform_f32(2 * (K - 1 + G * Cs), seed, 0).reshape(-1, 2)
        ns.scatter(x)
        _poison_halos(ns)
        ns.step()
        got = ns.gather()
        assert np.isfinite(got).all()
        ref = _oracle_fir(oracle, h, x, G * Cs)
        # K = 9000 runs the sliding-window kernel in the reference's own order: bit-identical; K = 5000 the 16384-sample plan
        assert nerr(got, ref) <= (TOL if K <= 8193 else 0.0) + 0.0, nerr(got, ref)


def _oracle_chain(o, taps, phase, x, n):
    r = o.rotate(x, phase)
    blk = o.Fir(o.F32, True, False)
    blk.set_taps(taps)
    blk.activate()
    y, c, p, _ = blk.work(r, n)
    assert p == n
    return o.FreqDemod(o.F32).work(y)


@pytest.mark.parametrize("G,Cs", [(2, 30000), (3, 7000), (2, 2080 * 3968)])
def test_the_fused_chain_shards_behind_the_c_abi(oracle, G, Cs):
    """pcx_shard_set_chain: Rotate -> FIR(127 real taps) -> FreqDemod over G shards (halo of K samples, one extra output in front of
    every shard but the first, dropped).  Small shards run the ungated path, the long ones ONE gated launch per shard; the seams
    (windows around every shard boundary) and the stream start agree with the oracle's three blocks, the whole stream with the
    single-device fused kernel."""
    import torch

    from pothoscomms_amd import _lib, device, taps as tp
    from tests.util import ang_err
    L = _lib.load()
    h = tp.c4_taps()
    K = len(h)
    n = G * Cs
    ns = device.NodeStream([0] * G, device.NodeStream.PEER_COPY)
    ns.set_chain(True, tp.C4_PHASE)
    ns.set_taps(h, complex_taps=False)
    ns.configure(Cs)
    for g in range(G):
        i, o, s, d = ns.buffers(g)
        # shard g: its K-1 history and its samples, straight from the node-wide stream (the exchange must overwrite the history)
        _lib.check(L.pcx_fill_uniform_f32_dev(C.c_void_p(i), 2 * (K - 1 + Cs), 5, 2 * g * Cs, C.c_void_p(s)))
    for rep in range(2):
        for g in range(1, G):     # NaN into every halo slot: K-1 history samples + the demodulator's predecessor in front of them
            i, _, s, _ = ns.buffers(g)
            nan = np.full((K, 2), np.nan, np.float32)
            _lib.check(L.pcx_memcpy_h2d(C.c_void_p(i - 8), nan.ctypes.data_as(C.c_void_p), nan.nbytes, C.c_void_p(s)))
            _lib.check(L.pcx_stream_sync(C.c_void_p(s)))
        ns.step()
        got = ns.gather()
        assert got.shape == (n,) and np.isfinite(got).all()
        # the whole stream through ONE fused call on one device
        ch = device.FmChain(); ch.set_phase(tp.C4_PHASE); ch.set_taps(h, False)
        xa = torch.empty((2 + n + K - 1, 2), dtype=torch.float32, device="cuda:0")
        device.fill_uniform_f32_dev(xa[2:], seed=5, offset=0)
        y = torch.empty((n,), dtype=torch.float32, device="cuda:0")
        assert ch.process_dev(xa[2:], y, n + K - 1, n) == (n, n)
        # Two device results with different block boundaries, each carrying the transform's ~3e-7 max|y| of rounding in y: the angle
        # of a sample whose |y| is a thousandth of the largest moves by a thousand times that, and a stream of millions of random
        # samples holds such samples.  So: twice the bar on all but one sample in 1e5, and no sample off by more than 1e-2 rad
        # (a seam error is an error of order one); the seams themselves are held to the bar against the oracle below.
        dd = np.abs((got.astype(np.float64) - y.cpu().numpy() + np.pi) % (2 * np.pi) - np.pi)
        assert np.quantile(dd, 1 - 1e-5) / np.pi <= 2 * TOL and dd.max() < 1e-2, (np.quantile(dd, 1 - 1e-5), dd.max())
        # stream start and every seam against the oracle's three blocks
        xs = xa[2:].cpu().numpy()
        W = 3000
        ref0 = _oracle_chain(oracle, h, tp.C4_PHASE, xs[:K - 1 + W], W)
        assert ang_err(got[:W], ref0) <= TOL
        for g in range(1, G):
            a = g * Cs - W                      # outputs a .. a + 2W - 1 straddle the boundary; one more in front seeds the demodulator
            ref = _oracle_chain(oracle, h, tp.C4_PHASE, xs[a - 1:a - 1 + K - 1 + 2 * W + 1], 2 * W + 1)[1:]
            assert ang_err(got[a:a + 2 * W], ref) <= TOL, "seam %d" % g


def test_a_gate_that_is_never_signalled_times_out_instead_of_hanging():
    """the wait is bounded (two seconds): the held block then runs on whatever the halo slot holds and says so in gate[1]"""
    import time

    import torch

    from pothoscomms_amd import device, taps as tp
    h = tp.c1_taps()
    K, n = len(h), 2100 * 3840
    dev = torch.device("cuda", 0)
    x = torch.zeros((K - 1 + n, 2), dtype=torch.float32, device=dev)
    y = torch.empty((n, 2), dtype=torch.float32, device=dev)
    gate = torch.zeros((64,), dtype=torch.int32, device=dev)
    f = device.FirFilter("complex_float32", "COMPLEX")
    f.set_taps(h)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    c, p, gated = f.process_dev_gated(x, y, gate, 1)
    assert gated
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert 1.5 < dt < 10.0, dt
    assert int(gate[1].item()) == 0xDEAD and int(gate[0].item()) == 0
    # the ranks driver's check (stream.py check_gate, what bench.py calls behind its timed region) reads exactly this word: it raises
    # once and clears it
    import types

    from pothoscomms_amd import stream
    owner = types.SimpleNamespace(_gate=gate, _buf=x, _side=torch.cuda.Stream(device=dev), _pass=1, ring=types.SimpleNamespace(rank=1))
    with pytest.raises(stream.GateTimeout):
        stream._check_gate(owner)
    stream._check_gate(owner)
    assert int(gate[1].item()) == 0


_DROP_SCRIPT = r"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, %(root)r)
from pothoscomms_amd import _lib, device, taps as tp
L = _lib.load()
h = tp.c1_taps()
K, Cs, G = len(h), 2080 * 3840, 2
ns = device.NodeStream([0] * G, device.NodeStream.PEER_COPY)
ns.set_taps(h); ns.configure(Cs)
for g in range(G):
    i, o, s, d = ns.buffers(g)
    _lib.check(L.pcx_fill_uniform_f32_dev(C.c_void_p(i), 2 * (K - 1 + Cs), 4, 2 * g * Cs, C.c_void_p(s)))
out = np.empty((G * Cs, 2), np.float32)
res = []
for p in (1, 2, 3):
    ns.step()
    try:
        ns.gather(out) if p != 3 else ns.sync()
        res.append("ok")
    except _lib.PcxError as e:
        res.append("state" if e.status == _lib.ERR_STATE and "halo" in str(e) else "other: %%s" %% e)
    try:
        ns.sync()
        res.append("ok")
    except _lib.PcxError as e:
        res.append("again")
print("RESULT " + ",".join(res))
"""


def test_a_dropped_gate_signal_is_reported_by_gather_and_cleared():
    """a gated launch that gives up waiting for its halo (two seconds) ran its first block on stale data and only
    left 0xDEAD behind the gate word -- which pcx_shard_gather, the call FIRFilter::work() makes, never read.  The diagnostic library
    leaves out the signals of pass 2 (PCX_SHARD_DROP_SIGNAL): gather of pass 2 must fail with PCX_ERR_STATE, the condition must be
    cleared by having been reported (the sync behind it is clean), and pass 3 -- whose gate value is one further -- must be clean."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    diag = os.path.join(root, "pothoscomms_amd", "libpcx_hip_diag.so")
    if not os.path.exists(diag):
        pytest.skip("diagnostic library not built (make -C pothoscomms_amd/csrc diag)")
    env = dict(os.environ, PCX_HIP_LIBRARY=diag, PCX_SHARD_DROP_SIGNAL="2")
    r = subprocess.run([sys.executable, "-c", _DROP_SCRIPT % {"root": root}], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    assert line == "RESULT ok,ok,state,ok,ok,ok", line


@pytest.mark.parametrize("chain", [False, True])
def test_long_shards_with_the_gate_switched_off_take_two_launches_and_have_no_seam(oracle, chain):
    """pcx_shard_set_gated(s, 0): the documented way out for a process that must run under AMD_DIRECT_DISPATCH=0 (every stream submitted
    by a thread of its own: a signal can land behind the launch it is to release).  Shards long enough for the gated kernel, the gate
    off: body while the halo is in flight, head behind the halo event; poisoned halos, two passes; seams against the oracle, and the
    stream equal to the gated form's within the float bar (FIR) / the angle bar (chain)."""
    from pothoscomms_amd import _lib, device, taps as tp
    from tests.util import ang_err
    L = _lib.load()
    G = 3
    h = tp.c4_taps() if chain else tp.c1_taps()
    K, Cs = len(h), 2080 * (3968 if chain else 3840)
    outs = []
    for gated in (True, False):
        ns = device.NodeStream([0] * G, device.NodeStream.PEER_COPY)
        if chain:
            ns.set_chain(True, tp.C4_PHASE)
        ns.set_taps(h, complex_taps=not chain)
        ns.set_gated(gated)
        ns.configure(Cs)
        halo = K if chain else K - 1
        for rep in range(2):
            for g in range(G):
                i, o, s, d = ns.buffers(g)
                if chain:       # the FM signal, each shard at its own amplitude (tests/test_c3_gpu.py has the reasons)
                    xs = (tp.fm_test_signal(K - 1 + Cs, start=0).view(np.float32).reshape(-1, 2) * np.float32(1.0 + 0.05 * g + 0.01 * rep)).astype(np.float32)
                    _lib.check(L.pcx_memcpy_h2d(C.c_void_p(i), xs.ctypes.data_as(C.c_void_p), xs.nbytes, C.c_void_p(s)))
                else:
                    _lib.check(L.pcx_fill_uniform_f32_dev(C.c_void_p(i), 2 * (K - 1 + Cs), 4 + rep, 2 * g * Cs, C.c_void_p(s)))
                if g > 0:
                    nan = np.full((halo, 2), np.nan, np.float32)
                    _lib.check(L.pcx_memcpy_h2d(C.c_void_p(i - 8 * (halo - (K - 1))), nan.ctypes.data_as(C.c_void_p), nan.nbytes, C.c_void_p(s)))
                _lib.check(L.pcx_stream_sync(C.c_void_p(s)))
            ns.step()
            got = ns.gather()
            assert np.isfinite(got).all(), (gated, rep)
        outs.append(got)
        # every seam of the last pass against the oracle, inputs cut from the shards' own buffers
        W = 3000
        for g in range(1, G):
            il, _, _, _ = ns.buffers(g - 1)
            ir, _, _, _ = ns.buffers(g)
            extra = 1 if chain else 0
            left = np.empty((W + extra + K - 1, 2), np.float32)
            right = np.empty((W, 2), np.float32)
            _lib.check(L.pcx_memcpy_d2h(left.ctypes.data_as(C.c_void_p), C.c_void_p(il + 8 * (Cs - W - extra)), left.nbytes, None))
            _lib.check(L.pcx_memcpy_d2h(right.ctypes.data_as(C.c_void_p), C.c_void_p(ir + 8 * (K - 1)), right.nbytes, None))
            xin = np.concatenate([left, right])
            seam = got[g * Cs - W:g * Cs + W]
            if chain:
                assert ang_err(seam, _oracle_chain(oracle, h, tp.C4_PHASE, xin, 2 * W + 1)[1:]) <= TOL, (gated, g)
            else:
                assert nerr(seam, _oracle_fir(oracle, h, xin, 2 * W)) <= TOL, (gated, g)
        ns.close()
    if chain:
        dd = np.abs((outs[0].astype(np.float64) - outs[1] + np.pi) % (2 * np.pi) - np.pi)
        assert np.quantile(dd, 1 - 1e-5) / np.pi <= 2 * TOL and dd.max() < 1e-2
    else:
        assert nerr(outs[1], outs[0]) <= TOL


def test_two_handles_double_buffered_the_next_batchs_halos_travel_beside_this_batchs_pass(oracle):
    """pcx_shard_post_exchange / pcx_shard_compute (include/pcx.h): two handles of three long shards each on one device, five batches
    through them alternately -- compute(batch k) queued first, then post_exchange(batch k+1) -- every halo poisoned before its exchange.
    Every shard of every batch is bit-identical to a plain call on its completed buffer (the gated launch walks the same blocks), the
    halos are the left neighbours' tails, no gate timed out; and the call order is enforced."""
    import torch

    from pothoscomms_amd import _lib, device, taps as tp
    L = _lib.load()
    h = tp.c1_taps()
    K, Cs, G = len(h), 2080 * 3840, 3
    pair = []
    for _ in range(2):
        ns = device.NodeStream([0] * G, device.NodeStream.PEER_COPY)
        ns.set_taps(h)
        ns.configure(Cs)
        pair.append(ns)
    with pytest.raises(_lib.PcxError, match="no exchange posted"):
        pair[0].compute()
    f = device.FirFilter("complex_float32", "COMPLEX")
    f.set_taps(h)

    def load(ns, batch):
        for g in range(G):
            i, o, s, d = ns.buffers(g)
            _lib.check(L.pcx_fill_uniform_f32_dev(C.c_void_p(i), 2 * (K - 1 + Cs), 40 + batch, 2 * g * Cs, C.c_void_p(s)))
        _poison_halos(ns)

    load(pair[0], 0)
    pair[0].post_exchange()
    with pytest.raises(_lib.PcxError, match="already posted"):
        pair[0].post_exchange()
    with pytest.raises(_lib.PcxError, match="exchange is posted"):      # the exchange is reading the buffers a scatter would overwrite
        pair[0].scatter(np.zeros((K - 1 + G * Cs, 2), np.float32))
    with pytest.raises(_lib.PcxError, match="exchange is posted"):
        pair[0].configure(Cs)
    #  the setters that would change -- or, with another K / the chain switched, FREE -- what the posted pass is about to use
    with pytest.raises(_lib.PcxError, match="exchange is posted") as e:
        pair[0].set_taps(h[:31])
    assert e.value.status == _lib.ERR_STATE
    with pytest.raises(_lib.PcxError, match="exchange is posted"):
        pair[0].set_chain(True, 0.3)
    with pytest.raises(_lib.PcxError, match="exchange is posted"):
        pair[0].set_algo(_lib.FIR_DIRECT)
    assert pair[0].info()[1:3] == (K, Cs)                       # nothing was changed or freed by the refused calls
    sums = set()
    for k in range(5):
        cur, nxt = pair[k & 1], pair[(k + 1) & 1]
        load(nxt, k + 1)
        cur.compute()
        nxt.post_exchange()
        cur.sync()                                             # (reports a gate that timed out)
        for g in range(G):
            i, o, s, d = cur.buffers(g)
            xin = np.empty((K - 1 + Cs, 2), np.float32)
            yout = np.empty((Cs, 2), np.float32)
            _lib.check(L.pcx_memcpy_d2h(xin.ctypes.data_as(C.c_void_p), C.c_void_p(i), xin.nbytes, None))
            _lib.check(L.pcx_memcpy_d2h(yout.ctypes.data_as(C.c_void_p), C.c_void_p(o), yout.nbytes, None))
            assert np.isfinite(yout).all(), "batch %d shard %d" % (k, g)
            lead = (-(K - 1)) % 16
            xa = torch.zeros((lead + K - 1 + Cs, 2), dtype=torch.float32, device="cuda:0")
            xa[lead:] = torch.from_numpy(xin).cuda()
            y = torch.empty((Cs, 2), dtype=torch.float32, device="cuda:0")
            assert f.process_dev(xa[lead:], y) == (Cs, Cs)
            assert np.array_equal(yout, y.cpu().numpy()), "batch %d shard %d" % (k, g)
            if g > 0:
                ip, _, _, _ = cur.buffers(g - 1)
                tail = np.empty((K - 1, 2), np.float32)
                _lib.check(L.pcx_memcpy_d2h(tail.ctypes.data_as(C.c_void_p), C.c_void_p(ip + 8 * Cs), tail.nbytes, None))
                assert np.array_equal(xin[:K - 1], tail)
            else:
                assert nerr(yout[:4000], _oracle_fir(oracle, h, xin[:K - 1 + 4000], 4000)) <= TOL
                sums.add(float(np.abs(yout[:1000]).sum()))
    assert len(sums) == 5                                      # five different batches went through
    pair[1 if 5 & 1 else 0].compute()                          # the exchange still posted for batch 5: use it up
    for ns in pair:
        ns.sync()


@pytest.mark.parametrize("chain", [False, True], ids=["fir", "chain"])
@pytest.mark.parametrize("G,Cs", [(4, 9000), (3, 2080 * 3840), (8, 300000)])
def test_submit_threads_queue_the_same_pass(oracle, G, Cs, chain):
    """pcx_shard_set_submit_threads: a thread per device queues that device's share of every pass.  Several passes with the stream
    scattered again in between (the threads' waits and records against the caller's own transfers), halos poisoned: the outputs are
    bit for bit those of a handle driven from one thread, and shard 0's front matches the oracle."""
    from pothoscomms_amd import _lib, device, taps as tp
    h = tp.c4_taps() if chain else tp.c1_taps()
    K = len(h)
    outs = []
    for threads in (False, True):
        ns = device.NodeStream([0] * G, device.NodeStream.PEER_COPY)
        if chain:
            ns.set_chain(True, tp.C4_PHASE)
        ns.set_taps(h, complex_taps=not chain)
        ns.set_submit_threads(threads)
        ns.configure(Cs)
        got = []
        for rep in range(4):
            x = tp.fm_test_signal(K - 1 + G * Cs, seed=60 + rep) if chain else oracle.fill_uniform_f32(2 * (K - 1 + G * Cs), 70 + rep, 0).reshape(-1, 2)
            ns.scatter(x)
            if not chain:
                _poison_halos(ns)
            ns.step()
            got.append(ns.gather().copy())
        with pytest.raises(_lib.PcxError):                  # (a posted exchange: the switch is refused like the other setters)
            ns.post_exchange()
            ns.set_submit_threads(not threads)
        ns.compute()
        ns.sync()
        ns.set_submit_threads(False)                        # joins the threads; the handle goes on working from the caller's thread
        ns.step()
        ns.sync()
        ns.close()
        outs.append(got)
    for a, b in zip(*outs):
        assert np.isfinite(a).all() and np.array_equal(a, b)
    if not chain:
        x = oracle.fill_uniform_f32(2 * (K - 1 + G * Cs), 73, 0).reshape(-1, 2)
        n = min(G * Cs, 30000)
        assert nerr(outs[1][3][:n], _oracle_fir(oracle, h, x[:K - 1 + n], n)) <= TOL
