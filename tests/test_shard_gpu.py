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
