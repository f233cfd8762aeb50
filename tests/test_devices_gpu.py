"""Placement and port-slab settings of the blocks (VERDICT r05 tasks 3 / 4), and the lifetime of what the FIR block page-locks
(ADVICE r05): setDevice / getDevice / setPortSlabBytes on every block of the module, FIRFilter::setDevices({d}) == setDevice(d),
registrations counted across blocks and tied to the mapping they were made for."""
import ctypes as C

import numpy as np
import pytest

from pothoscomms_amd import _lib, blocks as B
from tests.util import TOL, nerr, rand_stream

pytestmark = pytest.mark.gpu

DEVICE_BLOCKS = [("/comms/fir_filter", ("complex_float32", "COMPLEX")), ("/comms/fft", ("complex_float32", 256, False)),
                 ("/comms/freq_demod", ("complex_float32",)), ("/comms/fm_demod_chain", ("complex_float32", "COMPLEX")),
                 ("/comms/rotate", ("complex_float32",)), ("/comms/scale", ("complex_float32",)), ("/comms/abs", ("complex_float32",)),
                 ("/comms/angle", ("complex_float32",)), ("/comms/conjugate", ("complex_float32",)),
                 ("/comms/arithmetic", ("complex_float32", "ADD")), ("/comms/split_complex", ("float32",)),
                 ("/comms/combine_complex", ("float32",))]


def _ngpus():
    import torch
    return torch.cuda.device_count()


def _kind(ptr):
    k = C.c_int()
    _lib.check(_lib.load().pcx_pointer_kind(C.c_void_p(ptr), C.byref(k)))
    return k.value


def test_every_device_block_has_the_placement_calls():
    n = _ngpus()
    for path, args in DEVICE_BLOCKS:
        blk = B.make(path, *args)
        calls = blk.calls()
        assert calls.get("setDevice") == 1 and calls.get("getDevice") == 0, path
        assert calls.get("setPortSlabBytes") == 1 and calls.get("getPortSlabBytes") == 0, path
        assert blk.call("getDevice") == 0
        blk.call("setDevice", 0)
        assert blk.call("getDevice") == 0
        with pytest.raises(_lib.InvalidArgument):
            blk.call("setDevice", n)                       # the process has devices 0 .. n-1
        assert blk.call("getDevice") == 0                  # ... and the block stays where it was
        blk.close()
    # the designer is host-side: no device, no ports
    assert "setDevice" not in B.make("/comms/fir_designer").calls()


def _run_stateful(path, args, x, place):
    blk = B.make(path, *args)
    K = 63
    rng = np.random.default_rng(5)
    taps = (rng.normal(size=K) + 1j * rng.normal(size=K)) / 8
    if "setTaps" in blk.calls():
        blk.call("setTaps", taps)
    if "setPhase" in blk.calls():
        blk.call("setPhase", 0.37)
    if path == "/comms/fir_filter":
        blk.call("setDecimation", 2); blk.call("setInterpolation", 3); blk.call("setKernel", "OLS_FFT")
    place(blk)
    blk.activate()
    n = x.shape[0] // 256 * 256 if path == "/comms/fft" else x.shape[0]
    y, c, p, _, _ = blk.work(x[:n], 4 * n)
    out = (y.copy(), c, p, blk.call("getDevice"))
    if path == "/comms/fir_filter":
        assert (blk.call("getDecimation"), blk.call("getInterpolation"), blk.call("getKernel")) == (2, 3, "OLS_FFT")
        assert len(blk.call("getTaps")) == K
    blk.close()
    return out


@pytest.mark.parametrize("path,args", DEVICE_BLOCKS[:4])
def test_a_block_placed_on_device_0_explicitly_is_bit_identical(path, args):
    """setDevice re-creates the handle(s) and pushes taps / phase / decimation / kernel choice again: the stream that comes out is
    the one of a block that was never moved, bit for bit"""
    rng = np.random.default_rng(3)
    x = rand_stream(rng, 1, 50000, True)
    a = _run_stateful(path, args, x, lambda b: None)
    b = _run_stateful(path, args, x, lambda b: b.call("setDevice", 0))
    assert a[1:] == b[1:] and a[1] > 0
    assert np.array_equal(a[0], b[0])
    if path == "/comms/fir_filter":
        c = _run_stateful(path, args, x, lambda b: b.call("setDevices", [0]))      # one ordinal = setDevice(that ordinal)
        assert np.array_equal(a[0], c[0])


def test_set_devices_with_one_ordinal_means_that_device():
    blk = B.make("/comms/fir_filter", "complex_float32", "COMPLEX")
    assert blk.call("getDevices") == []
    blk.call("setDevices", [0])
    assert blk.call("getDevices") == [0] and blk.call("getDevice") == 0
    with pytest.raises(_lib.InvalidArgument):
        blk.call("setDevices", [_ngpus()])
    assert blk.call("getDevices") == [0]
    blk.call("setDevices", [])
    assert blk.call("getDevices") == [] and blk.call("getDevice") == 0
    blk.close()


def test_the_maps_leave_the_callers_current_device_alone():
    L = _lib.load()
    cur = C.c_int(-1)
    _lib.check(L.pcx_get_device(C.byref(cur)))
    assert cur.value == 0
    rng = np.random.default_rng(8)
    x = rand_stream(rng, 1, 4096, True)
    blk = B.make("/comms/conjugate", "complex_float32")
    blk.call("setDevice", 0)
    y, c, p, _, _ = blk.work(x, 4096)
    assert (c, p) == (4096, 4096) and np.array_equal(y[:, 0], x[:, 0]) and np.array_equal(y[:, 1], -x[:, 1])
    _lib.check(L.pcx_get_device(C.byref(cur)))
    assert cur.value == 0
    blk.close()


@pytest.mark.skipif("_ngpus() < 2", reason="needs two GPUs")      # (the string form: evaluated when the test is set up, not when the file is imported)
def test_two_chains_on_two_devices(oracle):
    """eight independent chains on eight GPUs from one process is this, four times over: every block of a chain is placed with
    setDevice, streams match the oracle, and the calling thread's current device is what it was"""
    L = _lib.load()
    rng = np.random.default_rng(21)
    K = 127
    taps = (rng.normal(size=K) + 1j * rng.normal(size=K)) / 8
    x = [rand_stream(rng, 1, 200000, True) for _ in range(2)]
    chains = []
    for d in range(2):
        rot = B.make("/comms/rotate", "complex_float32"); rot.call("setDevice", d); rot.call("setPhase", 0.2 + d)
        fir = B.make("/comms/fir_filter", "complex_float32", "COMPLEX"); fir.call("setDevice", d); fir.call("setTaps", taps)
        dem = B.make("/comms/freq_demod", "complex_float32"); dem.call("setDevice", d)
        for b in (rot, fir, dem):
            assert b.call("getDevice") == d
            b.activate()
        chains.append((rot, fir, dem))
    for d, (rot, fir, dem) in enumerate(chains):
        y1, c, p, _, _ = rot.work(x[d], 200000)
        assert np.array_equal(y1, oracle.rotate(x[d], 0.2 + d))
        y2, c, p, _, _ = fir.work(y1, 200000)
        ref = oracle.Fir(1, True, True); ref.set_taps(taps); ref.activate()
        r2, rc, rp, _ = ref.work(y1, 200000)
        assert (c, p) == (rc, rp) and nerr(y2, r2) <= TOL
        y3, c, p, _, _ = dem.work(y2, p)
        r3 = oracle.FreqDemod(1).work(y2)
        dd = (y3.astype(np.float64) - r3 + np.pi) % (2 * np.pi) - np.pi
        assert float(np.max(np.abs(dd))) / np.pi <= 1e-5
    cur = C.c_int(-1)
    _lib.check(L.pcx_get_device(C.byref(cur)))
    assert cur.value == 0
    for ch in chains:
        for b in ch:
            b.close()


def test_port_slab_bytes_is_a_setting():
    fir = B.make("/comms/fir_filter", "complex_float32", "COMPLEX")
    assert fir.call("getPortSlabBytes") == 64 << 20
    assert fir.buffer_manager(False) == ("circular", 64 << 20) and fir.buffer_manager(True) == ("generic", 64 << 20)
    fir.call("setPortSlabBytes", 8 << 20)
    assert fir.buffer_manager(False) == ("circular", 8 << 20) and fir.buffer_manager(True) == ("generic", 8 << 20)
    for bad in (0, 1000, (1 << 30) + 1):
        with pytest.raises(_lib.InvalidArgument):
            fir.call("setPortSlabBytes", bad)
    assert fir.call("getPortSlabBytes") == 8 << 20
    fir.close()
    # FFT.cpp:54-59: the output slabs hold WHOLE frames, never less than one, whatever the setting
    for bins, slab in ((4096, 32 << 20), (4096, 100000), (3000, 1 << 20), (1 << 20, 1 << 20)):
        fft = B.make("/comms/fft", "complex_float32", bins, False)
        fft.call("setPortSlabBytes", slab)
        name, size = fft.buffer_manager(True)
        frame = bins * 8
        assert name == "generic" and size % frame == 0 and size >= frame and (size <= slab or size == frame), (bins, slab, size)
        fft.close()


def test_two_blocks_on_one_buffer_share_one_lock(oracle):
    """registrations were not counted across blocks -- the first block's destructor unlocked the buffer under the second.
    Now the second block is a second holder: the buffer stays page-locked until both have let go."""
    rng = np.random.default_rng(31)
    K = 63
    taps = (rng.normal(size=K) + 1j * rng.normal(size=K)) / 8
    circ = B.CircularBuffer(1 << 20)
    n = 40000
    x = rand_stream(rng, 1, n, True)
    win = circ.view(4096, n * 8, np.float32).reshape(-1, 2)
    win[:] = x
    ref = oracle.Fir(1, True, True); ref.set_taps(taps); ref.activate()
    ry, rc, rp, _ = ref.work(x, n)
    a = B.make("/comms/fir_filter", "complex_float32", "COMPLEX"); a.call("setTaps", taps); a.activate()
    b = B.make("/comms/fir_filter", "complex_float32", "COMPLEX"); b.call("setTaps", taps); b.activate()
    for blk in (a, b):
        y, c, p, _, _ = blk.work(win, n)
        assert (c, p) == (rc, rp) and nerr(y, ry) <= TOL
    assert _kind(circ.base + 64) == 1
    a.close()
    assert _kind(circ.base + 64) == 1                      # b still holds it
    y, c, p, _, _ = b.work(win, n)
    assert (c, p) == (rc, rp) and nerr(y, ry) <= TOL
    b.deactivate()                                         # lets go without being destroyed (a topology being re-committed)
    assert _kind(circ.base + 64) == 0
    b.activate()
    y, c, p, _, _ = b.work(win, n)                         # ... and locks again on first sight
    assert (c, p) == (rc, rp) and nerr(y, ry) <= TOL and _kind(circ.base + 64) == 1
    b.close()
    assert _kind(circ.base + 64) == 0
    circ.close()


def test_a_lock_does_not_outlive_its_mapping(oracle):
    """a buffer that is unmapped while a block still holds it, and a new one mapped at the SAME address: the old entry
    must not pass for a lock on the new memory.  The owner's munmap path lets go of everything in the range (pcxb_circular_destroy ->
    pcx_host_release_range), activate() drops what is no longer alive, and the new buffer is locked on first sight."""
    L = _lib.load()
    rng = np.random.default_rng(32)
    K = 63
    taps = (rng.normal(size=K) + 1j * rng.normal(size=K)) / 8
    n = 40000
    blk = B.make("/comms/fir_filter", "complex_float32", "COMPLEX"); blk.call("setTaps", taps); blk.activate()
    ref = oracle.Fir(1, True, True); ref.set_taps(taps)
    seen = set()
    for round_ in range(4):
        circ = B.CircularBuffer(1 << 20)                   # (the allocator tends to hand the same address out again)
        seen.add(circ.base)
        x = rand_stream(rng, 1, n, True)
        win = circ.view(8192, n * 8, np.float32).reshape(-1, 2)
        win[:] = x
        assert _kind(circ.base + 64) == 0                  # fresh pageable memory, whatever was there before
        ref.activate()
        ry, rc, rp, _ = ref.work(x, n)
        y, c, p, _, _ = blk.work(win, n)
        assert (c, p) == (rc, rp) and nerr(y, ry) <= TOL, round_
        assert _kind(circ.base + 64) == 1
        if round_ % 2 == 0:
            blk.deactivate()                               # the orderly way: the block lets go first ...
            assert _kind(circ.base + 64) == 0
            circ.close()
            blk.activate()
        else:
            base = circ.base
            alive = C.c_int(-1)
            _lib.check(L.pcx_host_mapping_alive(C.c_void_p(base), C.byref(alive)))
            assert alive.value == 1
            circ.close()                                   # ... and the other way: the owner unmaps under a holder
            with pytest.raises(_lib.PcxError):
                _lib.check(L.pcx_host_mapping_alive(C.c_void_p(base), C.byref(alive)))    # the registration went with the mapping
            blk.deactivate(); blk.activate()               # (the block's stale entry is dropped, quietly)
    blk.close()
    assert len(seen) >= 1
