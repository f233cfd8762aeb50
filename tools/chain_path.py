"""Rotate -> FIRFilter(127 real taps) -> FreqDemod as THREE separate blocks: PCIe-inclusive rate with the inner edges in
page-locked host memory (what round 1 had: every block a round trip) against device-resident edges (the module's port domain),
against the one fused block (/comms/fm_demod_chain)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from pothoscomms_amd import _lib, blocks as B, taps as tp

L = _lib.load()
h, phase = tp.c4_taps(), tp.C4_PHASE
K = len(h)
for n in (1 << 18, 1 << 20, 1 << 22):
    rot = B.make("/comms/rotate", "complex_float32"); rot.call("setPhase", phase)
    fir = B.make("/comms/fir_filter", "complex_float32", "REAL"); fir.call("setTaps", h)
    dem = B.make("/comms/freq_demod", "complex_float32")
    fused = B.make("/comms/fm_demod_chain", "complex_float32", "REAL"); fused.call("setTaps", h); fused.call("setPhase", phase)
    for b in (rot, fir, dem, fused): b.activate()
    xin, _ = rot.port_buffer(0, (n + K - 1, 2), np.float32)
    yout, _ = dem.port_buffer(1, (n,), np.float32)
    xin[:] = np.random.default_rng(0).uniform(-1, 1, xin.shape).astype(np.float32)
    # (a) host edges: each block's own pinned output slab feeds the next block
    h1, _ = rot.port_buffer(1, (n + K - 1, 2), np.float32)
    h2, _ = fir.port_buffer(1, (n, 2), np.float32)
    def host_edges():
        rot.work_raw(xin.ctypes.data, n + K - 1, h1.ctypes.data, n + K - 1)
        fir.work_raw(h1.ctypes.data, n + K - 1, h2.ctypes.data, n)
        dem.work_raw(h2.ctypes.data, n, yout.ctypes.data, n)
    # (b) device edges
    e1, k1 = rot.link_buffer(fir, (n + K - 1) * 8)
    e2, k2 = fir.link_buffer(dem, n * 8)
    assert (k1, k2) == (2, 2)
    def dev_edges():
        rot.work_raw(xin.ctypes.data, n + K - 1, e1, n + K - 1)
        fir.work_raw(e1, n + K - 1, e2, n)
        dem.work_raw(e2, n, yout.ctypes.data, n)
    # (c) fused block
    xf, _ = fused.port_buffer(0, (n + K - 1, 2), np.float32); xf[:] = xin
    yf, _ = fused.port_buffer(1, (n,), np.float32)
    def one_block():
        fused.work_raw(xf.ctypes.data, n + K - 1, yf.ctypes.data, n)
    for name, fn in (("three blocks, host (pinned) edges", host_edges), ("three blocks, device-resident edges", dev_edges), ("one fused block", one_block)):
        fn()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        dt = (time.perf_counter() - t0) / reps
        print("n=%8d  %-38s %7.3f ms  %.2f Gsamples/s" % (n, name, dt * 1e3, n / dt / 1e9))
