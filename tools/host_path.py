"""PCIe-inclusive rate of the host-buffer entry points (what a Pothos work() call pays)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pothoscomms_amd import device, taps as tp
for n in (1 << 16, 1 << 20, 1 << 24):
    K = 255
    x = np.random.default_rng(0).uniform(-1, 1, (n + K - 1, 2)).astype(np.float32)
    f = device.FirFilter("complex_float32", "COMPLEX"); f.set_taps(tp.c1_taps())
    f.process(x, n)
    t0 = time.perf_counter(); reps = 5
    for _ in range(reps): y, c, p = f.process(x, n)
    dt = (time.perf_counter() - t0) / reps
    print("fir host path  n=%9d  %.3f ms  %.2f Gsamples/s  (%.1f GB/s each way)" % (n, dt * 1e3, n / dt / 1e9, 8 * n / dt / 1e9))
    device.conj(x)                      # (warm-up: the thread's map workspace and its buffers)
    t0 = time.perf_counter()
    for _ in range(reps): device.conj(x)
    dt = (time.perf_counter() - t0) / reps
    print("conj host path n=%9d  %.3f ms  %.2f Gsamples/s" % (n, dt * 1e3, n / dt / 1e9))

# the same call on page-locked buffers (pcx_host_alloc): what a pinned BufferManager would give
import ctypes as C
from pothoscomms_amd import _lib
L = _lib.load()
n, K = 1 << 24, 255
def pinned(shape, dtype):
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = C.c_void_p(); _lib.check(L.pcx_host_alloc(C.byref(p), nbytes))
    return np.ctypeslib.as_array((C.c_char * nbytes).from_address(p.value)).view(dtype).reshape(shape), p
x, px = pinned((n + K - 1, 2), np.float32); y, py = pinned((n, 2), np.float32)
x[:] = np.random.default_rng(0).uniform(-1, 1, x.shape).astype(np.float32)
f = device.FirFilter("complex_float32", "COMPLEX"); f.set_taps(tp.c1_taps())
c, p = C.c_size_t(), C.c_size_t()
def run(): _lib.check(L.pcx_fir_process(f._h, x.ctypes.data, n + K - 1, y.ctypes.data, n, C.byref(c), C.byref(p)))
run()
t0 = time.perf_counter()
for _ in range(5): run()
dt = (time.perf_counter() - t0) / 5
print("fir host path, pinned buffers  n=%9d  %.3f ms  %.2f Gsamples/s (%.1f GB/s each way)" % (n, dt * 1e3, n / dt / 1e9, 8 * n / dt / 1e9))


# ---- a plain block work() loop on the buffers the BLOCK hands the scheduler (pinned slabs): nothing staged ----
from pothoscomms_amd import blocks
for n in (1 << 18, 1 << 20, 1 << 22):
    blk = blocks.make("/comms/fir_filter", "complex_float32", "COMPLEX")
    blk.call("setTaps", tp.c1_taps())
    blk.activate()
    K = 255
    xin, pin_in = blk.port_buffer(0, (n + K - 1, 2), np.float32)
    yout, pin_out = blk.port_buffer(1, (n, 2), np.float32)
    xin[:] = np.random.default_rng(0).uniform(-1, 1, xin.shape).astype(np.float32)
    blk.work(xin, n, outbuf=yout)
    t0 = time.perf_counter(); reps = 20
    for _ in range(reps):
        _, c, p, _, _ = blk.work(xin, n, outbuf=yout)
    dt = (time.perf_counter() - t0) / reps
    assert p == n
    print("/comms/fir_filter work() on its own port buffers (pinned in=%s out=%s)  n=%9d  %.3f ms  %.2f Gsamples/s" % (pin_in, pin_out, n, dt * 1e3, n / dt / 1e9))
