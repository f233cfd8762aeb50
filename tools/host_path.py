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


# ---- the FRAMEWORK's circular input (pageable, double-mapped: what the FIR gets inside Pothos, FIRFilter.cpp:196-199) ----
# work() after work() on a ring of 4x the call size, the window sliding through it and across the wrap; the block page-locks the
# mapping on first sight (pcx_host_register_mapping), the output goes to the block's own pinned slabs.  `lock=False` shows what the
# same loop costs when the ring is NOT locked (kLockFrom bytes per call are not reached: K-1 history + a call below 64 KiB never is,
# so the comparison run keeps the window just under that... no: it uses a private copy of the window, which cannot be locked).
def circular_loop(n, lock=True, reps=20):
    K = 255
    blk = blocks.make("/comms/fir_filter", "complex_float32", "COMPLEX")
    blk.call("setTaps", tp.c1_taps())
    blk.activate()
    circ = blocks.CircularBuffer(4 * (n + K) * 8)
    cap = circ.size // 8
    yout, pin_out = blk.port_buffer(1, (n, 2), np.float32)
    src = np.random.default_rng(0).uniform(-1, 1, (n, 2)).astype(np.float32)
    rd = 0
    circ.view(0, (K - 1) * 8, np.float32)[:] = 0
    wr = K - 1
    dt = 0.0
    for it in range(reps + 2):
        circ.view((wr % cap) * 8, n * 8, np.float32).reshape(-1, 2)[:] = src        # the producer (not timed)
        wr += n
        win = circ.view((rd % cap) * 8, (wr - rd) * 8, np.float32).reshape(-1, 2)
        if not lock:
            win = np.array(win)                            # a private (heap) copy: pageable and not lockable -> staged
        t0 = time.perf_counter()
        _, c, p, _, _ = blk.work(win, n, outbuf=yout)
        if it >= 2:
            dt += time.perf_counter() - t0
        assert c == n and p == n
        rd += c
    blk.close()
    circ.close()
    return dt / reps


for n in (1 << 18, 1 << 20, 1 << 22):
    a = circular_loop(n, lock=False)
    b = circular_loop(n, lock=True)
    print("/comms/fir_filter work() on the framework's CIRCULAR input, n=%9d: pageable (staged) %.3f ms %.2f Gsamples/s | page-locked where it lies %.3f ms %.2f Gsamples/s"
          % (n, a * 1e3, n / a / 1e9, b * 1e3, n / b / 1e9))
