"""Host-pointer (page-locked, in place over PCIe) rates of the other entry points: FFT 4096, conj / rotate (maps), freq_demod, the fused chain,
a decimating FIR -- GB/s each way against the 43 a plain copy kernel reaches on the same buffers (tools/pcie_lab.hip)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from pothoscomms_amd import _lib, device, taps as tp

L = _lib.load()


def pinned(shape, dtype=np.float32):
    nb = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = C.c_void_p()
    _lib.check(L.pcx_host_alloc(C.byref(p), nb))
    return np.ctypeslib.as_array((C.c_char * nb).from_address(p.value)).view(dtype).reshape(shape)


def best(fn, n):
    for _ in range(4):
        fn()
    reps = max(5, min(200, (1 << 26) // n))
    b = None
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        dt = (time.perf_counter() - t0) / reps
        b = dt if b is None else min(b, dt)
    return b


rng = np.random.default_rng(0)
for n in (1 << 18, 1 << 20, 1 << 22, 1 << 24):
    x = pinned((n + 4096, 2)); x[:] = rng.uniform(-1, 1, x.shape).astype(np.float32)
    y = pinned((n + 4096, 2))
    yr = pinned((n + 4096,))
    row = []
    fft = device.Fft("complex_float32", 4096, False)
    dt = best(lambda: _lib.check(L.pcx_fft_transform(fft._h, x.ctypes.data, y.ctypes.data, n // 4096)), n)
    row.append("fft4096 %.3f ms %5.2f Gs/s (%4.1f GB/s in, %4.1f out)" % (dt * 1e3, n / dt / 1e9, 8 * n / dt / 1e9, 8 * n / dt / 1e9))
    dt = best(lambda: _lib.check(L.pcx_conj(_lib.F32, x.ctypes.data, y.ctypes.data, n)), n)
    row.append("conj %.3f ms %5.2f Gs/s (%4.1f / %4.1f)" % (dt * 1e3, n / dt / 1e9, 8 * n / dt / 1e9, 8 * n / dt / 1e9))
    fd = device.FreqDemod("complex_float32")
    dt = best(lambda: _lib.check(L.pcx_freqdemod_process(fd._h, x.ctypes.data, yr.ctypes.data, n)), n)
    row.append("freq_demod %.3f ms %5.2f Gs/s (%4.1f / %4.1f)" % (dt * 1e3, n / dt / 1e9, 8 * n / dt / 1e9, 4 * n / dt / 1e9))
    ch = device.FmChain(); ch.set_phase(0.7); ch.set_taps(tp.c4_taps(), False)
    c, p = C.c_size_t(), C.c_size_t()
    dt = best(lambda: _lib.check(L.pcx_fmchain_process(ch._h, x.ctypes.data, n + 126, yr.ctypes.data, n, C.byref(c), C.byref(p))), n)
    row.append("fm chain %.3f ms %5.2f Gs/s (%4.1f / %4.1f)" % (dt * 1e3, n / dt / 1e9, 8 * n / dt / 1e9, 4 * n / dt / 1e9))
    f = device.FirFilter("complex_float32", "COMPLEX"); f.set_taps(tp.complex_bandpass(255, 0.05 / 8, 0.05 / 8)); f.set_decimation(8)
    dt = best(lambda: _lib.check(L.pcx_fir_process(f._h, x.ctypes.data, n + 254, y.ctypes.data, n // 8, C.byref(c), C.byref(p))), n)
    row.append("fir decim 8 %.3f ms %5.2f Gs/s in (%4.1f / %4.1f)" % (dt * 1e3, n / dt / 1e9, 8 * n / dt / 1e9, n / dt / 1e9))
    f = device.FirFilter("complex_float32", "COMPLEX"); f.set_taps(tp.complex_bandpass(255 * 4, 0.05 / 4, 0.05 / 4) * 4); f.set_interpolation(4)
    dt = best(lambda: _lib.check(L.pcx_fir_process(f._h, x.ctypes.data, n // 4 + 254, y.ctypes.data, n, C.byref(c), C.byref(p))), n)
    row.append("fir interp 4 %.3f ms %5.2f Gs/s out" % (dt * 1e3, n / dt / 1e9))
    xi = x.view(np.int16); yi = y.view(np.int16)         # complex_int16: 4 bytes per sample
    f = device.FirFilter("complex_int16", "COMPLEX"); f.set_taps(tp.c1_taps() * 0.9)
    dt = best(lambda: _lib.check(L.pcx_fir_process(f._h, xi.ctypes.data, n + 254, yi.ctypes.data, n, C.byref(c), C.byref(p))), n)
    row.append("fir i16 %.3f ms %5.2f Gs/s (%4.1f GB/s each way)" % (dt * 1e3, n / dt / 1e9, 4 * n / dt / 1e9))
    f = device.FirFilter("float32", "REAL"); f.set_taps(tp.lowpass(255, 0.1))
    dt = best(lambda: _lib.check(L.pcx_fir_process(f._h, x.ctypes.data, 2 * n + 254, y.ctypes.data, 2 * n, C.byref(c), C.byref(p))), n)
    row.append("fir real f32 %.3f ms %5.2f Gs/s (%4.1f GB/s each way)" % (dt * 1e3, 2 * n / dt / 1e9, 8 * n / dt / 1e9))
    print("n=%9d  " % n + " | ".join(row), flush=True)
