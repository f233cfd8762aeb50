import ctypes as C, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from pothoscomms_amd import _lib, device, taps as tp
L = _lib.load()
def pinned(shape):
    nb = int(np.prod(shape)) * 4
    p = C.c_void_p(); _lib.check(L.pcx_host_alloc(C.byref(p), nb))
    return np.ctypeslib.as_array((C.c_char * nb).from_address(p.value)).view(np.float32).reshape(shape)
n = 1 << 24
x, y = pinned((n + 2100, 2)), pinned((n, 2))
x[:] = np.random.default_rng(0).uniform(-1, 1, x.shape).astype(np.float32)
for K in (17, 255, 1025, 2049):
    f = device.FirFilter("complex_float32", "COMPLEX"); f.set_taps(tp.complex_bandpass(K, 0.05, 0.05))
    c, p = C.c_size_t(), C.c_size_t()
    run = lambda: _lib.check(L.pcx_fir_process(f._h, x.ctypes.data, n + K - 1, y.ctypes.data, n, C.byref(c), C.byref(p)))
    for _ in range(3): run()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(8): run()
        best = min(best, (time.perf_counter() - t0) / 8)
    Kov = (K - 1 + 15) // 16 * 16
    print("K=%5d  re-read %.3f  %.3f ms  %.2f Gs/s  (PCIe read %.1f GB/s, write %.1f)" % (K, 4096 / (4096 - Kov), best * 1e3, n / best / 1e9, 8 * n * 4096 / (4096 - Kov) / best / 1e9, 8 * n / best / 1e9))
