"""Does a host-pointer (in-place, PCIe) FIR call run faster on FEWER resident workgroups?  With 1024 slots a call of 1 Mi samples is
273 blocks = 273 workgroups of ONE block each: all load, then all compute, then all store -- reads and writes of the link never overlap.
With fewer workgroups each walks several blocks and fetches block k+1 at the foot of block k (fir_ols.hip), i.e. beside block k's stores.
pcx_fir_set_slots (a public call) sets the number; page-locked buffers, 255-tap complex_float32 FIR."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from pothoscomms_amd import _lib, device, taps as tp

L = _lib.load()
K = 255


def pinned(shape):
    nb = int(np.prod(shape)) * 4
    p = C.c_void_p()
    _lib.check(L.pcx_host_alloc(C.byref(p), nb))
    return np.ctypeslib.as_array((C.c_char * nb).from_address(p.value)).view(np.float32).reshape(shape)


for n in (1 << 16, 1 << 18, 1 << 20, 1 << 22, 1 << 24):
    x, y = pinned((n + K - 1, 2)), pinned((n, 2))
    x[:] = np.random.default_rng(0).uniform(-1, 1, x.shape).astype(np.float32)
    row = []
    for slots in ((1024,) if len(sys.argv) > 1 else (1024, 512, 256, 128)):     # (any argument: one column -- the diag library's PCX_OLS_SLOTS / PCX_SCHED_STATIC decide)
        f = device.FirFilter("complex_float32", "COMPLEX")
        f.set_taps(tp.c1_taps())
        f.set_slots(slots)
        c, p = C.c_size_t(), C.c_size_t()
        run = lambda: _lib.check(L.pcx_fir_process(f._h, x.ctypes.data, n + K - 1, y.ctypes.data, n, C.byref(c), C.byref(p)))
        for _ in range(5):
            run()
        reps = max(5, min(300, (1 << 26) // n))
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(reps):
                run()
            dt = (time.perf_counter() - t0) / reps
            best = dt if best is None else min(best, dt)
        row.append("%4d slots %7.3f ms %5.2f Gs/s" % (slots, best * 1e3, n / best / 1e9))
    print("n=%9d  " % n + " | ".join(row), flush=True)
