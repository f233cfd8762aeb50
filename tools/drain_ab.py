"""A/B of the drained output of the host-pointer FIR call (pcx_api.hip drain_*) on ONE box: the diagnostic library reads
PCX_DRAIN_FROM (bytes of output from which a page-locked call is drained; huge = never: in place both ways) and PCX_DRAIN_CHUNK.
Each setting runs in a fresh process; page-locked buffers, 255-tap complex_float32 FIR, n samples per call."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, %r)
from pothoscomms_amd import _lib, device, taps as tp
L = _lib.load()
K = 255
def pinned(shape):
    nb = int(np.prod(shape)) * 4
    p = C.c_void_p(); _lib.check(L.pcx_host_alloc(C.byref(p), nb))
    return np.ctypeslib.as_array((C.c_char * nb).from_address(p.value)).view(np.float32).reshape(shape)
for n in (1 << 18, 1 << 20, 1 << 22, 1 << 24):
    x, y = pinned((n + K - 1, 2)), pinned((n, 2))
    x[:] = np.random.default_rng(0).uniform(-1, 1, x.shape).astype(np.float32)
    f = device.FirFilter("complex_float32", "COMPLEX"); f.set_taps(tp.c1_taps())
    c, p = C.c_size_t(), C.c_size_t()
    run = lambda: _lib.check(L.pcx_fir_process(f._h, x.ctypes.data, n + K - 1, y.ctypes.data, n, C.byref(c), C.byref(p)))
    for _ in range(5): run()
    reps = max(5, min(200, (1 << 26) // n))
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps): run()
        ts.append((time.perf_counter() - t0) / reps)
    dt = min(ts)
    print("  n=%%9d  %%.3f ms  %%.2f Gsamples/s  (%%.1f GB/s each way)" %% (n, dt * 1e3, n / dt / 1e9, 8 * n / dt / 1e9), flush=True)
''' % ROOT

for name, env in (("in place both ways (drain off)", {"PCX_DRAIN_FROM": str(1 << 40)}),
                  ("drained, 1 MiB chunks", {"PCX_DRAIN_CHUNK": str(1 << 20), "PCX_DRAIN_FROM": str(1 << 20)}),
                  ("drained, 2 MiB chunks", {"PCX_DRAIN_CHUNK": str(2 << 20), "PCX_DRAIN_FROM": str(1 << 20)}),
                  ("drained, 4 MiB chunks", {"PCX_DRAIN_CHUNK": str(4 << 20), "PCX_DRAIN_FROM": str(1 << 20)}),
                  ("drained, 8 MiB chunks", {"PCX_DRAIN_CHUNK": str(8 << 20), "PCX_DRAIN_FROM": str(1 << 20)}),
                  ("drained, 32 MiB chunks", {"PCX_DRAIN_CHUNK": str(32 << 20), "PCX_DRAIN_FROM": str(1 << 20)})):
    print(name, flush=True)
    e = dict(os.environ, PCX_HIP_LIBRARY=os.path.join(ROOT, "pothoscomms_amd", "libpcx_hip_diag.so"), **env)
    subprocess.run([sys.executable, "-c", CHILD], env=e, check=False)
