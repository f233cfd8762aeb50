"""complex_int16 / complex_float64 interpolating FIR output rate (255 taps per polyphase row), 8 Mi input samples."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from pothoscomms_amd import _lib, device, taps as tp
d = torch.device("cuda", 0)
n = 8 * 1024 * 1024
for dtype, tdt in (("complex_int16", torch.int16), ("complex_float64", torch.float64)):
    for L in (2, 4, 8):
        K = 255
        h = tp.complex_bandpass(K * L, 0.05 / L, 0.05 / L) * L * 0.4
        x = torch.randint(-100, 100, (n + K - 1, 2), device=d).to(tdt)
        y = torch.empty((n * L + 8, 2), dtype=tdt, device=d)
        for algo, name in ((_lib.FIR_EXACT, "time-domain"), (_lib.FIR_AUTO, "auto")):
            f = device.FirFilter(dtype, "COMPLEX"); f.set_taps(h); f.set_interpolation(L); f.set_algo(algo)
            for _ in range(2): f.process_dev(x, y)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): f.process_dev(x, y)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 3
            print("%s L=%d %-12s %.3f ms  %.1f Gsamples/s out" % (dtype, L, name, ms, n * L / ms / 1e6), flush=True)
