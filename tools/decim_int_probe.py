"""complex_int16 decimating FIR input rate on the time-domain resampling kernel (one output per lane)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from pothoscomms_amd import device, taps as tp
d = torch.device("cuda", 0)
n = 16 * 1024 * 1024
for dtype, tdt in (("complex_int16", torch.int16), ("complex_float64", torch.float64)):
    for M in (2, 4, 8, 16):
        for K in (63, 255):
            h = tp.complex_bandpass(K, 0.05, 0.05)
            x = torch.randint(-100, 100, (n + K - 1, 2), device=d).to(tdt)
            y = torch.empty((n // M + 8, 2), dtype=tdt, device=d)
            f = device.FirFilter(dtype, "COMPLEX"); f.set_taps(h); f.set_decimation(M)
            for _ in range(2): f.process_dev(x, y)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): f.process_dev(x, y)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 3
            print("%s M=%2d K=%3d: %.3f ms  %.1f Gsamples/s in" % (dtype, M, K, ms, n / ms / 1e6), flush=True)
