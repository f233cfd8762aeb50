"""complex_int16 / complex_int8 FIR throughput vs tap count: time-domain kernels (packed dot product / sliding window)
against the exact integer overlap-save on the double transform (fir_ols_f64.hip); device-resident, 16 Mi samples."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pothoscomms_amd import _lib, device, taps as tp
d = torch.device("cuda", 0)
n = 16 * 1024 * 1024
REAL = "real" in sys.argv      # `real`: float64 / int16 / int8 REAL streams instead of the complex integer ones
if REAL:
    sys.argv.remove("real")
Ks = [int(a) for a in sys.argv[1:]] or [8, 16, 24, 32, 48, 63, 127, 255, 511, 1023, 2049, 4097]
TYPES = (("float64", torch.float64), ("int16", torch.int16), ("int8", torch.int8)) if REAL else (("complex_int16", torch.int16), ("complex_int8", torch.int8))
for dtype, tdt in TYPES:
    for K in Ks:
        h = tp.lowpass(K, 0.05) if REAL else tp.complex_bandpass(K, 0.05, 0.05)
        shape = (n + K - 1,) if REAL else (n + K - 1, 2)
        x = torch.randint(-100, 100, shape, device=d).to(tdt)
        y = torch.empty((n,) if REAL else (n, 2), dtype=tdt, device=d)
        row = []
        for algo in (_lib.FIR_EXACT, _lib.FIR_OLS_FFT):
            if algo == _lib.FIR_EXACT and K > 1023:
                row.append("-"); continue
            f = device.FirFilter(dtype, "REAL" if REAL else "COMPLEX"); f.set_taps(h); f.set_algo(algo)
            warm, reps = (10, 10) if algo == _lib.FIR_OLS_FFT else (2, 3)
            try:
                for _ in range(warm): f.process_dev(x, y)
            except Exception as e:
                row.append("n/a"); continue
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): f.process_dev(x, y)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            row.append("%.1f Gs/s" % (n / ms / 1e6))
        print("%s K=%5d  time-domain %-14s overlap-save (exact) %-14s" % (dtype, K, row[0], row[1]), flush=True)
        del x, y
