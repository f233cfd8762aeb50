"""FIR throughput vs tap count for the two complex_float32 kernels (device-resident, 16 Mi samples)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pothoscomms_amd import _lib, device, taps as tp
d = torch.device("cuda", 0)
n = 16 * 1024 * 1024
for K in (1, 4, 8, 16, 24, 32, 48, 64, 127, 255, 511, 1023, 2049):
    h = tp.complex_bandpass(K, 0.05, 0.05) if K > 1 else np.array([1.0 + 0j])
    x = torch.empty((n + K - 1, 2), dtype=torch.float32, device=d); device.fill_uniform_f32_dev(x, seed=1)
    y = torch.empty((n, 2), dtype=torch.float32, device=d)
    row = []
    for algo in (_lib.FIR_DIRECT, _lib.FIR_OLS_FFT):
        f = device.FirFilter("complex_float32", "COMPLEX"); f.set_taps(h); f.set_algo(algo)
        try:
            for _ in range(5): f.process_dev(x, y)
        except Exception as e:
            row.append("n/a"); continue
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f.process_dev(x, y)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        row.append("%.1f Gs/s" % (n / ms / 1e6))
    print("K=%5d  direct %-12s ols %-12s" % (K, row[0], row[1]))
