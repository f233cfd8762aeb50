"""FIR throughput vs tap count for the complex_float32 kernels (device-resident, 64 Mi samples, shard on a 128-byte line)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pothoscomms_amd import _lib, device, taps as tp
d = torch.device("cuda", 0)
n = 64 * 1024 * 1024
for K in (1, 2, 4, 8, 16, 32, 63, 127, 255, 511, 1023, 1537, 2049, 2050, 3073, 4097, 4098, 6145, 6146, 8193):
    h = tp.complex_bandpass(K, 0.05, 0.05) if K > 1 else np.array([1.0 + 0j])
    lead = (-(K - 1)) % 16
    xa = torch.empty((lead + n + K - 1, 2), dtype=torch.float32, device=d); device.fill_uniform_f32_dev(xa, seed=1)
    x = xa[lead:]
    y = torch.empty((n, 2), dtype=torch.float32, device=d)
    row = []
    for algo in (_lib.FIR_DIRECT, _lib.FIR_OLS_FFT):
        if algo == _lib.FIR_DIRECT and K > 511:
            row.append("-"); continue
        f = device.FirFilter("complex_float32", "COMPLEX"); f.set_taps(h); f.set_algo(algo)
        try:
            for _ in range(60 if algo == _lib.FIR_OLS_FFT else 5): f.process_dev(x, y)
        except Exception as e:
            row.append("n/a"); continue
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 40 if algo == _lib.FIR_OLS_FFT else 5
        e0.record()
        for _ in range(reps): f.process_dev(x, y)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        row.append("%.1f Gs/s" % (n / ms / 1e6))
    print("K=%5d  time-domain tile %-12s overlap-save %-12s" % (K, row[0], row[1]), flush=True)
    del xa, x, y
